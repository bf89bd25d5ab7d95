// Bandwidth-bound glue layers on NHWC fp32 activations (all float4-vectorised over
// channels, grid-stride).  Reference counterparts:
//   max pool 2x2/2      caffe/src/caffe/layers/pooling_layer.cu:11-47 (+ceil sizing, pooling_layer.cpp:79-123)
//   depthwise deconv    caffe/src/caffe/layers/deconv_layer.cu:8-23 (256 serial GEMMs + col2im in the reference)
//   concat              caffe/src/caffe/layers/concat_layer.cu (zero-copy here: producers write channel slices)
//   NCHW<->NHWC         the host-visible Blob.data layout (caffe/python/caffe/_caffe.cpp:222-242)
#include <algorithm>
#include <type_traits>

#include "conv_common.h"
#include "shf_internal.h"

namespace shf {

static inline unsigned grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

__global__ void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                               int Ho, int Wo, int k, int stride, int pad, int in_stride, int out_stride) {
  const int C4 = C >> 2;
  const long long total = (long long)B * Ho * Wo * C4;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total;
       n += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(n % C4);
    long long P = n / C4;
    const int ox = (int)(P % Wo), oy = (int)((P / Wo) % Ho), b = (int)(P / ((long long)Wo * Ho));
    int hs = oy * stride - pad, ws = ox * stride - pad;
    const int he = min(hs + k, H), we = min(ws + k, W);
    hs = max(hs, 0);
    ws = max(ws, 0);
    float4 m = make_float4(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
    for (int y = hs; y < he; ++y)
      for (int x = ws; x < we; ++x) {
        const float4 v = *(const float4*)(in + ((size_t)(b * H + y) * W + x) * in_stride + c4 * 4);
        m.x = v.x > m.x ? v.x : m.x;  // first max wins ('>'), pooling_layer.cpp:160-163
        m.y = v.y > m.y ? v.y : m.y;
        m.z = v.z > m.z ? v.z : m.z;
        m.w = v.w > m.w ? v.w : m.w;
      }
    *(float4*)(out + (size_t)P * out_stride + c4 * 4) = m;
  }
}

int launch_maxpool(const View& in, const View& out, int k, int stride, int pad, hipStream_t s) {
  if (in.C % 4 || in.cstride % 4 || in.coff % 4 || out.cstride % 4 || out.coff % 4) {
    set_error("maxpool: channel count / views must be multiples of 4");
    return -1;
  }
  const long long total = (long long)out.B * out.H * out.W * (in.C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, s, in.p + in.coff, out.p + out.coff, in.B,
                     in.H, in.W, in.C, out.H, out.W, k, stride, pad, in.cstride, out.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// A pooled blob's activation-exponent slot: max |pooled| <= max |input|, so the input's published maximum serves as
// the bound.  RAISED, never copied: the slot may be shared (concat members use their owner's) and other producers raise it.
__global__ void amax_raise_kernel(unsigned* dst, const unsigned* src) { atomicMax(dst, *src); }

int launch_amax_raise(unsigned* dst, const unsigned* src, hipStream_t s) {
  hipLaunchKernelGGL(amax_raise_kernel, dim3(1), dim3(1), 0, s, dst, src);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// out[b,y,x,c] = sum_{a,bb} in[b,(y+pad-a)/s,(x+pad-bb)/s,c] * w[c,a,bb] over taps whose
// source index is integral and in range (col2im accumulation order a-major, like im2col.cpp:168-185).
// Grid: x over (ox, channel quad) of one output row, y = output row, z = batch -- 32-bit index math only (the
// flat 64-bit div/mod chain of a grid-stride loop cost more than the memory traffic).
__global__ void deconv_dw_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W, int C,
                                 int Ho, int Wo, int k, int stride, int pad, int in_stride, int out_stride,
                                 int* range_flag, unsigned* out_amax) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < (unsigned)Wo * C4) {   // (no early return: every lane takes part in the wave reduction below)
  const int c4 = (int)(n % C4), ox = (int)(n / C4), oy = blockIdx.y, b = blockIdx.z;
  // only the taps a = (oy+pad) mod stride, +stride, ... hit an integral source row (same for columns): walk
  // exactly those, in the same ascending (a, bb) order as the full k x k scan
  for (int a = (oy + pad) % stride; a < k; a += stride) {
    const int ty = oy + pad - a;
    if (ty < 0) break;
    const int iy = ty / stride;
    if (iy >= H) continue;
    for (int bb = (ox + pad) % stride; bb < k; bb += stride) {
      const int tx = ox + pad - bb;
      if (tx < 0) break;
      const int ix = tx / stride;
      if (ix >= W) continue;
      const float4 v = *(const float4*)(in + ((size_t)(b * H + iy) * W + ix) * in_stride + c4 * 4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(vv[j], w[((size_t)(c4 * 4 + j) * k + a) * k + bb], acc[j]);
    }
  }
  if (bias)
    for (int j = 0; j < 4; ++j) acc[j] += bias[c4 * 4 + j];
  *(float4*)(out + ((size_t)(b * Ho + oy) * Wo + ox) * out_stride + c4 * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  // split-fp16 mode: the consumer of this map splits it to fp16 hi/lo -- raise the range flag beyond 65504, and the
  // unit's max |value| for the consumer's activation exponent (conv_common.h)
  const float amax = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
  if (range_flag && !(amax <= 65504.0f)) atomicOr(range_flag, 1);
  conv_publish_amax(out_amax, nullptr, amax);
}

// the same for a GROUP of maps sharing the layer (the units of an image's pyramid): one launch, the members stacked
// along the output-row axis of the grid
struct DeconvGM {
  const float* in;
  float* out;
  int H, W, Ho, Wo, in_stride, out_stride, row_start;
  unsigned* out_amax;   // activation-exponent slot of the member's output blob (conv_common.h) or null
};
struct DeconvGK {
  int n, C, k, stride, pad;
  const float* w;
  const float* bias;
  int* range_flag;
  int rows_per_block;
  DeconvGM m[16];
};
// K4S2P1: the detector's only deconvolution (kernel 4, stride 2, pad 1: exact 2x upsampling) with the tap geometry
// resolved at compile time -- output row oy takes input rows (oy + 1) / 2 - 1 + {0, 1} with taps (oy + 1) % 2 + {2, 0}
template <bool K4S2P1>
__global__ void deconv_dw_group_kernel(DeconvGK g) {
  int mi = 0;
#pragma unroll
  for (int q = 1; q < 16; ++q) mi += (q < g.n && (int)blockIdx.y >= g.m[q].row_start) ? 1 : 0;
  const DeconvGM& p = g.m[mi];
  const unsigned C4 = (unsigned)g.C >> 2;
  const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
  float amax = 0.f;
  if (n < (unsigned)p.Wo * C4) {   // (no early return: every lane takes part in the wave reduction at the end)
  const int c4 = (int)(n % C4), ox = (int)(n / C4);
  // the thread's 4 channels x 16 taps, fetched ONCE for all its rows (re-fetching them per output pixel was 1.8 GB of
  // cache traffic per image: the whole cost of this kernel)
  float wv[4][16];
  if constexpr (K4S2P1) {
    const float4* wq = (const float4*)(g.w + (size_t)c4 * 4 * 16);   // 16-byte aligned (launcher)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = wq[j * 4 + q];
        wv[j][4 * q] = t.x; wv[j][4 * q + 1] = t.y; wv[j][4 * q + 2] = t.z; wv[j][4 * q + 3] = t.w;
      }
  }
  // a block does g.rows_per_block output rows (the launch is bound by the block dispatch rate otherwise: ~8 ns per
  // 256-thread block of one row)
  for (int oy = ((int)blockIdx.y - p.row_start) * g.rows_per_block, oy_end = min(oy + g.rows_per_block, p.Ho); oy < oy_end; ++oy) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (K4S2P1) {
    // the generic loop below visits the taps in rising a (falling input row), rising bb: the same order here
    const int a0 = (oy + 1) & 1, b0 = (ox + 1) & 1;
    // (a0, b0) are wave-uniform (a wave is one output pixel x 64 channel quads): four straight-line variants with
    // compile-time tap indices instead of a register-array lookup
    auto taps = [&](auto A0_, auto B0_) {
      constexpr int A0 = decltype(A0_)::value, B0 = decltype(B0_)::value;
#pragma unroll
      for (int da = 0; da < 2; ++da) {
        constexpr int dummy = 0; (void)dummy;
        const int a = A0 + 2 * da, iy = (oy + 1 - a) >> 1;
        if (oy + 1 - a < 0 || iy >= p.H) continue;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int bb = B0 + 2 * db, ix = (ox + 1 - bb) >> 1;
          if (ox + 1 - bb < 0 || ix >= p.W) continue;
          const float4 v = *(const float4*)(p.in + ((size_t)iy * p.W + ix) * p.in_stride + c4 * 4);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(vv[j], wv[j][(A0 + 2 * da) * 4 + B0 + 2 * db], acc[j]);
        }
      }
    };
    using std::integral_constant;
    if (a0 == 0 && b0 == 0) taps(integral_constant<int, 0>{}, integral_constant<int, 0>{});
    else if (a0 == 0) taps(integral_constant<int, 0>{}, integral_constant<int, 1>{});
    else if (b0 == 0) taps(integral_constant<int, 1>{}, integral_constant<int, 0>{});
    else taps(integral_constant<int, 1>{}, integral_constant<int, 1>{});
  } else {
    const int k = g.k, stride = g.stride, pad = g.pad;
    for (int a = (oy + pad) % stride; a < k; a += stride) {
      const int ty = oy + pad - a;
      if (ty < 0) break;
      const int iy = ty / stride;
      if (iy >= p.H) continue;
      for (int bb = (ox + pad) % stride; bb < k; bb += stride) {
        const int tx = ox + pad - bb;
        if (tx < 0) break;
        const int ix = tx / stride;
        if (ix >= p.W) continue;
        const float4 v = *(const float4*)(p.in + ((size_t)iy * p.W + ix) * p.in_stride + c4 * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(vv[j], g.w[((size_t)(c4 * 4 + j) * k + a) * k + bb], acc[j]);
      }
    }
  }
  if (g.bias)
    for (int j = 0; j < 4; ++j) acc[j] += g.bias[c4 * 4 + j];
  *(float4*)(p.out + ((size_t)oy * p.Wo + ox) * p.out_stride + c4 * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))));
  if (acc[0] != acc[0] || acc[1] != acc[1] || acc[2] != acc[2] || acc[3] != acc[3]) amax = __builtin_inff();  // (fmaxf drops NaNs)
  }
  }
  if (g.range_flag && !(amax <= 65504.0f)) atomicOr(g.range_flag, 1);
  conv_publish_amax(p.out_amax, nullptr, amax);
}

int launch_deconv_depthwise_group(const View* ins, const View* outs, int n, const float* w, const float* bias, int k,
                                  int stride, int pad, hipStream_t s, int* range_flag, unsigned* const* out_amax) {
  if (n < 1 || n > 16) { set_error("deconv group: 1..16 members"); return -1; }
  DeconvGK g;
  g.n = n; g.C = ins[0].C; g.k = k; g.stride = stride; g.pad = pad;
  g.w = w; g.bias = bias; g.range_flag = range_flag;
  g.rows_per_block = 8;
  int rows = 0;
  unsigned per_row_max = 1;
  for (int i = 0; i < n; ++i) {
    const View& in = ins[i];
    const View& out = outs[i];
    if (in.B != 1 || in.C != g.C || in.C % 4 || in.cstride % 4 || in.coff % 4 || out.cstride % 4 || out.coff % 4) {
      set_error("deconv group: batch 1, shared channel count, views multiples of 4");
      return -1;
    }
    DeconvGM& m = g.m[i];
    m.in = in.p + in.coff; m.out = out.p + out.coff;
    m.H = in.H; m.W = in.W; m.Ho = out.H; m.Wo = out.W;
    m.in_stride = in.cstride; m.out_stride = out.cstride;
    m.out_amax = out_amax ? out_amax[i] : nullptr;
    m.row_start = rows;
    rows += (out.H + g.rows_per_block - 1) / g.rows_per_block;
    per_row_max = std::max(per_row_max, ((unsigned)out.W * (unsigned)(in.C / 4) + 255) / 256);
  }
  if (k == 4 && stride == 2 && pad == 1 && (((uintptr_t)w) & 15) == 0)
    hipLaunchKernelGGL(deconv_dw_group_kernel<true>, dim3(per_row_max, rows, 1), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(deconv_dw_group_kernel<false>, dim3(per_row_max, rows, 1), dim3(256), 0, s, g);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_deconv_depthwise(const View& in, const View& out, const float* w, const float* bias, int k, int stride,
                            int pad, hipStream_t s, int* range_flag, unsigned* out_amax) {
  if (in.C % 4 || in.cstride % 4 || in.coff % 4 || out.cstride % 4 || out.coff % 4) {
    set_error("deconv: channel count / views must be multiples of 4");
    return -1;
  }
  const unsigned per_row = (unsigned)out.W * (unsigned)(in.C / 4);
  hipLaunchKernelGGL(deconv_dw_kernel, dim3((per_row + 255) / 256, out.H, out.B), dim3(256), 0, s, in.p + in.coff, w,
                     bias, out.p + out.coff, in.B, in.H, in.W, in.C, out.H, out.W, k, stride, pad, in.cstride,
                     out.cstride, range_flag, out_amax);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

__global__ void copy_view_kernel(const float* __restrict__ in, float* __restrict__ out, long long pixels, int C,
                                 int in_stride, int out_stride) {
  const long long total = pixels * C;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total;
       n += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(n % C);
    const long long P = n / C;
    out[(size_t)P * out_stride + c] = in[(size_t)P * in_stride + c];
  }
}

int launch_copy_view(const View& in, const View& out, hipStream_t s) {
  const long long pixels = (long long)in.B * in.H * in.W;
  hipLaunchKernelGGL(copy_view_kernel, dim3(grid_for(pixels * in.C)), dim3(256), 0, s, in.p + in.coff,
                     out.p + out.coff, pixels, in.C, in.cstride, out.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// 32x32 LDS-tiled transpose between [B][HW][C(strided)] and [B][C][HW]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C,
                                    int in_stride) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    t[r][tx] = (p < HW && c < C) ? in[((size_t)b * HW + p) * in_stride + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = t[tx][r];
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C,
                                    int out_stride) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    t[r][tx] = (p < HW && c < C) ? in[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) out[((size_t)b * HW + p) * out_stride + c] = t[tx][r];
  }
}

int launch_nhwc_to_nchw(const View& in, float* out_nchw, hipStream_t s) {
  const int HW = in.H * in.W;
  dim3 g((HW + 31) / 32, (in.C + 31) / 32, in.B);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, g, dim3(256), 0, s, in.p + in.coff, out_nchw, HW, in.C, in.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_nchw_to_nhwc(const float* in_nchw, const View& out, hipStream_t s) {
  const int HW = out.H * out.W;
  dim3 g((HW + 31) / 32, (out.C + 31) / 32, out.B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, g, dim3(256), 0, s, in_nchw, out.p + out.coff, HW, out.C, out.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
