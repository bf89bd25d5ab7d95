// Convolution kernels for gfx950 (MI355X).
//
// conv_mfma_f32_kernel: im2col-free implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32; exact fp32 == an fmaf chain, 157 TFLOP/s chip peak).
// Replaces the reference's per-image im2col + SGEMM + rank-1 bias GEMM + ReLU
// (caffe/src/caffe/layers/base_conv_layer.cpp:256-279,326-348, util/im2col.cu:9-62,
// cudnn_conv_layer.cu:11-46, relu_layer.cu:9-15) with one launch:
//   * activations NHWC, so the K (=cin) run of every pixel is one contiguous line;
//   * a (TH+2p)x(TW+2p) input halo tile of 32 channels is staged in LDS once and
//     reused by all k*k taps (dilation only changes the halo size / tap offsets);
//   * weights are pre-packed [cin/32][tap][cout][32] so every (chunk,tap) step of a
//     block is ONE contiguous BNx32 slab; slabs are double-buffered in LDS while the
//     64 MFMAs/wave of the current step run;
//   * bias + ReLU fused in the accumulator epilogue, stores are 128-B cout runs;
//   * block id = pixel_tile * n_cout_tiles + cout_tile, so the blocks an XCD sees
//     (id % 8) share the same weight slab in that XCD's private L2.
#include "shf_internal.h"

namespace shf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;   // input channels per K chunk
constexpr int LDK = 36;  // LDS row pitch in floats (32 + 4 pad: conflict-free ds_read_b128)

constexpr int MAX_GROUP = 16;

// One launch covers a GROUP of independent problems that share the layer (weights,
// channels, dilation) but not the spatial size: the 10 (level, flip) units of an image's
// test pyramid go through each conv layer in ONE grid, so the latency-bound small levels
// ride along with the large ones instead of paying a whole K loop on a mostly idle chip.
struct ConvMember {
  const float* in;   // already offset to the view's first channel
  float* out;        // already offset to the view's first channel
  int B, H, W;
  int tiles_x, tiles_per_img, tile_start;  // tile_start: first pixel-tile index of this member
};

struct ConvK {
  const float* wp;   // packed weights
  const float* bias;
  int Cin, Cout;
  int in_stride, out_stride;
  int dil, relu;
  int nct, nmem;
  ConvMember m[MAX_GROUP];
};

// row i (0..31) of a 32-row MFMA tile -> pixel inside the wave's 2x16 strip.
// The two low bits walk a 2x2 window so that the 4 consecutive C rows a lane owns
// form one pooling window (kept for a fused 2x2 max-pool epilogue).
__device__ __forceinline__ void row_to_pixel(int i, int& dy, int& px) {
  dy = (i >> 1) & 1;
  px = ((i >> 2) << 1) | (i & 1);
}

template <int KS, int BN, int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TAPS = KS * KS;
  constexpr int WN = BN / 64;  // waves along cout
  constexpr int WM = 4 / WN;   // waves along pixels
  static_assert(TH == 4 * WM && TW == 16, "tile shape");
  const int pad = (KS == 3) ? p.dil : 0;
  const int HTW = TW + 2 * pad, HTH = TH + 2 * pad;
  const int HP = HTH * HTW;
  float* As = smem;              // [HP][LDK]
  float* Bs = smem + HP * LDK;   // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int bid = blockIdx.x;
  const int ct = bid % p.nct;
  int pt = bid / p.nct;
  int mi = 0;
#pragma unroll 1
  for (int q = 1; q < p.nmem; ++q)
    if (pt >= p.m[q].tile_start) mi = q;
  const ConvMember& mem = p.m[mi];
  pt -= mem.tile_start;
  const int b = pt / mem.tiles_per_img;
  pt -= b * mem.tiles_per_img;
  const int ty0 = (pt / mem.tiles_x) * TH, tx0 = (pt % mem.tiles_x) * TW;
  const int H = mem.H, W = mem.W;
  const float* __restrict__ gin = mem.in;
  float* __restrict__ gout = mem.out;

  const int i = lane & 31, kh = lane >> 5;
  int dy, px;
  row_to_pixel(i, dy, px);
  int a_off[2], b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_off[t] = ((wm * 4 + t * 2 + dy) * HTW + px) * LDK + kh * 4;
    b_off[t] = (wn * 64 + t * 32 + i) * LDK + kh * 4;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  const int nchunks = p.Cin / KC;
  const int S = nchunks * TAPS;
  const size_t wstep = (size_t)p.Cout * KC;
  const float* wbase = p.wp + (size_t)ct * BN * KC;
  constexpr int BLD = BN * 8 / 256;
  float4 breg[BLD];
  const int brow = tid >> 3, bq = tid & 7;

#pragma unroll
  for (int j = 0; j < BLD; ++j) breg[j] = *(const float4*)(wbase + (size_t)(brow + 32 * j) * KC + bq * 4);
#pragma unroll
  for (int j = 0; j < BLD; ++j) *(float4*)(Bs + (brow + 32 * j) * LDK + bq * 4) = breg[j];

  for (int s = 0; s < S; ++s) {
    const int c = s / TAPS, tap = s - c * TAPS;
    if (tap == 0) {
      const float* inc = gin + c * KC;
      for (int idx = tid; idx < HP * 8; idx += 256) {
        const int hp = idx >> 3, q = idx & 7;
        const int hy = hp / HTW, hx = hp - hy * HTW;
        const int gy = ty0 - pad + hy, gx = tx0 - pad + hx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
          v = *(const float4*)(inc + ((size_t)(b * H + gy) * W + gx) * p.in_stride + q * 4);
        *(float4*)(As + hp * LDK + q * 4) = v;
      }
    }
    __syncthreads();
    if (s + 1 < S) {
      const float* wn_ = wbase + (size_t)(s + 1) * wstep;
#pragma unroll
      for (int j = 0; j < BLD; ++j) breg[j] = *(const float4*)(wn_ + (size_t)(brow + 32 * j) * KC + bq * 4);
    }
    const int ky = (KS == 3) ? tap / 3 : 0, kx = (KS == 3) ? tap - ky * 3 : 0;
    const float* Ap = As + (ky * p.dil * HTW + kx * p.dil) * LDK;
    const float* Bp = Bs + (s & 1) * (BN * LDK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 a0 = *(const f32x4*)(Ap + a_off[0] + kk * 8);
      const f32x4 a1 = *(const f32x4*)(Ap + a_off[1] + kk * 8);
      const f32x4 b0 = *(const f32x4*)(Bp + b_off[0] + kk * 8);
      const f32x4 b1 = *(const f32x4*)(Bp + b_off[1] + kk * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
      }
    }
    if (s + 1 < S) {
      float* Bn = Bs + ((s + 1) & 1) * (BN * LDK);
#pragma unroll
      for (int j = 0; j < BLD; ++j) *(float4*)(Bn + (brow + 32 * j) * LDK + bq * 4) = breg[j];
    }
    if (tap == TAPS - 1) __syncthreads();  // the halo tile is rewritten next step
  }

  // epilogue: C row = (r&3) + 8*(r>>2) + 4*(lane>>5), C col = lane&31
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int cout = ct * BN + wn * 64 + tn * 32 + i;
    const float bv = p.bias ? p.bias[cout] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        int dy2, px2;
        row_to_pixel(row, dy2, px2);
        const int gy = ty0 + wm * 4 + tm * 2 + dy2, gx = tx0 + px2;
        if (gy < H && gx < W) {
          float v = acc[tm][tn][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          gout[((size_t)(b * H + gy) * W + gx) * p.out_stride + cout] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// First layer (tiny Cin, HBM-bound): NCHW input -> NHWC output, direct FMA.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int B, int H, int W, int Cin, int Cout, int k, int dil,
                                                          int pad, int relu, int out_stride) {
  extern __shared__ __attribute__((aligned(16))) float ws[];  // [Cin*k*k][Cout]
  const int K = Cin * k * k;
  for (int idx = threadIdx.x; idx < K * Cout; idx += 256) {
    const int co = idx / K, r = idx - co * K;  // w is (Cout, Cin*k*k)
    ws[r * Cout + co] = w[idx];
  }
  __syncthreads();
  const int groups = Cout / 16;
  const int ppb = 256 / groups;
  const int cg = threadIdx.x % groups;
  const long long P = (long long)blockIdx.x * ppb + threadIdx.x / groups;
  const long long total = (long long)B * H * W;
  if (threadIdx.x / groups >= ppb || P >= total) return;
  const int x = (int)(P % W);
  const int y = (int)((P / W) % H);
  const int b = (int)(P / ((long long)W * H));
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[cg * 16 + j] : 0.f;
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < k; ++ky) {
      const int iy = y - pad + ky * dil;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = x - pad + kx * dil;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          v = in[((size_t)(b * Cin + c) * H + iy) * W + ix];
        const float* wr = ws + ((c * k + ky) * k + kx) * Cout + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
      }
    }
  float* o = out + (size_t)P * out_stride + cg * 16;
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
    float4 v = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *(float4*)(o + j) = v;
  }
}

// ---------------------------------------------------------------------------
// Generic direct convolution (any shape, stride 1), one thread per output value.
// Only used for layer shapes the MFMA kernel does not cover.
// ---------------------------------------------------------------------------
__global__ void conv_direct_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W,
                                   int Cin, int Cout, int k, int dil, int pad, int relu, int in_stride,
                                   int out_stride) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * Cout;
  if (n >= total) return;
  const int co = (int)(n % Cout);
  const long long P = n / Cout;
  const int x = (int)(P % W), y = (int)((P / W) % H), b = (int)(P / ((long long)W * H));
  float acc = bias ? bias[co] : 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = y - pad + ky * dil;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = x - pad + kx * dil;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float* ip = in + ((size_t)(b * H + iy) * W + ix) * in_stride;
      const float* wp = w + ((size_t)co * Cin * k + ky) * k + kx;
      for (int c = 0; c < Cin; ++c) acc = fmaf(ip[c], wp[(size_t)c * k * k], acc);
    }
  }
  if (relu) acc = fmaxf(acc, 0.f);
  out[(size_t)P * out_stride + co] = acc;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
size_t packed_conv_weight_floats(int Cout, int Cin, int k) { return (size_t)Cout * Cin * k * k; }

void pack_conv_weights(const float* w, int Cout, int Cin, int k, float* dst) {
  // (Cout,Cin,k,k) -> [Cin/32][k*k][Cout][32]
  const int taps = k * k;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const size_t d = (((size_t)(ci / KC) * taps + t) * Cout + co) * KC + (ci % KC);
        dst[d] = w[((size_t)co * Cin + ci) * taps + t];
      }
}

int conv_kernel_class(int Cin, int Cout, int k, int pad, int dil, bool in_nchw) {
  if (in_nchw) return 1;
  const bool same = (k == 3 && pad == dil) || (k == 1 && pad == 0);
  if (same && Cin % KC == 0 && Cout % 64 == 0) return 0;
  return 2;
}

static size_t mfma_lds_bytes(int k, int dil, int BN, int TH, int TW) {
  const int pad = k == 3 ? dil : 0;
  return ((size_t)(TH + 2 * pad) * (TW + 2 * pad) * LDK + 2 * (size_t)BN * LDK) * sizeof(float);
}

template <int KS, int BN, int TH, int TW>
static int launch_mfma_t(const ConvArgs* as, int n, hipStream_t s) {
  const ConvArgs& a = as[0];
  ConvK p;
  p.wp = a.wpacked;
  p.bias = a.bias;
  p.Cin = a.in.C; p.Cout = a.out.C;
  p.in_stride = a.in.cstride; p.out_stride = a.out.cstride;
  p.dil = a.dil; p.relu = a.relu;
  p.nct = p.Cout / BN;
  p.nmem = n;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    const ConvArgs& q = as[i];
    if (q.in.C != p.Cin || q.out.C != p.Cout || q.in.cstride != p.in_stride || q.out.cstride != p.out_stride ||
        q.dil != p.dil || q.wpacked != p.wp) {
      set_error("conv group: members must share the layer");
      return -1;
    }
    ConvMember& m = p.m[i];
    m.in = q.in.p + q.in.coff;
    m.out = q.out.p + q.out.coff;
    m.B = q.in.B; m.H = q.in.H; m.W = q.in.W;
    m.tiles_x = (m.W + TW - 1) / TW;
    m.tiles_per_img = m.tiles_x * ((m.H + TH - 1) / TH);
    m.tile_start = (int)tiles;
    tiles += (long long)m.tiles_per_img * m.B;
  }
  const size_t lds = mfma_lds_bytes(KS, a.dil, BN, TH, TW);
  if (lds > 160 * 1024) { set_error("conv: dilation too large for the LDS halo tile"); return -1; }
  const long long blocks = tiles * p.nct;
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, BN, TH, TW>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int conv_init_attributes() {
  const int maxlds = 160 * 1024;
  SHF_HIP_OK(hipFuncSetAttribute((const void*)conv_mfma_f32_kernel<3, 128, 8, 16>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
  SHF_HIP_OK(hipFuncSetAttribute((const void*)conv_mfma_f32_kernel<3, 64, 16, 16>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
  SHF_HIP_OK(hipFuncSetAttribute((const void*)conv_mfma_f32_kernel<1, 128, 8, 16>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
  SHF_HIP_OK(hipFuncSetAttribute((const void*)conv_mfma_f32_kernel<1, 64, 16, 16>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
  return 0;
}

int launch_conv_mfma_group(const ConvArgs* as, int n, hipStream_t s) {
  if (n < 1 || n > MAX_GROUP) { set_error("conv group: 1..16 members"); return -1; }
  for (int i = 0; i < n; ++i)
    if ((as[i].in.cstride % 4) || (as[i].in.coff % 4)) { set_error("conv: input view not 16-byte aligned"); return -1; }
  const ConvArgs& a = as[0];
  const bool bn128 = (a.out.C % 128 == 0);
  if (a.k == 3) return bn128 ? launch_mfma_t<3, 128, 8, 16>(as, n, s) : launch_mfma_t<3, 64, 16, 16>(as, n, s);
  return bn128 ? launch_mfma_t<1, 128, 8, 16>(as, n, s) : launch_mfma_t<1, 64, 16, 16>(as, n, s);
}

int launch_conv_mfma(const ConvArgs& a, hipStream_t s) { return launch_conv_mfma_group(&a, 1, s); }

int launch_conv_first(const float* in_nchw, const ConvArgs& a, hipStream_t s) {
  const int Cin = a.in.C, Cout = a.out.C;
  if (Cout % 16 || 256 % (Cout / 16)) { set_error("conv_first: unsupported Cout"); return -1; }
  const int ppb = 256 / (Cout / 16);
  const long long total = (long long)a.in.B * a.in.H * a.in.W;
  const size_t lds = (size_t)Cin * a.k * a.k * Cout * sizeof(float);
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)((total + ppb - 1) / ppb)), dim3(256), lds, s, in_nchw,
                     a.wraw, a.bias, a.out.p + a.out.coff, a.in.B, a.in.H, a.in.W, Cin, Cout, a.k, a.dil, a.pad,
                     a.relu, a.out.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
  const long long total = (long long)a.in.B * a.in.H * a.in.W * a.out.C;
  hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     a.in.p + a.in.coff, a.wraw, a.bias, a.out.p + a.out.coff, a.in.B, a.in.H, a.in.W, a.in.C,
                     a.out.C, a.k, a.dil, a.pad, a.relu, a.in.cstride, a.out.cstride);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
