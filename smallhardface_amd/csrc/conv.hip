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
#include <cstdio>
#include <cstdlib>

#include "conv_common.h"
#include "shf_internal.h"

namespace shf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;   // input channels per K chunk
constexpr int LDK = 36;  // LDS row pitch in floats (32 + 4 pad: conflict-free ds_read_b128)

// DIL is a template parameter (1/2/4 in the detector, 0 = "1x1, no halo") so the halo tile
// size is a compile-time constant: its loads are fully unrolled, all issued back to back
// into registers UNDER the last tap of the previous chunk, and written to LDS after the
// end-of-chunk barrier (a runtime-bound loop made the compiler serialise load->wait->ds_write
// round trips: 6-12 exposed memory latencies per chunk).
template <int KS, int DIL, int BN, int TH, int TW>
__global__ __launch_bounds__(256, 1) void conv_mfma_f32_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TAPS = KS * KS;
  constexpr int WN = BN / 64;  // waves along cout
  constexpr int WM = 4 / WN;   // waves along pixels
  static_assert(TH == 4 * WM && TW == 16, "tile shape");
  constexpr int PAD = (KS == 3) ? DIL : 0;
  constexpr int HTW = TW + 2 * PAD, HTH = TH + 2 * PAD;
  constexpr int HP = HTH * HTW;
  constexpr int ALD = (HP * 8 + 255) / 256;  // float4 halo pieces per thread
  constexpr int BLD = BN * 8 / 256;          // float4 weight pieces per thread
  float* As = smem;              // [HP][LDK]
  float* Bs = smem + HP * LDK;   // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int bid = blockIdx.x;
  const int ct = bid % p.nct;
  int pt = bid / p.nct;
  const int mi = conv_find_member(p, pt);
  const ConvMember& mem = p.m[mi];
  pt -= mem.tile_start;
  int b, ty_, tx_;
  conv_split_tile(mem, pt, b, ty_, tx_);
  const int ty0 = ty_ * TH, tx0 = tx_ * TW;
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
  // named scalars, not an array: the staged weights are loop-carried (loaded in step s, parked in
  // LDS in step s+1) and hipcc keeps a loop-carried local ARRAY in scratch memory
  float4 bq0, bq1, bq2 = make_float4(0.f, 0.f, 0.f, 0.f), bq3 = make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(BLD == 2 || BLD == 4, "weight staging");
  float4 areg[ALD];
  const int brow = tid >> 3, bq = tid & 7;

  // per-thread halo piece geometry (chunk-invariant): global offset or -1 when outside the image
  int a_goff[ALD];  // element offsets fit 32 bits (largest activation: 1408^2 x 64 floats)
  int a_loff[ALD];
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    const int idx = tid + 256 * j;
    const int hp = idx >> 3, q = idx & 7;
    const int hy = hp / HTW, hx = hp - hy * HTW;
    const int gy = ty0 - PAD + hy, gx = tx0 - PAD + hx;
    const bool in = (idx < HP * 8) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
    a_goff[j] = in ? ((b * H + gy) * W + gx) * p.in_stride + q * 4 : -1;
    a_loff[j] = (idx < HP * 8) ? hp * LDK + q * 4 : -1;
  }
  // prologue: halo(0) and B(0) into LDS, B(1) in flight in registers
  {
    const float* inc_ = gin + (0) * KC;
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      areg[j] = (a_goff[j] >= 0) ? *(const float4*)(inc_ + a_goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    const float* wn_ = wbase + (size_t)((p.relu & 2) ? 0 : (0)) * wstep + (size_t)brow * KC + bq * 4;
    bq0 = *(const float4*)(wn_);
    bq1 = *(const float4*)(wn_ + 32 * KC);
    if constexpr (BLD == 4) {
      bq2 = *(const float4*)(wn_ + 64 * KC);
      bq3 = *(const float4*)(wn_ + 96 * KC);
    }
  }
  {
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      if (a_loff[j] >= 0) *(float4*)(As + a_loff[j]) = areg[j];
  }
  {
    float* Bn_ = Bs + (0) * (BN * LDK) + brow * LDK + bq * 4;
    *(float4*)(Bn_) = bq0;
    *(float4*)(Bn_ + 32 * LDK) = bq1;
    if constexpr (BLD == 4) {
      *(float4*)(Bn_ + 64 * LDK) = bq2;
      *(float4*)(Bn_ + 96 * LDK) = bq3;
    }
  }
  if (S > 1) {
    const float* wn_ = wbase + (size_t)((p.relu & 2) ? 0 : (1)) * wstep + (size_t)brow * KC + bq * 4;
    bq0 = *(const float4*)(wn_);
    bq1 = *(const float4*)(wn_ + 32 * KC);
    if constexpr (BLD == 4) {
      bq2 = *(const float4*)(wn_ + 64 * KC);
      bq3 = *(const float4*)(wn_ + 96 * KC);
    }
  }

  int c = 0, tap = 0;
#ifdef SHF_CONV_TIMING
  unsigned long long tb = 0, ti = 0, tc = 0, tx = 0, t0, t1, t2, t3;
#define SHF_T(x) x = __builtin_amdgcn_s_memtime()
#else
#define SHF_T(x)
#endif
  for (int s = 0; s < S; ++s) {
    SHF_T(t0);
    __syncthreads();  // B(s) [and the halo of chunk c] visible; everyone is done with step s-1
    SHF_T(t1);
    // B(s+1) was loaded during step s-1: park it in the other buffer now (free since the barrier),
    // then start B(s+2); nothing but the barrier is left after the MFMAs of this step.
    if (s + 1 < S) {
    float* Bn_ = Bs + ((s + 1) & 1) * (BN * LDK) + brow * LDK + bq * 4;
    *(float4*)(Bn_) = bq0;
    *(float4*)(Bn_ + 32 * LDK) = bq1;
    if constexpr (BLD == 4) {
      *(float4*)(Bn_ + 64 * LDK) = bq2;
      *(float4*)(Bn_ + 96 * LDK) = bq3;
    }
  }
    if (s + 2 < S && !(p.relu & 4)) {
    const float* wn_ = wbase + (size_t)((p.relu & 2) ? 0 : (s + 2)) * wstep + (size_t)brow * KC + bq * 4;
    bq0 = *(const float4*)(wn_);
    bq1 = *(const float4*)(wn_ + 32 * KC);
    if constexpr (BLD == 4) {
      bq2 = *(const float4*)(wn_ + 64 * KC);
      bq3 = *(const float4*)(wn_ + 96 * KC);
    }
  }
    const bool last_tap = (tap == TAPS - 1);
    const bool more_chunks = (c + 1 < nchunks);
    if (last_tap && more_chunks) {
    const float* inc_ = gin + (c + 1) * KC;
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      areg[j] = (a_goff[j] >= 0) ? *(const float4*)(inc_ + a_goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
    SHF_T(t2);
    const int ky = (KS == 3) ? tap / 3 : 0, kx = (KS == 3) ? tap - ky * 3 : 0;
    const float* Ap = As + (ky * DIL * HTW + kx * DIL) * LDK;
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
#ifdef SHF_CONV_TIMING
    asm volatile("s_nop 0" ::: "memory");
    SHF_T(t3);
    tb += t1 - t0; ti += t2 - t1; tc += t3 - t2;
#endif
    if (last_tap) {
      if (more_chunks) {
        __syncthreads();  // every wave is done reading the halo tile of chunk c
        {
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      if (a_loff[j] >= 0) *(float4*)(As + a_loff[j]) = areg[j];
  }
      }
      tap = 0;
      ++c;
    } else {
      ++tap;
    }
#ifdef SHF_CONV_TIMING
    SHF_T(t0);
    tx += t0 - t3;
#endif
  }
#ifdef SHF_CONV_TIMING
  if (p.dbg && lane == 0 && (bid == 0 || bid == 300)) {
    unsigned long long* d = p.dbg + ((bid ? 1 : 0) * 4 + wave) * 5;
    d[0] = tb; d[1] = ti; d[2] = tc; d[3] = tx; d[4] = S;
  }
#endif

  // epilogue: C row = (r&3) + 8*(r>>2) + 4*(lane>>5), C col = lane&31
  float amax = 0.f;  // split-fp16 modes only (a layer the split kernels do not cover): range guard + activation exponent
  const bool track = p.range_flag || mem.out_amax || mem.pool_amax;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int cout = ct * BN + wn * 64 + tn * 32 + i;
    const float bv = p.bias ? p.bias[cout] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const f32x16 a = acc[tm][tn];
      conv_store_tile([&](int r) { return a[r]; }, bv, p.relu, ty0 + wm * 4 + tm * 2, tx0, kh, H, W, b, cout, gout,
                      p.out_stride, mem.pool, p.pool_stride, track ? &amax : nullptr);
    }
  }
  if (track) {
    conv_raise_range_flag(p.range_flag, amax);
    conv_publish_amax(mem.out_amax, mem.pool_amax, amax);
  }
}

// ---------------------------------------------------------------------------
// First layer (tiny Cin, HBM-bound): NCHW input -> NHWC output, direct FMA.
// ---------------------------------------------------------------------------
// lane = pixel, wave = group of 16 output channels: the weights of a wave are uniform, so hipcc
// keeps them in SGPRs (s_load + v_fma with a scalar operand: no LDS traffic at all); the 64x64
// output tile is transposed through LDS so that every store instruction writes whole 256-B pixels.
// (A variant with the detector's conv1_1 geometry as compile-time constants -- 27 fully unrolled taps --
// measured SLOWER: 2.04 ms against 1.25 ms per image; the rolled loops keep the scalar loads in flight.)
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int B, int H, int W, int Cin, int Cout, int k, int dil,
                                                          int pad, int relu, int out_stride, int* range_flag,
                                                          unsigned* out_amax) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [64][Cout + 4]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pitch = Cout + 4;
  const int ngroups = Cout / 16;
  const int c4n = Cout / 4;
  const long long total = (long long)B * H * W;
  float amax = 0.f;  // split-fp16 mode: the consumer splits these outputs to fp16 hi/lo (range guard, conv_common.h)
  for (long long P0 = (long long)blockIdx.x * 64; P0 < total; P0 += (long long)gridDim.x * 64) {
    const long long P = P0 + lane;
    const bool valid = P < total;
    const int x = (int)(P % W);
    const int y = (int)((P / W) % H);
    const int b = (int)(P / ((long long)W * H));
    for (int g = wave; g < ngroups; g += 4) {
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[g * 16 + j] : 0.f;
#pragma unroll
      for (int c = 0; c < Cin; ++c)
#pragma unroll
        for (int ky = 0; ky < k; ++ky) {
          const int iy = y - pad + ky * dil;
#pragma unroll
          for (int kx = 0; kx < k; ++kx) {
            const int ix = x - pad + kx * dil;
            float v = 0.f;
            if (valid && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
              v = in[((size_t)(b * Cin + c) * H + iy) * W + ix];
            const float* wr = wt + (size_t)((c * k + ky) * k + kx) * Cout + g * 16;  // wave-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
          }
        }
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        float4 v = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        amax = fmaxf(fmaxf(fmaxf(amax, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
        *(float4*)(tile + lane * pitch + g * 16 + j) = v;
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * c4n; idx += 256) {
      const int pix = idx / c4n, c4 = idx - pix * c4n;
      if (P0 + pix < total)
        *(float4*)(out + (size_t)(P0 + pix) * out_stride + c4 * 4) = *(const float4*)(tile + pix * pitch + c4 * 4);
    }
    __syncthreads();
  }
  conv_raise_range_flag(range_flag, amax);
  conv_publish_amax(out_amax, nullptr, amax);
}

// ---------------------------------------------------------------------------
// Generic direct convolution (any shape, stride 1), one thread per output value.
// Only used for layer shapes the MFMA kernel does not cover.
// ---------------------------------------------------------------------------
__global__ void conv_direct_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ out, int B, int H, int W,
                                   int Cin, int Cout, int k, int dil, int pad, int relu, int in_stride,
                                   int out_stride, int* range_flag, unsigned* out_amax) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * Cout;
  float amax = 0.f;
  if (n < total) {  // (no early return: every lane takes part in the wave reduction below)
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
  amax = acc != acc ? __builtin_inff() : fabsf(acc);
  }
  conv_raise_range_flag(range_flag, amax);   // split-fp16 modes: the consumer may be a split kernel (conv_common.h)
  conv_publish_amax(out_amax, nullptr, amax);
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
  const bool same = (k == 3 && pad == dil && (dil == 1 || dil == 2 || dil == 4)) || (k == 1 && pad == 0);
  if (same && Cin % KC == 0 && Cout % 64 == 0) return 0;
  return 2;
}

static size_t mfma_lds_bytes(int k, int dil, int BN, int TH, int TW) {
  const int pad = k == 3 ? dil : 0;
  return ((size_t)(TH + 2 * pad) * (TW + 2 * pad) * LDK + 2 * (size_t)BN * LDK) * sizeof(float);
}

template <int KS, int DIL, int BN, int TH, int TW>
static int launch_mfma_t(const ConvArgs* as, int n, hipStream_t s) {
  const ConvArgs& a = as[0];
  ConvK p = {};
  p.wp = a.wpacked;
  p.bias = a.bias;
  p.Cin = a.in.C; p.Cout = a.out.C;
  p.in_stride = a.in.cstride; p.out_stride = a.out.cstride;
  p.dil = a.dil; p.relu = a.relu | (a.pool.p && !a.write_main ? 8 : 0);
  p.pool_stride = a.pool.p ? a.pool.cstride : 0;
  p.nct = p.Cout / BN;
  p.nmem = n;
  for (int i = 0; i < MAX_GROUP; ++i) p.tile_starts[i] = 0x7fffffff;
  p.w1t = nullptr;
  p.w1f = nullptr;
  p.b1 = nullptr;
  p.range_flag = a.range_flag;
  p.wph = nullptr;
  p.wscale_inv = 1.f;
  p.tile_base = 0;
  p.ntile_blocks = 0;
  p.pc_tab = 0;
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
    m.pool = q.pool.p ? q.pool.p + q.pool.coff : nullptr;
    m.img = nullptr;
    m.in_amax = q.in_amax; m.out_amax = q.out_amax; m.pool_amax = q.pool.p ? q.pool_amax : nullptr;
    m.B = q.in.B; m.H = q.in.H; m.W = q.in.W;
    m.tiles_x = (m.W + TW - 1) / TW;
    m.tiles_per_img = m.tiles_x * ((m.H + TH - 1) / TH);
    m.inv_tiles_x = conv_inv32(m.tiles_x);
    m.inv_tiles_per_img = conv_inv32(m.tiles_per_img);
    m.tile_start = (int)tiles;
    p.tile_starts[i] = (int)tiles;
    tiles += (long long)m.tiles_per_img * m.B;
    // conv_split_tile's multiply-high quotients are exact while tile index x divisor < 2^32
    if ((unsigned long long)m.tiles_per_img * m.B * (unsigned long long)m.tiles_per_img >= (1ull << 32)) {
      set_error("conv: more than 2^32 / tiles-per-image pixel tiles in one member (shrink the batch or the map)");
      return -1;
    }
  }
  if (tiles * p.nct >= (1ll << 31)) { set_error("conv: grid too large"); return -1; }
  p.dbg = nullptr;
#ifdef SHF_CONV_TIMING
  static unsigned long long* dbg_dev = nullptr;
  if (!dbg_dev) hipMalloc((void**)&dbg_dev, 8 * 5 * 8);
  hipMemset(dbg_dev, 0, 8 * 5 * 8);
  p.dbg = dbg_dev;
#endif
  size_t lds = mfma_lds_bytes(KS, DIL, BN, TH, TW);
  const long long blocks = tiles * p.nct;
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, DIL, BN, TH, TW>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  SHF_HIP_OK(hipGetLastError());
#ifdef SHF_CONV_TIMING
  {
    unsigned long long h[40];
    hipStreamSynchronize(s);
    hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w)
      if (h[w * 5 + 4])
        fprintf(stderr, "[conv timing] blk%d wave%d steps %llu: per-step cycles barrier %.0f issue %.0f compute %.0f tail %.0f\n",
                w / 4 ? 300 : 0, w % 4, h[w * 5 + 4], (double)h[w * 5] / h[w * 5 + 4], (double)h[w * 5 + 1] / h[w * 5 + 4],
                (double)h[w * 5 + 2] / h[w * 5 + 4], (double)h[w * 5 + 3] / h[w * 5 + 4]);
  }
#endif
  return 0;
}

template <int KS, int DIL, int BN, int TH, int TW>
static int set_attr_t() {
  SHF_HIP_OK(hipFuncSetAttribute((const void*)conv_mfma_f32_kernel<KS, DIL, BN, TH, TW>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return 0;
}

int conv_init_attributes() {
  if (set_attr_t<3, 1, 128, 8, 16>() || set_attr_t<3, 2, 128, 8, 16>() || set_attr_t<3, 4, 128, 8, 16>() ||
      set_attr_t<3, 1, 64, 16, 16>() || set_attr_t<3, 2, 64, 16, 16>() || set_attr_t<3, 4, 64, 16, 16>() ||
      set_attr_t<1, 0, 128, 8, 16>() || set_attr_t<1, 0, 64, 16, 16>())
    return -1;
  return 0;
}

int launch_conv_mfma_group(const ConvArgs* as, int n, hipStream_t s) {
  if (n < 1 || n > MAX_GROUP) { set_error("conv group: 1..16 members"); return -1; }
  for (int i = 0; i < n; ++i)
    if ((as[i].in.cstride % 4) || (as[i].in.coff % 4)) { set_error("conv: input view not 16-byte aligned"); return -1; }
  const ConvArgs& a = as[0];
  const bool bn128 = (a.out.C % 128 == 0);
  if (a.k == 1) return bn128 ? launch_mfma_t<1, 0, 128, 8, 16>(as, n, s) : launch_mfma_t<1, 0, 64, 16, 16>(as, n, s);
  switch (a.dil) {
    case 1: return bn128 ? launch_mfma_t<3, 1, 128, 8, 16>(as, n, s) : launch_mfma_t<3, 1, 64, 16, 16>(as, n, s);
    case 2: return bn128 ? launch_mfma_t<3, 2, 128, 8, 16>(as, n, s) : launch_mfma_t<3, 2, 64, 16, 16>(as, n, s);
    case 4: return bn128 ? launch_mfma_t<3, 4, 128, 8, 16>(as, n, s) : launch_mfma_t<3, 4, 64, 16, 16>(as, n, s);
    default: set_error("conv: the MFMA kernel is instantiated for dilation 1, 2 and 4"); return -1;
  }
}

int launch_conv_mfma(const ConvArgs& a, hipStream_t s) { return launch_conv_mfma_group(&a, 1, s); }

int launch_conv_first(const float* in_nchw, const ConvArgs& a, hipStream_t s) {
  const int Cin = a.in.C, Cout = a.out.C;
  if (Cout % 16 || !a.wfirst) { set_error("conv_first: Cout must be a multiple of 16 (transposed weights required)"); return -1; }
  const long long total = (long long)a.in.B * a.in.H * a.in.W;
  const size_t lds = (size_t)64 * (Cout + 4) * sizeof(float);
  long long blocks = (total + 63) / 64;
  if (blocks > 256 * 8) blocks = 256 * 8;  // persistent blocks, grid-stride over 64-pixel groups
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)blocks), dim3(256), lds, s, in_nchw, a.wfirst, a.bias,
                     a.out.p + a.out.coff, a.in.B, a.in.H, a.in.W, Cin, Cout, a.k, a.dil, a.pad, a.relu,
                     a.out.cstride, a.range_flag, a.out_amax);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
  const long long total = (long long)a.in.B * a.in.H * a.in.W * a.out.C;
  hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     a.in.p + a.in.coff, a.wraw, a.bias, a.out.p + a.out.coff, a.in.B, a.in.H, a.in.W, a.in.C,
                     a.out.C, a.k, a.dil, a.pad, a.relu, a.in.cstride, a.out.cstride, a.range_flag, a.out_amax);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
