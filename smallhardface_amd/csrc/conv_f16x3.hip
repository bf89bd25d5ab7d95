// Split-fp16 implicit-GEMM convolution for gfx950 (MI355X): fp32-class accuracy at the
// fp16 matrix-core rate.
//
// Every fp32 operand x is split as  x = hi + lo * 2^-11,  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
// (relative representation error 2^-24, i.e. fp32-grade), and a product is formed from three
// fp16 MFMAs accumulating in fp32:
//      main += a_hi * b_hi
//      corr += a_hi * b_lo + a_lo * b_hi            (both carry the 2^11 scale)
//      out   = main + corr * 2^-11                   (a_lo*b_lo ~ 2^-22 relative is dropped)
// v_mfma_f32_32x32x16_f16 retires 16 k per 32 cycles against 2 k per 64 cycles for the exact
// v_mfma_f32_32x32x2_f32, so three of them are 16/3 = 5.3x the fp32 MFMA rate.  Activations keep 4 B per
// element in HBM -- fp32, split while the halo tile is staged into LDS, or already split by the producer's
// epilogue (ConvArgs::in_split / out_split) -- and weights are split once at load time.  Every MFMA convolution
// of the detector runs here in the split-fp16 modes: 3x3 at dilation 1, 2, 4 and the 1x1s.
//
// Kernels: conv_mfma_f16x3_w4d_kernel (the dual-tile 4-wave family: Cin >= 128, Cout % 128 == 0 -- 70 % of the time),
// conv_mfma_f16x3_pc_kernel (fused first pair conv1_1 -> conv1_2, producer/consumer waves) and the 8-wave
// conv_mfma_f16x3_kernel below (Cin 64, dilated heads, 1x1).  Common structure: tile 256 px (16x16) x BN couts; a
// STAGE is one kernel row (3 taps) of one 32-channel chunk: its three BNx32 weight slabs are double-buffered in
// LDS and arrive by LDS DMA; the 18x18x32 halo tile is staged once per chunk and reused by all 9 taps.  LDS rows
// are [hi: 32 halfs][lo: 32 halfs][16 B pad] = 144 B (conflict-free ds_read_b128 over consecutive rows).
// (The dual-tile family has its own geometry: 16-channel chunks, 64 / 80-byte rows, unscaled low parts -- see its header.)
// Epilogues: the 4-wave kernels run the MFMA as D[cout][pixel] and store from registers (conv_epilogue_regs1); the
// 8-wave and fused-pair kernels run D[pixel][cout] and transpose the tile through LDS (conv_stage_tile / conv_flush_tile).
// (The single-tile two-accumulator 4-wave kernel the family replaced, its persistent form and the timing-only
// F16X3_EXPERIMENT_* builds of round 2 are in the history at c610267; DESIGN.md keeps what they measured.)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "conv_common.h"

namespace shf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef F16X3_DMA_LATE
#define F16X3_DMA_LATE 1   // 1: the early-finishing waves 0-3 issue the next stage's weight DMA after their MFMAs
#endif
#ifndef F16X3_CONV_MID
#define F16X3_CONV_MID 1   // 1: split the next halo tile to fp16 hi/lo in the middle of the last stage's MFMAs
#endif

namespace f16x3 {
constexpr int KC = 32;      // input channels per chunk
constexpr int ROWB = 144;   // bytes per LDS row (pixel or cout)
constexpr int TH = 16, TW = 16, HTW = TW + 2, HTH = TH + 2, HP = HTH * HTW;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
}  // namespace f16x3

// conv1_1 on the matrix cores (fused first pair): which tap (ci * 9 + ky * 3 + kx; -1: none, zero weight) K slot
// (k-step kk, half-wave kh, element j) of the 27 -> 32 padded reduction multiplies.  Chosen so that a lane's sixteen patch
// reads are base(kk, kh) + a compile-time offset: kk = 0 is input channel kh at kernel positions 0..7 (the half-waves' taps
// lie one channel plane apart), kk = 1 holds channel 2 -- half-wave kh reads column kx = kh of the three kernel rows in
// j = 0..2 (the taps lie one pixel apart), column 2 in j = 3..5 -- and position 8 of channels 0 / 1 in j = 6 / 7; the
// slots half-wave 1 has no tap for read the patch one pixel further (a finite value) against a zero weight.
__host__ __device__ constexpr int first_conv_slot_tap(int kk, int kh, int j) {
  if (kk == 0) return kh * 9 + j;
  if (j < 3) return 18 + j * 3 + kh;
  if (kh != 0) return -1;
  if (j < 6) return 18 + (j - 3) * 3 + 2;
  return (j - 6) * 9 + 8;
}

// bf16 mode (BF = true kernels; conv mode "bf16"): ONE product per fp32 product on v_mfma_f32_32x32x16_bf16, operands
// rounded to bf16 (8 mantissa bits, fp32's exponent range: no fp16 range guard, no activation exponent).  The 16-bit
// "hi" halves of the LDS rows / weight packs then hold bf16 bit patterns and the "lo" halves are never read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
template <bool BF>
__device__ __forceinline__ f32x16 mma16(const half8 a, const half8 b, const f32x16 c) {
  if constexpr (BF)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ _Float16 bf16_as_half(float x) { return __builtin_bit_cast(_Float16, (__bf16)x); }
// two floats -> their bf16 bit patterns in one register (low half = a).  (hipcc 7.2 lowers a VECTOR float2 -> bf16x2
// conversion to v_cvt_pk_bf16_f32 with the first element in both source slots -- the odd element is lost: tools/diag_bf16.py
// -- so the instruction is spelled out.)
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// four floats -> four bf16 bit patterns in a half4
__device__ __forceinline__ void bf16x4_of(const float4 v, _Float16 (&h)[4]) {
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  const h2_ a = __builtin_bit_cast(h2_, pk_bf16(v.x, v.y)), b = __builtin_bit_cast(h2_, pk_bf16(v.z, v.w));
  h[0] = a[0]; h[1] = a[1]; h[2] = b[0]; h[3] = b[1];
}

// x -> (hi, lo) for four values, two per instruction: v_cvt_pk_f16_f32 for both halves, packed fp32
// subtract / scale in between (3 VALU ops per value instead of 6; same results as the scalar form)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 v, half4& hi, half4& lo) {
  const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
  const half2v h01 = __builtin_convertvector(x01, half2v), h23 = __builtin_convertvector(x23, half2v);
  const f32x2 r01 = (x01 - __builtin_convertvector(h01, f32x2)) * f16x3::LO_SCALE;
  const f32x2 r23 = (x23 - __builtin_convertvector(h23, f32x2)) * f16x3::LO_SCALE;
  const half2v l01 = __builtin_convertvector(r01, half2v), l23 = __builtin_convertvector(r23, half2v);
  hi = half4{h01[0], h01[1], h23[0], h23[1]};
  lo = half4{l01[0], l01[1], l23[0], l23[1]};
}
template <bool BF>
__device__ __forceinline__ void split4t(const float4 v, half4& hi, half4& lo) {
  if constexpr (BF) {
    _Float16 h[4];
    bf16x4_of(v, h);
    hi = half4{h[0], h[1], h[2], h[3]};
    lo = half4{(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
  } else {
    split4(v, hi, lo);
  }
}

// BN = 128: waves 4(M) x 2(N), each 64 px x 64 couts (MT = 2 M-tiles); BN = 64: waves 8 x 1, each 32 px x 64 couts.
// FUSE1: the input of this layer is the first conv of the net (3x3, pad 1, Cin <= 3, + ReLU) applied
// to the raw image: instead of reading its 64-channel output from HBM, the halo tile is COMPUTED
// in place from a 20x20x3 image patch staged in LDS (one thread per halo pixel, 27 x 32 FMAs per
// chunk, under the MFMAs of the previous chunk).  conv1_1 never touches HBM: -1.8 GB written and
// read per image on the bench pyramid.
// NP: fp16 products per fp32 product -- 3 (fp32-class), 2 (a_lo * b_hi dropped: activations act as fp16) or 1 (hi * hi).
template <int BN, bool FUSE1, int DIL, int KS, int NP = 3, bool BF = false>
__global__ __launch_bounds__(512) void conv_mfma_f16x3_kernel(ConvK p) {
  static_assert(!BF || NP == 1, "bf16 mode is a one-product mode");
  using namespace f16x3;
  // halo tile for dilation DIL (the dilated heads: 2 and 4, BN = 64 only -- a 24x24 tile plus 128-cout weight
  // buffers would not fit the 160 KiB of LDS)
  // KS = 3: a stage is one kernel row (3 taps) of a 32-channel chunk; KS = 1 (1x1 convolutions): a stage is
  // the single tap of a chunk, no halo, and every stage hands the next chunk's tile over
  constexpr int PADH = KS == 3 ? DIL : 0, KROWS = KS == 3 ? 3 : 1;
  constexpr int HTW = TW + 2 * PADH, HTH = TH + 2 * PADH, HP = HTH * HTW;
  static_assert(KS == 3 || (KS == 1 && DIL == 1 && !FUSE1), "kernel sizes 3 (any dilation) and 1");
  static_assert(!FUSE1 || DIL == 1, "the fused first layer is a dilation-1 path");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef SHF_CONV_TIMING
  const unsigned long long t_entry8 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int WN = BN / 64;
  constexpr int WM = 8 / WN;
  constexpr int MT = TH / (2 * WM);      // 2x16-pixel MFMA row tiles per wave: 2 (BN=128) or 1 (BN=64)
  constexpr int ALD = (HP * 8 + 511) / 512;  // float4 halo pieces per thread: 6
  unsigned char* As = smem;                  // [HP][ROWB]
  unsigned char* Bs = smem + HP * ROWB;      // [2][3][BN][ROWB]
  // FUSE1 extras behind the weight buffers
  constexpr int PW = TW + 4, PH = TH + 4;    // image patch: halo of the halo
  float* patch = (float*)(Bs + 2 * KS * BN * ROWB);  // [3][PH][PW]
  float* w1s = patch + 3 * PH * PW;                 // [27][64]
  float* b1s = w1s + 27 * 64;                       // [64]
  // first-layer weights [27][64]: read through the CONSTANT address space so that the wave-uniform accesses
  // become s_load_dwordx8/16 (scalar cache -> SGPRs), not per-lane memory instructions
  const __attribute__((address_space(4))) float* w1g = (const __attribute__((address_space(4))) float*)(unsigned long long)p.w1t;

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
  int a_off[MT], b_off[2];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = ((wm * 2 * MT + t * 2 + dy) * HTW + px) * ROWB + kh * 16;
#pragma unroll
  for (int t = 0; t < 2; ++t) b_off[t] = (wn * 64 + t * 32 + i) * ROWB + kh * 16;

  const int nchunks = p.Cin / KC;
  const int NST = nchunks * KROWS;  // stages
  const _Float16* wsp = (const _Float16*)p.wp;
  // weights: [chunk][ky][kx][cout][hi 32 | lo 32 | 8 pad] halfs = the LDS row image (144 B)
  const size_t slab = (size_t)p.Cout * 72;        // halfs per (chunk,ky,kx)
  const _Float16* wbase = wsp + (size_t)ct * BN * 72;

  // per-thread halo piece geometry (chunk-invariant)
  int a_goff[ALD], a_loff[ALD];
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    const int idx = tid + 512 * j;
    const int hp = idx >> 3, q = idx & 7;
    const int hy = hp / HTW, hx = hp - hy * HTW;
    const int gy = ty0 - PADH + hy, gx = tx0 - PADH + hx;
    const bool in = (idx < HP * 8) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
    a_goff[j] = in ? ((b * H + gy) * W + gx) * p.in_stride + q * 4 : -1;
    a_loff[j] = (idx < HP * 8) ? hp * ROWB + q * 8 : -1;
  }
  float4 areg0[ALD];  // prologue only: the halo registers of the main loop are local to the hand-over stage
  // Weight slabs go global -> LDS by DMA (global_load_lds_dwordx4: no registers, no ds_write): the
  // packed global layout already has the padded 144-B rows, so a stage (3 slabs of the block's BN
  // couts) is 3 contiguous runs copied in 1-KiB pieces, one piece per wave-instruction.
  constexpr int SLAB_B = BN * ROWB;            // bytes per slab in LDS and in global
  constexpr int PCS_SLAB = SLAB_B / 1024;      // 18 (BN=128) or 9 (BN=64)
  constexpr int PCS = KS * PCS_SLAB;
  static_assert(SLAB_B % 1024 == 0, "slab must be whole DMA pieces");
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define F16X3_DMA_W(STAGE, BUF, NWAVES)                                                              \
  {                                                                                                  \
    const unsigned char* ws_ = (const unsigned char*)(wbase + (size_t)(STAGE) * KS * slab);           \
    unsigned char* bd_ = Bs + (BUF) * (KS * SLAB_B);                                                  \
    _Pragma("unroll") for (int j = 0; j < (PCS + (NWAVES) - 1) / (NWAVES); ++j) {                    \
      const int pc = wave_u + (NWAVES) * j;                                                          \
      if (pc < PCS) {                                                                                \
        const int sl = pc / PCS_SLAB, within = pc - sl * PCS_SLAB;                                   \
        const unsigned char* src = ws_ + (size_t)sl * slab * 2 + within * 1024 + lane * 16;          \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,         \
                                         (__attribute__((address_space(3))) void*)(bd_ + pc * 1024), 16, 0, 0); \
      }                                                                                              \
    }                                                                                                \
  }

  // prologue: halo(0) and W(0) into LDS
  // FUSE1 per-thread state: thread `tid` owns halo pixel hp = tid (tid < HP)
  half4 fhi[8], flo[8];
  const int f_hy = tid / HTW, f_hx = tid - (tid / HTW) * HTW;
  const bool f_own = FUSE1 && tid < HP;
  const bool f_inside = f_own && ((unsigned)(ty0 - 1 + f_hy) < (unsigned)H) && ((unsigned)(tx0 - 1 + f_hx) < (unsigned)W);
  auto first_conv = [&](int chunk) {
    // conv1_1 + ReLU for channels chunk*32 .. +31 at this thread's halo pixel; zero outside the image
    // (that is conv1_2's zero padding, not conv1_1 evaluated out there)
    // two channels per instruction (v_pk_fma_f32): the same fused multiply-adds at half the VALU issue
    f32x2 acc2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc2[j] = f32x2{b1s[chunk * 32 + 2 * j], b1s[chunk * 32 + 2 * j + 1]};
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kyy = 0; kyy < 3; ++kyy)
#pragma unroll
        for (int kxx = 0; kxx < 3; ++kxx) {
          const float v = patch[(ci * PH + f_hy + kyy) * PW + f_hx + kxx];
          const f32x2 vv = {v, v};
          // wave-uniform address into the kernel argument's array: scalar loads (s_load_dwordx8/16), weights stay
          // in SGPRs.  (From LDS every tap was a dependent ds_read_b128 round trip: 216 x ~64 cycles per pass.)
          const __attribute__((address_space(4))) f32x4* wv =
              (const __attribute__((address_space(4))) f32x4*)(w1g + ((ci * 3 + kyy) * 3 + kxx) * 64 + chunk * 32);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x4 w4 = wv[q];
            acc2[2 * q] = __builtin_elementwise_fma(vv, f32x2{w4[0], w4[1]}, acc2[2 * q]);
            acc2[2 * q + 1] = __builtin_elementwise_fma(vv, f32x2{w4[2], w4[3]}, acc2[2 * q + 1]);
          }
        }
    float acc[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[2 * j] = acc2[j][0]; acc[2 * j + 1] = acc2[j][1]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float4 v4 = make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f),
                              fmaxf(acc[4 * q + 3], 0.f));
      if (!f_inside) v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      split4t<BF>(v4, fhi[q], flo[q]);
    }
  };
  auto first_store = [&]() {
    if (f_own) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        *(half4*)(As + tid * ROWB + q * 8) = fhi[q];
        *(half4*)(As + tid * ROWB + q * 8 + 64) = flo[q];
      }
    }
  };
  // W(0) is requested before anything else; the accumulator clearing fills part of the wait
  F16X3_DMA_W(0, 0, 8);
  if constexpr (FUSE1) {
    const float* img = mem.img + (size_t)b * 3 * H * W;
    for (int idx = tid; idx < 3 * PH * PW; idx += 512) {
      const int ci = idx / (PH * PW), r = idx - ci * (PH * PW);
      const int py = r / PW, pxx = r - py * PW;
      const int gy = ty0 - 2 + py, gx = tx0 - 2 + pxx;
      patch[idx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? img[((size_t)ci * H + gy) * W + gx] : 0.f;
    }
    for (int idx = tid; idx < 27 * 64; idx += 512) w1s[idx] = p.w1t[idx];
    if (tid < 64) b1s[tid] = p.b1 ? p.b1[tid] : 0.f;
    __syncthreads();
    if (f_own) first_conv(0);
  } else {
    const float* inc_ = gin;
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      areg0[j] = (a_goff[j] >= 0) ? *(const float4*)(inc_ + a_goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x16 accm[MT][2], accc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][c][r] = 0.f; accc[a][c][r] = 0.f; }
  if constexpr (FUSE1) {
    first_store();
  } else {
#pragma unroll
    for (int j = 0; j < ALD; ++j)
      if (a_loff[j] >= 0) {
        half4 hi, lo;
        split4t<BF>(areg0[j], hi, lo);
        *(half4*)(As + a_loff[j]) = hi;
        *(half4*)(As + a_loff[j] + 64) = lo;
      }
  }

#ifdef SHF_CONV_TIMING
  const unsigned long long t_loop8 = __builtin_amdgcn_s_memtime();
  unsigned long long tb = 0, ti = 0, tc = 0, tx = 0, t0, t1, t2, t3;
#define SHF_T(x) x = __builtin_amdgcn_s_memtime()
#else
#define SHF_T(x)
#endif
  for (int c = 0; c < nchunks; ++c) {
   float4 areg[ALD];
   half4 ahi[ALD], alo[ALD];
#pragma unroll
   for (int ky = 0; ky < KROWS; ++ky) {
    const int st = c * KROWS + ky;
    SHF_T(t0);
    // LDS-DMA is only ordered by the issuing wave's own vmcnt: drain it by hand before the barrier
    // (hipcc drops this wait when the DMA sits behind the loop back-edge / in a wave-uniform branch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SHF_T(t1);
#if !F16X3_DMA_LATE
    if (st + 1 < NST) F16X3_DMA_W(st + 1, (st + 1) & 1, 8);
#endif
    const bool last_row = (ky == KROWS - 1);
    const bool more_chunks = (c + 1 < nchunks);
    if constexpr (!FUSE1) {
      if (last_row && more_chunks) {
        const float* inc_ = gin + (c + 1) * KC;
#pragma unroll
        for (int j = 0; j < ALD; ++j)
          areg[j] = (a_goff[j] >= 0) ? *(const float4*)(inc_ + a_goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    SHF_T(t2);
    const unsigned char* Arow = As + (ky * DIL * HTW) * ROWB;
    const unsigned char* Bst = Bs + (st & 1) * (KS * BN * ROWB);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const unsigned char* Ap = Arow + kx * DIL * ROWB;
      const unsigned char* Bp = Bst + kx * (BN * ROWB);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8 ah[MT], al[MT], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          ah[t] = *(const half8*)(Ap + a_off[t] + kk * 32);
          al[t] = *(const half8*)(Ap + a_off[t] + kk * 32 + 64);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bh[t] = *(const half8*)(Bp + b_off[t] + kk * 32);
          bl[t] = *(const half8*)(Bp + b_off[t] + kk * 32 + 64);
        }
        if constexpr (MT == 1) {
        // three passes over the tiles so that consecutive MFMAs never chain on one accumulator
  #pragma unroll
          for (int tm = 0; tm < MT; ++tm)
  #pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              accm[tm][tn] = mma16<BF>(ah[tm], bh[tn], accm[tm][tn]);
  #pragma unroll
          for (int tm = 0; tm < MT; ++tm)
  #pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              if constexpr (NP >= 2) accc[tm][tn] = mma16<BF>(ah[tm], bl[tn], accc[tm][tn]);
  #pragma unroll
          for (int tm = 0; tm < MT; ++tm)
  #pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              if constexpr (NP >= 3) accc[tm][tn] = mma16<BF>(al[tm], bh[tn], accc[tm][tn]);
        } else {
          // with 4 output tiles per wave hipcc's own interleave of the tile-major order measured faster
#pragma unroll
          for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              accm[tm][tn] = mma16<BF>(ah[tm], bh[tn], accm[tm][tn]);
              if constexpr (NP >= 2) accc[tm][tn] = mma16<BF>(ah[tm], bl[tn], accc[tm][tn]);
              if constexpr (NP >= 3) accc[tm][tn] = mma16<BF>(al[tm], bh[tn], accc[tm][tn]);
            }
        }
      }
      if (F16X3_CONV_MID && kx == KS / 2 && last_row && more_chunks) {
        // prepare the next chunk's halo while the matrix pipe drains
        if constexpr (FUSE1) {
          if (f_own) first_conv(c + 1);
        } else {
#pragma unroll
          for (int j = 0; j < ALD; ++j) split4t<BF>(areg[j], ahi[j], alo[j]);
        }
      }
    }
    // The waves of the first half finish their MFMAs early (they win the matrix-pipe arbitration
    // against their SIMD partners of the second half), so they feed the DMA engine for the next
    // stage from that slack instead of every wave paying the issue cost before its MFMAs.
#if F16X3_DMA_LATE
    if (wave_u < 4 && st + 1 < NST) F16X3_DMA_W(st + 1, (st + 1) & 1, 4);
#endif
#ifdef SHF_CONV_TIMING
    asm volatile("s_nop 0" ::: "memory");
    SHF_T(t3);
    tb += t1 - t0; ti += t2 - t1; tc += t3 - t2;
#endif
    if (last_row && more_chunks) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave is done reading the halo tile of chunk c
      if constexpr (FUSE1) {
        first_store();
      } else {
#pragma unroll
        for (int j = 0; j < ALD; ++j)
          if (a_loff[j] >= 0) {
            if (!F16X3_CONV_MID) split4t<BF>(areg[j], ahi[j], alo[j]);
            *(half4*)(As + a_loff[j]) = ahi[j];
            *(half4*)(As + a_loff[j] + 64) = alo[j];
          }
      }
    }
#ifdef SHF_CONV_TIMING
    SHF_T(t0);
    tx += t0 - t3;
#endif
   }
  }
#ifdef SHF_CONV_TIMING
  const unsigned long long t_end8 = __builtin_amdgcn_s_memtime();
  if (p.dbg && lane == 0 && (bid == 0 || bid == 100)) {
    unsigned long long* d = p.dbg + ((bid ? 1 : 0) * 8 + wave) * 5;
    d[0] = tb; d[1] = ti; d[2] = tc; d[3] = tx; d[4] = NST;
    if (wave == 0) printf("[f16x3 8w] blk%d prologue %llu loop %llu (%d stages)\n", bid, t_loop8 - t_entry8, t_end8 - t_loop8, NST);
  }
#endif
#undef F16X3_DMA_W

  // epilogue: C row = (r&3) + 8*(r>>2) + 4*(lane>>5), C col = lane&31
  float amax = 0.f;  // fp16 range guard: largest |output| of this lane
  if (p.relu & 16) {
    __syncthreads();  // the K loop's LDS buffers are dead: the output tile is transposed through them
    float* Cs = (float*)smem;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int cl = wn * 64 + tn * 32 + i;
      const float bv = p.bias ? p.bias[ct * BN + cl] : 0.f;
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        if (p.relu & 1)
          conv_stage_tile_pk<BN, true>(Cs, accm[tm][tn], accc[tm][tn], LO_INV, bv, wm * 2 * MT + tm * 2, kh, cl, amax, H - ty0, W - tx0);
        else
          conv_stage_tile_pk<BN, false>(Cs, accm[tm][tn], accc[tm][tn], LO_INV, bv, wm * 2 * MT + tm * 2, kh, cl, amax, H - ty0, W - tx0);
      }
    }
    __syncthreads();
    conv_flush_tile<BN, 512>(Cs, tid, ty0, tx0, H, W, b, ct * BN, gout, p.out_stride, mem.pool, p.pool_stride,
                             !(p.relu & 8), (p.relu & 32) != 0, (p.relu & 64) != 0);
    conv_raise_range_flag(p.range_flag, amax);
    conv_publish_amax(mem.out_amax, mem.pool ? mem.pool_amax : nullptr, amax);
#ifdef SHF_CONV_TIMING
    if (p.dbg && tid == 0 && (bid == 0 || bid == 100))
      printf("[f16x3 8w] blk%d epilogue %llu\n", bid, (unsigned long long)__builtin_amdgcn_s_memtime() - t_end8);
#endif
    return;
  }
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int cout = ct * BN + wn * 64 + tn * 32 + i;
    const float bv = p.bias ? p.bias[cout] : 0.f;
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
      const f32x16 am = accm[tm][tn], ac = accc[tm][tn];
      conv_store_tile([&](int r) { return am[r] + ac[r] * LO_INV; }, bv, p.relu, ty0 + wm * 2 * MT + tm * 2, tx0, kh,
                      H, W, b, cout, gout, p.out_stride, mem.pool, p.pool_stride, &amax);
    }
  }
  conv_raise_range_flag(p.range_flag, amax);
  conv_publish_amax(mem.out_amax, mem.pool ? mem.pool_amax : nullptr, amax);
}

// DUAL-TILE form of the 4-wave kernel: a block computes TWO 16x16-pixel tiles (consecutive in the launch's tile order)
// x 128 couts and every weight slab it fetches serves both -- the weights' way from L2 to LDS is what this power-limited
// kernel pays most for after the MFMAs themselves (halving it is worth 15 %).  What makes room for the second
// tile's accumulators is ONE accumulator per output instead of two: the low parts are kept UNSCALED in LDS
// (lo = fp16(x - hi); the split activation format of HBM keeps its 2^11 -- the halo staging multiplies it away; the
// weights come from their own pack, pre-scaled by a power of two: pack_conv_weights_split16h), so hi*hi, hi*lo and
// lo*hi have one scale and share a register (v_mfma_f32_32x32x16_f16 honours fp16 subnormals: tools/mfma_denorm.hip;
// end-to-end error of the scheme: tools/single_acc_study.py).  What makes room for the second halo tile's hand-over
// registers is a CHUNK of 16 input channels instead of 32: a stage is still one kernel row of a chunk = 144 MFMAs per
// wave (3 taps x 1 k-step x 2 tiles x 24), its three weight slabs are 30 KB instead of 55, a halo tile 30 KB instead
// of 48, and a hand-over moves 2 x 6 pieces per thread.  The six half-steps of a stage (tap kx, tile t) are
// software-pipelined like the six k-steps of the single-tile kernel: while (kx, t) runs, the A fragments of the next
// (kx, t) -- and, on even half-steps, the B fragments of tap kx + 1 -- are read.
//
// HALO TILES ARE DOUBLE-BUFFERED (round 3): with one buffer per tile the hand-over was a serial section -- barrier,
// convert + park 12 pieces, barrier, first fragment reads -- that cost the dominant launch 7.6 % with the matrix pipe
// idle (tools/experiments/w4d_power_ablation.sh: no_halo).  Chunk c + 1's pieces are now requested during kernel row
// 1 of chunk c and converted + parked into the OTHER buffer pair under the MFMAs of kernel row 2, two pieces per
// half-step; the next stage's barrier -- which the weights need anyway -- publishes them.  Four halo tiles fit in
// 160 KB because the layout is PLANAR, without per-row padding: plane q (hi k 0-7 | hi k 8-15 | lo k 0-7 | lo k 8-15)
// holds one 16-byte piece per halo pixel, rows of 24 pixels (384 B = 8 sixteen-byte slots mod 16, so the two pixel
// rows a ds_read_b128 lane group touches land on complementary halves of the 256-B bank row), planes 32 B apart
// mod 128 (the 8-lane groups of the parking ds_write_b128 -- 2 pixels x 4 planes -- cover all 32 banks).  Every
// fragment address is lane offset + immediate: tap kx = +16 B, kernel row = +384 B, lo = +2 planes.
template <bool IN_SPLIT, int MT_, int NTILE, int NP = 3, bool BF = false>
__global__ __launch_bounds__(256) void conv_mfma_f16x3_w4d_kernel(ConvK p) {
  static_assert(!BF || (NP == 1 && !IN_SPLIT), "bf16 mode: one product, fp32 activations in HBM");
  static_assert((MT_ == 4 || MT_ == 2) && (NTILE == 1 || NTILE == 2), "16- or 8-row tiles, one or two per block");
  constexpr int MT = MT_, TH = 4 * MT, TW = 16, HTW = 18, HTH = TH + 2, HP = HTH * HTW;
  constexpr int KC = 16, BN = 128, NT = 256;
  constexpr int PROW = 24 * 16;                       // 384 B per halo-tile row of a plane (18 pixels used)
  constexpr int PLANE = HTH * PROW + 32;              // 6 944 B (16-row tiles) / 3 872 B (8-row tiles)
  constexpr int AS_B = 4 * PLANE;                     // 27 776 B / 15 488 B per halo tile
  constexpr int NB_B = NTILE * AS_B;                  // one buffer set (the tiles of one chunk)
  constexpr int WROWB = 64;                           // weight rows: no padding, the 16-byte pieces rotated by row / 4
  constexpr int SLAB_B = BN * WROWB;                  // 8 192 B per tap slab
  constexpr int ALD = (HP * 4 + NT - 1) / NT;         // 16-byte halo pieces per thread and tile: 6 (16 rows) or 3
  constexpr float LO_SCALE = 2048.0f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                           // [2 buffer sets][NTILE][4 planes][HTH][24 px][16 B]
  unsigned char* Bs = smem + 2 * NB_B;                // [2 buffers][3 taps][BN][64 B]
  float* biasL = (float*)(Bs + 2 * 3 * SLAB_B);       // [BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  int bid = blockIdx.x;
  if (p.xcd_remap) {   // (experiment, SHF_F16X3_XCD_REMAP=1) blocks go to XCDs round-robin: give XCD x one contiguous run
    const int G = gridDim.x, q = G >> 3, r = G & 7, x = bid & 7;
    bid = x * q + (x < r ? x : r) + (bid >> 3);
  }
  const int ct = bid % p.nct;
  const int pp = bid / p.nct;
  const int ntiles = p.ntile_blocks / p.nct;          // the launch covers pixel tiles [tile_base, ntiles) of the group

  const int nchunks = p.Cin / KC;
  const int NST = nchunks * 3;
  const size_t slab = (size_t)p.Cout * 32;            // halfs per tap slab of the whole layer
  const _Float16* wbase = (const _Float16*)p.wph + (size_t)ct * BN * 32;

  // weight DMA: round r (0..5) of a wave moves 1-KiB piece q = wave + 4 r of the stage's 24 (8 per tap slab: rounds
  // 0-1 / 2-3 / 4-5 are slabs 0 / 1 / 2 for every wave).  LDS offset = q KiB; global offset = slab (q / 8) + (q % 8) KiB.
  constexpr int W_ROUNDS = 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane16 = (unsigned)lane * 16u;
  const size_t slab_b = slab * 2;
  auto w_goff = [&](int r) -> size_t { return (size_t)(r >> 1) * slab_b + (size_t)(4 * (r & 1) + wave_u) * 1024; };
  auto w_loff = [&](int r) { return (4 * r + wave_u) * 1024; };
  auto dma_w = [&](int stage, int buf, int r0, int n) {
    const unsigned char* ws_ = (const unsigned char*)(wbase + (size_t)stage * 3 * slab);
    unsigned char* bd_ = Bs + buf * (3 * SLAB_B);
#pragma unroll
    for (int r = r0; r < r0 + n; ++r) {
      // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell the DMA's LDS destination (weights)
      // from the halo buffers, and would drain vmcnt -- i.e. wait out the weight fetch it has just issued -- before
      // every ds_write that parks a halo piece inside the stage.  Completion is waited for by hand at the stage start.)
      const unsigned char* ub = ws_ + w_goff(r);
      const unsigned lds = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(bd_ + w_loff(r));
      // (M0 is compiler-reserved and not preserved around a statement: it is written in the statement that reads it,
      // with the one wait state an SALU write of M0 needs before the LDS-DMA that uses it -- nothing inside an asm string
      // is padded by the compiler.  The "s" operands are SALU results; a value fresh from v_readfirstlane would need five
      // wait states before the load.)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane16), "s"(ub));
    }
  };

  // the first stage's weights only depend on the cout tile: requested BEFORE the tile decode (dozens of dependent scalar
  // loads through the member table), so that their round trip runs under it
  dma_w(0, 0, 0, W_ROUNDS);

  struct Geo { int ty0, tx0, H, W, b; const float* in; float* out; float* pool; const unsigned* in_amax; unsigned* out_amax; unsigned* pool_amax; };
  auto geometry = [&](int t) {
    Geo g;
    int pt = t;
    const int mi = conv_find_member(p, pt);
    const ConvMember mem = p.m[mi];   // (a COPY: the whole record in a few wide scalar loads, not a dependent load per field)
    pt -= mem.tile_start;
    int ty_, tx_;
    conv_split_tile(mem, pt, g.b, ty_, tx_);
    g.ty0 = ty_ * TH;
    g.tx0 = tx_ * TW;
    g.H = mem.H; g.W = mem.W;
    g.in = mem.in; g.out = mem.out; g.pool = mem.pool;
    g.in_amax = mem.in_amax; g.out_amax = mem.out_amax; g.pool_amax = mem.pool_amax;
    return g;
  };
  const int t0 = p.tile_base + NTILE * pp;
  const bool has1 = NTILE == 2 && t0 + 1 < ntiles;    // (an odd tile count: the last block's second tile is a dummy)
  const Geo g0 = geometry(t0), g1 = geometry(has1 ? t0 + 1 : t0);


  // halo piece j of this thread (per tile): 16-byte piece q = idx & 3 of halo pixel idx >> 2, idx = tid + 256 j.
  //   split input : q = 0, 1: hi channels 0-7 / 8-15 of the 16-channel half chunk; q = 2, 3: lo (scaled by 2^11 in HBM)
  //   fp32 input  : q = channels 4q .. 4q+3 (float4)
  // a_goff = BYTE offset of the piece inside the member's input for chunk 0 (the chunk adds a uniform offset)
  unsigned a_goff[NTILE][ALD];
  unsigned a_valid = 0;                               // bit t * 8 + j
  // (the pixel stride in a vector register: out of scalar registers here, the compiler re-read the kernel argument for
  // each of the twelve pieces, a scalar load and a wait apiece)
  int in_stride_v = p.in_stride;
  asm volatile("" : "+v"(in_stride_v));
  auto halo_offsets = [&](const Geo& g, int t, bool exists) {
#pragma unroll
    for (int j = 0; j < ALD; ++j) {
      const int idx = tid + NT * j;
      const int hp = idx >> 2, q = idx & 3;
      const int hy = hp / HTW, hx = hp - hy * HTW;
      const int gy = g.ty0 - 1 + hy, gx = g.tx0 - 1 + hx;
      const bool in = exists && (idx < HP * 4) && ((unsigned)gy < (unsigned)g.H) && ((unsigned)gx < (unsigned)g.W);
      const unsigned pix = (unsigned)(((g.b * g.H + gy) * g.W + gx) * in_stride_v) * 4u;
      a_goff[t][j] = in ? pix + (IN_SPLIT ? (unsigned)((q >> 1) * 64 + (q & 1) * 16) : (unsigned)(q * 16)) : 0u;
      a_valid |= in ? (1u << (t * 8 + j)) : 0u;
    }
  };
  halo_offsets(g0, 0, true);
  if constexpr (NTILE == 2) halo_offsets(g1, 1, has1);
  // chunk c16 -> byte offset inside a pixel
  auto chunk_off = [&](int c16) -> unsigned {
    return IN_SPLIT ? (unsigned)((c16 >> 1) * 128 + (c16 & 1) * 32) : (unsigned)(c16 * 64);
  };
  // ACTIVATION EXPONENT (conv_common.h): the tile's unit publishes max |input| (its producers' epilogues); the halo
  // staging multiplies hi by 2^e and the format's lo (which carries 2^11) by 2^(e - 11), exactly, in fp16 -- the top of
  // the unit's input lands in [2^13, 2^14), so the UNSCALED low parts the single accumulator needs are normal fp16
  // numbers whatever the layer's magnitude (without it a layer living around 1e-3 kept 14 bits, not 22) -- and the
  // epilogue multiplies 2^-e back together with the weights' scale.  e <= 15, so 2^e and 2^(e - 11) are fp16 numbers and
  // the lift is one exact multiplication per value.
  // e is a function of the unit alone, so every grouping of tiles into launches / blocks forms the same bits.
  const int e_t0 = __builtin_amdgcn_readfirstlane(conv_act_exponent(g0.in_amax));
  const int e_t1 = NTILE == 2 ? __builtin_amdgcn_readfirstlane(conv_act_exponent(g1.in_amax)) : 0;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  // pk_f1 = this thread's factor for a split-format piece (its pieces are all hi or all lo: q = tid & 3), hi1 / lo1 = the
  // two factors of an fp32 piece -- plain registers, no struct (hipcc parks a struct that is indexed by a lane-dependent
  // select in scratch memory)
  struct ActScale { unsigned pk_f1, hi1, lo1; };
  const int lo_shift = (tid & 2) ? 11 : 0;
  auto act_scale = [&](int e) {
    return ActScale{conv_pk_pow2_f16(e - lo_shift), conv_pk_pow2_f16(e), conv_pk_pow2_f16(e - 11)};
  };
  const ActScale as0 = act_scale(e_t0), as1 = act_scale(e_t1);
  // piece as fetched -> the 16 bytes (split input) / the hi half4 | lo half4 pair (fp32 input) that go to LDS.  A piece
  // outside the image (bit `vbit` of a_valid clear; it fetched the member's first bytes) becomes zeros by way of its
  // FACTOR -- no select on the data and, above all, no branch: the parking runs inside the MFMA stages, and control flow
  // there would split the region the sched_group_barriers order
  auto convert = [&](float4& v, int vbit, const ActScale& sc_) {
    const unsigned keep = (unsigned)((int)(a_valid << (31 - vbit)) >> 31);   // all ones / zero
    struct { h2 hi1, lo1; } sc = {__builtin_bit_cast(h2, sc_.hi1 & keep), __builtin_bit_cast(h2, sc_.lo1 & keep)};
    if constexpr (IN_SPLIT) {
      const h2 f1 = __builtin_bit_cast(h2, sc_.pk_f1 & keep);
      float* e = &v.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h2 x = __builtin_bit_cast(h2, e[k]);
        x = x * f1;
        e[k] = __builtin_bit_cast(float, x);
      }
    } else if constexpr (BF) {
      _Float16 h[4];
      bf16x4_of(v, h);   // (no activation exponent: bf16 has fp32's range; the weights' pack is unscaled)
      const half2v h01 = {h[0], h[1]}, h23 = {h[2], h[3]};
      v = make_float4(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, h01) & keep),
                      __builtin_bit_cast(float, __builtin_bit_cast(unsigned, h23) & keep), 0.f, 0.f);
    } else {
      const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
      // (lo through the split activation format's 2^11, like a producer's epilogue + the staging above would: the two
      // input formats then give the same bits even where hi or lo is an fp16 subnormal)
      const h2 h01 = __builtin_convertvector(x01, h2), h23 = __builtin_convertvector(x23, h2);
      const h2 l01 = __builtin_convertvector((x01 - __builtin_convertvector(h01, f32x2)) * LO_SCALE, h2);
      const h2 l23 = __builtin_convertvector((x23 - __builtin_convertvector(h23, f32x2)) * LO_SCALE, h2);
      v = make_float4(__builtin_bit_cast(float, h01 * sc.hi1), __builtin_bit_cast(float, h23 * sc.hi1),
                      __builtin_bit_cast(float, l01 * sc.lo1), __builtin_bit_cast(float, l23 * sc.lo1));
    }
  };
  // (set_off: byte offset of the buffer set the pieces go to)
  auto store_piece = [&](const float4& v, int t, int j, unsigned set_off) {
    const int idx = tid + NT * j;
    const int hp = idx >> 2, q = idx & 3;
    int hy = hp / HTW, hx = hp - hy * HTW;
    if (NT * (j + 1) > HP * 4) {
      // the ragged last piece: threads past the tile's end store theirs in the unused columns 18..23 of the first rows
      // (no branch inside the stage -- it would split the scheduling region)
      const int hpd = hp - HP, ry = hpd / 6;
      const bool past = idx >= HP * 4;
      hy = past ? ry : hy;
      hx = past ? HTW + hpd - ry * 6 : hx;
    }
    unsigned char* pix = As + set_off + t * AS_B + hy * PROW + hx * 16;
    if constexpr (IN_SPLIT) {
      *(float4*)(pix + q * PLANE) = v;
    } else {
      *(float2*)(pix + (q >> 1) * PLANE + (q & 1) * 8) = make_float2(v.x, v.y);
      *(float2*)(pix + (2 + (q >> 1)) * PLANE + (q & 1) * 8) = make_float2(v.z, v.w);
    }
  };

  // prologue
  float4 areg0[ALD], areg1[ALD];  // (two named arrays, indexed by unrolled inner loops only: anything indexed by the
                                  // half-step variable stays in scratch memory)
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    areg0[j] = *(const float4*)((const char*)g0.in + a_goff[0][j]);
    if constexpr (NTILE == 2) areg1[j] = *(const float4*)((const char*)g1.in + a_goff[1][j]);
  }
  if (tid < BN) biasL[tid] = p.bias ? p.bias[ct * BN + tid] : 0.f;

  const int i = lane & 31, kh = lane >> 5;
  int dy, px;
  row_to_pixel(i, dy, px);
  int a_off[MT], b_off[2];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = kh * PLANE + (wm * 2 * MT + t * 2 + dy) * PROW + px * 16;
  int a_delta = NB_B;              // the chunk's buffer set lives in a_off: += a_delta after every chunk but the last
  unsigned park_off = NB_B;        // the buffer set the NEXT chunk is parked in
#pragma unroll
  for (int t = 0; t < 2; ++t)   // the lane's hi piece (the lo piece sits two rotated positions further: b_off ^ ... below)
    b_off[t] = (wn * 64 + t * 32 + i) * WROWB + ((kh + ((wn * 64 + t * 32 + i) >> 2)) & 3) * 16;
  f32x16 acc0[MT][2], acc1[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[a][c][r] = 0.f; acc1[a][c][r] = 0.f; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    convert(areg0[j], j, as0);
    store_piece(areg0[j], 0, j, 0u);
  }
  if constexpr (NTILE == 2) {
#pragma unroll
    for (int j = 0; j < ALD; ++j) {
      convert(areg1[j], 8 + j, as1);
      store_piece(areg1[j], 1, j, 0u);
    }
  }

  unsigned seen0 = 0xffffffffu, seen0p = 0xffffffffu, seen1 = 0xffffffffu, seen1p = 0xffffffffu;
  // one stage = kernel row KY of the 16-channel chunk c.  MODE 1 (kernel row 1 of a chunk that has a successor):
  // request the pieces of chunk c + 1's halo tiles; MODE 2 (kernel row 2): convert them and park them in the other
  // buffer set, spread over the half-steps
  auto stage = [&](int c, auto KY_, auto MODE_) {
    constexpr int ky = decltype(KY_)::value;
    constexpr int MODE = decltype(MODE_)::value;
    const int st = c * 3 + ky;
    // this wave's share of W(st) (and, MODE 2, its halo pieces) has landed; its parked pieces are in LDS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int st_next = st + 1 < NST ? st + 1 : st;   // the last stage re-fetches itself (unused) instead of branching
    const int buf_next = (st + 1) & 1;
    const unsigned coff = chunk_off(c + 1);
    const unsigned char* Bst = Bs + (st & 1) * (3 * SLAB_B);
    half8 fa[2][2 * MT], fb[2][4];
    auto load_a = [&](int h, half8* a) {              // half-step h = NTILE kx + tile
      const unsigned char* Ap = As + (h % NTILE) * AS_B + ky * PROW + (h / NTILE) * 16;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        a[2 * t] = *(const half8*)(Ap + a_off[t]);
        a[2 * t + 1] = *(const half8*)(Ap + a_off[t] + 2 * PLANE);
      }
    };
    auto load_b = [&](int kx, half8* bf) {
      const unsigned char* Bp = Bst + kx * SLAB_B;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf[2 * t] = *(const half8*)(Bp + b_off[t]);
        bf[2 * t + 1] = *(const half8*)(Bp + (b_off[t] ^ 32));   // (piece + 2) mod 4 within the 64-byte row
      }
    };
    load_a(0, fa[0]);
    load_b(0, fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NH = 3 * NTILE;                     // half-steps per stage
    constexpr int NPC = NTILE * ALD;                  // pieces per thread and chunk
    constexpr int PP = (NPC + NH - 1) / NH;           // pieces parked per half-step (MODE 2): 2 or 1
    constexpr int PARK_VALU = IN_SPLIT ? 4 : 7;       // vector instructions the scheduler may put beside one MFMA
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      constexpr int DMA_N2[6] = {1, 1, 1, 1, 1, 1}, DMA_J2[6] = {0, 1, 2, 3, 4, 5};
      constexpr int DMA_N1[3] = {2, 2, 2}, DMA_J1[3] = {0, 2, 4};
      const int dma_n = NTILE == 2 ? DMA_N2[h] : DMA_N1[h], dma_j = NTILE == 2 ? DMA_J2[h] : DMA_J1[h];
      const int kx = h / NTILE, tl = h % NTILE;
      half8* a = fa[h & 1];
      half8* bf = fb[kx & 1];
      int n_ds = 0;
      if (h + 1 < NH) { load_a(h + 1, fa[(h + 1) & 1]); n_ds += 2 * MT; }
      if (tl == 0 && kx + 1 < 3) { load_b(kx + 1, fb[(kx + 1) & 1]); n_ds += 4; }
      if (dma_n) dma_w(st_next, buf_next, dma_j, dma_n);
      int n_vmem = dma_n;
      if constexpr (MODE == 1) {
        // tile 0's / tile 1's pieces of the next chunk are requested in the first half-steps
        if (h == 0) {
#pragma unroll
          for (int j = 0; j < ALD; ++j) areg0[j] = *(const float4*)((const char*)g0.in + (a_goff[0][j] + coff));
          n_vmem += ALD;
        } else if (NTILE == 2 && h == 1) {
#pragma unroll
          for (int j = 0; j < ALD; ++j) areg1[j] = *(const float4*)((const char*)g1.in + (a_goff[NTILE - 1][j] + coff));
          n_vmem += ALD;
        }
      }
      if constexpr (MODE == 3) {
        // the very last stage: read the units' max |output| slots now (conv_amax_peek), under the MFMAs
        if (h == 0) {
          seen0 = conv_amax_peek(g0.out_amax);
          seen0p = conv_amax_peek(g0.pool ? g0.pool_amax : nullptr);
          if constexpr (NTILE == 2) {
            seen1 = conv_amax_peek(g1.out_amax);
            seen1p = conv_amax_peek(g1.pool ? g1.pool_amax : nullptr);
          }
          n_vmem += 2 * NTILE;
        }
      }
      int n_park = 0;
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = h * PP; k < (h + 1) * PP && k < NPC; ++k) {
          const int t = k / ALD, j = k % ALD;
          if (t == 0) {
            convert(areg0[j], j, as0);
            store_piece(areg0[j], 0, j, park_off);
          } else {
            convert(areg1[j], 8 + j, as1);
            store_piece(areg1[j], 1, j, park_off);
          }
          ++n_park;
        }
      }
      auto mfmas = [&](f32x16 (&acc)[MT][2]) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = mma16<BF>(bf[2 * tn], a[2 * tm], acc[tm][tn]);
        if constexpr (NP >= 2) {
#pragma unroll
          for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = mma16<BF>(bf[2 * tn + 1], a[2 * tm], acc[tm][tn]);
        }
        if constexpr (NP >= 3) {
#pragma unroll
          for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = mma16<BF>(bf[2 * tn], a[2 * tm + 1], acc[tm][tn]);
        }
      };
      if (tl) mfmas(acc1);
      else mfmas(acc0);
      constexpr int NM = 2 * NP * MT;                 // MFMAs of the half-step
      if (h + 1 < NH || n_park > 0) {
        // next half-step's fragment reads go out under the first MFMAs, the VMEM issues and the parking (vector
        // instructions of a piece, then its LDS store) over the rest
        const int n_first = n_ds < NM ? n_ds : NM;
#pragma unroll
        for (int g = 0; g < n_first; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        const int n_rest = NM > n_first + 2 ? NM - n_first - 2 : 0;
        const int per_piece = n_park > 0 ? (n_rest / n_park > 0 ? n_rest / n_park : 1) : 0;   // MFMA slots per parked piece
#pragma unroll
        for (int g = 0; g < n_rest; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < n_vmem) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          if (n_park > 0) {
            __builtin_amdgcn_sched_group_barrier(0x002, PARK_VALU, 0);
            if (g % per_piece == per_piece - 1 && g / per_piece < n_park)
              __builtin_amdgcn_sched_group_barrier(0x200, IN_SPLIT ? 1 : 2, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using std::integral_constant;
#pragma unroll 1
  for (int c = 0; c + 1 < nchunks; ++c) {
    stage(c, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    stage(c, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    stage(c, integral_constant<int, 2>{}, integral_constant<int, 2>{});
#pragma unroll
    for (int t = 0; t < MT; ++t) a_off[t] += a_delta;
    a_delta = -a_delta;
    park_off = NB_B - park_off;
  }
  stage(nchunks - 1, integral_constant<int, 0>{}, integral_constant<int, 0>{});
  stage(nchunks - 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
  stage(nchunks - 1, integral_constant<int, 2>{}, integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last stage's (unused) self re-fetch, the slot peeks
  asm volatile("" : "+v"(seen0), "+v"(seen0p), "+v"(seen1), "+v"(seen1p));   // (the compiler's own wait for them goes HERE)
  // (wave-uniform: parked in scalar registers until the end of the epilogue -- a vector register would be spilled)
  seen0 = __builtin_amdgcn_readfirstlane(seen0);
  seen0p = __builtin_amdgcn_readfirstlane(seen0p);
  seen1 = __builtin_amdgcn_readfirstlane(seen1);
  seen1p = __builtin_amdgcn_readfirstlane(seen1p);

  // register epilogue, one tile after the other (each with its unit's scale and its unit's max |output| slot)
  float amax0 = 0.f, amax1 = 0.f;
  {
    const bool relu = (p.relu & 1) != 0, write_main = !(p.relu & 8), main_split = (p.relu & 32) != 0,
               pool_split = (p.relu & 64) != 0;
    // the lane's coordinates are formed AGAIN here, from the lane id the hardware hands out (mbcnt) and the wave number
    // in its scalar register: kept alive across the K loop they were spilled, and every scratch reload in an epilogue
    // is followed by a vmcnt(0) that waits for all the stores issued so far
    int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_e));
    const int i_e = lane_e & 31, kh_e = lane_e >> 5;
    int px_e, dy_e;
    row_to_pixel(i_e, dy_e, px_e);
    const int wn = wave_u & 1, wm = wave_u >> 1;
    // (kernel arguments the 16 accumulator tiles all use: in vector registers, or the compiler -- out of scalar
    // registers here -- re-reads each of them from the argument segment for every tile, an s_load + wait apiece)
    int out_stride_e = p.out_stride, pool_stride_e = p.pool_stride;
    float wscale_inv_e = p.wscale_inv;
    asm volatile("" : "+v"(out_stride_e), "+v"(pool_stride_e), "+v"(wscale_inv_e));
    float4 bias16[2][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
      bias16[g >> 2][g & 3] = *(const float4*)(biasL + wn * 64 + (g >> 2) * 32 + kh_e * 16 + 4 * (g & 3));
    auto tile_out = [&](f32x16 (&acc)[MT][2], const Geo& g, bool exists, int e_act, float& amax) {
      const float out_scale = wscale_inv_e * __builtin_bit_cast(float, (unsigned)(127 - e_act) << 23);   // 2^-e, exact
      const int Hp = (g.H + 1) >> 1, Wp = (g.W + 1) >> 1;
      const bool interior = exists && g.ty0 + TH <= g.H && g.tx0 + TW <= g.W;
      const int x = g.tx0 + px_e;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int cout16 = ct * BN + wn * 64 + tn * 32 + kh_e * 16;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
          int y = g.ty0 + wm * 2 * MT + tm * 2 + dy_e;
          // (opaque: this tile's address arithmetic starts HERE -- hoisted to the top for all 16 tiles it was spilled, and
          // a scratch reload between the stores waits for every store issued so far)
          asm volatile("" : "+v"(y));
          const bool valid = exists && y < g.H && x < g.W;
          const unsigned pix_m = (unsigned)((g.b * g.H + y) * g.W + x), pix_q = (unsigned)((g.b * Hp + (y >> 1)) * Wp + (x >> 1));
          float* pm = write_main ? g.out + (size_t)pix_m * (unsigned)out_stride_e : nullptr;
          float* pq = g.pool ? g.pool + (size_t)pix_q * (unsigned)pool_stride_e : nullptr;
          if (relu)
            conv_epilogue_regs1<true>(acc[tm][tn], out_scale, bias16[tn], valid, interior, pm, cout16, main_split, pq,
                                      valid && (i_e & 3) == 0, pool_split, amax);
          else
            conv_epilogue_regs1<false>(acc[tm][tn], out_scale, bias16[tn], valid, interior, pm, cout16, main_split, pq,
                                       valid && (i_e & 3) == 0, pool_split, amax);
        }
      }
    };
    // POOL-ONLY layers (conv2_2, conv3_3 of VGG-16: the un-pooled map has no other reader): conv_epilogue_pool_only -- the
    // quad max on the raw accumulator order, then each lane of a quad finishes a quarter of the couts; same bits
    auto tile_out_pool = [&](f32x16 (&acc)[MT][2], const Geo& g, bool exists, int e_act, float& amax) {
      const float out_scale = wscale_inv_e * __builtin_bit_cast(float, (unsigned)(127 - e_act) << 23);   // 2^-e, exact
      const int Hp = (g.H + 1) >> 1, Wp = (g.W + 1) >> 1;
      const bool interior = exists && g.ty0 + TH <= g.H && g.tx0 + TW <= g.W;
      const int x = g.tx0 + px_e;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
          int y = g.ty0 + wm * 2 * MT + tm * 2 + dy_e;
          asm volatile("" : "+v"(y));
          const bool valid = exists && y < g.H && x < g.W;
          const bool window = exists && y - dy_e < g.H && x - (i_e & 1) < g.W;
          const unsigned pix_q = (unsigned)((g.b * Hp + (y >> 1)) * Wp + (x >> 1));
          float* pq = g.pool + (size_t)pix_q * (unsigned)pool_stride_e;
          const f32x16 a_ = acc[tm][tn];
          conv_epilogue_pool_only<true>([&](int r) { return a_[r] * out_scale; },
                                        [&](int q) { return *(const float4*)(biasL + wn * 64 + tn * 32 + 8 * q + 4 * kh_e); }, valid, window,
                                        interior, pq, ct * BN + wn * 64 + tn * 32, kh_e, i_e & 3, pool_split, amax);
        }
      }
    };
    if (relu && !write_main && g0.pool) {   // (wave-uniform; a launch's members share the layer)
      tile_out_pool(acc0, g0, true, e_t0, amax0);
      if constexpr (NTILE == 2) tile_out_pool(acc1, g1, has1, e_t1, amax1);
    } else {
      tile_out(acc0, g0, true, e_t0, amax0);
      if constexpr (NTILE == 2) tile_out(acc1, g1, has1, e_t1, amax1);
    }
  }
  conv_raise_range_flag(p.range_flag, conv_absmax_bits(amax0, amax1));
  conv_amax_commit(g0.out_amax, seen0, g0.pool ? g0.pool_amax : nullptr, seen0p, amax0);
  if constexpr (NTILE == 2) {
    if (has1) conv_amax_commit(g1.out_amax, seen1, g1.pool ? g1.pool_amax : nullptr, seen1p, amax1);   // (wave-uniform)
  }
}

// Producer / consumer variant of the fused first pair (conv1_1 -> conv1_2, Cin = Cout = 64).  With only
// 64 couts a wave of the 8-wave kernel owns ONE 32-pixel MFMA row tile (MT = 1) and needs a ds_read_b128 per
// MFMA -- LDS-bound at ~40 % matrix-pipe use -- and its conv1_1 (lane = halo pixel, 27 taps x 32 channels
// with one weight fetch per packed FMA) is latency-bound: ~12 k cycles per 32-channel pass, two passes.
// Here:
//  * conv1_1 runs ONCE, in the prologue, on all eight waves with lane = output channel: the lane keeps
//    its 27 weights in registers, the pixel values are wave-uniform LDS broadcasts of the image patch, and
//    one v_pk_fma_f32 advances two neighbouring pixels.  Both 32-channel chunks of the 18x18 halo tile are
//    written to LDS (two tiles: the first-layer weights no longer live there, so both fit) -- no second
//    pass, no hand-over barrier in the K loop.
//  * waves 0-3 are CONSUMERS (one per SIMD): 64 px x 64 couts = 4 accumulator tiles each, 8 fragment reads
//    per 12 MFMAs, the six k-steps of a stage software-pipelined like the 4-wave kernel (~2.6 k cycles per
//    stage against 2.3 k of pure MFMA issue).  Waves 4-7 are PRODUCERS: they issue every weight DMA.
//  * PERSIST (round 3): one block per CU WALKS the tiles (tile = block, block + grid, ...).  The producers fetch the
//    next tile's image patch, its validity flags and its first weight stage under the current K loop, so a tile no
//    longer pays the block turnaround, the tile decode and the patch's round trip (≈6 k of ≈37 k cycles).
template <int NP, bool BF = false, bool PERSIST = false>
__global__ __launch_bounds__(512) void conv_mfma_f16x3_pc_kernel(ConvK p) {
  using namespace f16x3;
  static_assert(!BF || NP == 1, "bf16 mode is a one-product mode");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef SHF_CONV_TIMING
  unsigned long long tt[14];
  int nt = 0;
#define PC_T() tt[nt++] = __builtin_amdgcn_s_memtime()
#else
#define PC_T()
#endif
  PC_T();
  constexpr int BN = 64, MT = 2;
  constexpr int PW = TW + 4, PH = TH + 4;
  constexpr int HPP = (HP + 31) / 32 * 32;          // 352: tile rows padded to whole 32-row MFMA tiles, so that
                                                    // conv1_1's epilogue stores need no per-row guard
  unsigned char* As0 = smem;                        // [HP][ROWB] channels  0..31 of conv1_1's output
  unsigned char* As1 = smem + HP * ROWB;            // [HP][ROWB] channels 32..63 (the last row tile's stores are guarded)
  unsigned char* Bs = smem + 2 * HP * ROWB;         // [2][3][BN][ROWB]
  // [3][PH][PW] image patch, already split: fp16 hi in the low half of a dword, fp16 lo (x 2^11) in the high half (bf16 mode:
  // the bf16 pattern | 0) -- conv1_1's fragments are then gathered with one byte permute per register, no conversion
  // (round 4; the conversions used to be redone for every fragment element: ~200 vector instructions per row tile).
  // (+ 8 dwords: half-wave 1's zero-weight slots read one element past a tap)
  unsigned* patch = (unsigned*)(Bs + 2 * 3 * BN * ROWB);
  constexpr int PATCH_DW = 3 * PH * PW + 8;
  unsigned char* valid = (unsigned char*)(patch + PATCH_DW);  // [HPP] halo pixel inside the image? (0 in the padding)
  float* bias2L = (float*)(valid + HPP);                          // [BN] conv1_2's biases (read by the register epilogue)
  // conv1_1's operands live in LDS (round 4): its weight fragments [n][kk][hi/lo][lane][8 halfs] (8 KiB, the global pack
  // as it is) and biases -- read where a row tile needs them (ds_read latency, no registers held across anything), by
  // whichever wave has claimed the row tile
  unsigned char* w1L = (unsigned char*)(bias2L + BN);             // 8192 B
  float* b1L = (float*)(w1L + 8192);                              // [64]
  unsigned* ctrL = (unsigned*)(b1L + 64);                         // [0] next row tile of the next tile's conv1_1 to claim, [1] its halo_inside
  unsigned* geoL = ctrL + 4;                                      // [16] the next tile's geometry (TileGeo), decoded ONCE, by a producer
  constexpr int PC_TABN = 300;
  unsigned* tabL = geoL + 16;                                     // [PC_TABN] packed geometry of the tiles this block walks (ConvK::pc_tab)

  int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool consumer = wave_u < 4;
  // what does not depend on the tile is requested FIRST -- conv1_1's weight fragments and biases (64 registers), conv1_2's
  // biases for LDS -- so that their round trip runs under the tile decode and the image patch's (they used to be
  // requested after the patch was parked: 2 k cycles of a second, serial round trip per tile)
  int i1 = lane & 31, kh1 = lane >> 5;
  const float* b2p = p.bias ? p.bias : (const float*)p.w1f;
  {   // (requested first: the round trip runs under the tile decode and the image patch's)
    const float4 wv = ((const float4*)p.w1f)[tid];                 // 512 threads x 16 B = the 8 KiB pack
    const float b1v = p.b1 ? p.b1[tid & 63] : 0.f;
    ((float4*)w1L)[tid] = wv;
    if (tid < 64) b1L[tid] = b1v;
    if (tid == 0) ctrL[0] = 0u;
  }
  if (PERSIST && p.pc_tab) {
    // this block's tiles (blockIdx + k gridDim), decoded by all lanes in parallel, once: in the walk the decode is then one LDS
    // word + the member's record instead of ~2-4 k cycles of dependent scalar loads on a producer wave beside the consumers'
    // MFMAs (scalar-register spills are vector instructions, and a matrix stream leaves its SIMD partner ~3 of those per MFMA)
    for (int k = tid; (int)blockIdx.x + k * (int)gridDim.x < p.ntile_blocks; k += 512) {
      int pt = (int)blockIdx.x + k * (int)gridDim.x, mi = 0;
      int ts = 0;
      unsigned tpi = (unsigned)p.m[0].tiles_per_img, itpi = p.m[0].inv_tiles_per_img, tlx = (unsigned)p.m[0].tiles_x, itlx = p.m[0].inv_tiles_x;
#pragma unroll
      for (int q = 1; q < MAX_GROUP; ++q) {
        const bool ge = pt >= p.tile_starts[q];   // (unused entries are INT_MAX)
        mi = ge ? q : mi;
        ts = ge ? p.tile_starts[q] : ts;
        tpi = ge ? (unsigned)p.m[q].tiles_per_img : tpi;
        itpi = ge ? p.m[q].inv_tiles_per_img : itpi;
        tlx = ge ? (unsigned)p.m[q].tiles_x : tlx;
        itlx = ge ? p.m[q].inv_tiles_x : itlx;
      }
      pt -= ts;
      const unsigned b_ = conv_div((unsigned)pt, tpi, itpi);
      pt -= (int)(b_ * tpi);
      const unsigned ty_ = conv_div((unsigned)pt, tlx, itlx), tx_ = (unsigned)pt - ty_ * tlx;
      tabL[k] = (unsigned)mi | (b_ << 4) | (ty_ << 12) | (tx_ << 22);
    }
  }
  const float bias2v = b2p[tid & (BN - 1)];
  const int bid = blockIdx.x;
  // the tile's geometry (wave-uniform; PERSIST: re-formed for every tile of the walk).  nct == 1: tile = pixel tile
  struct TileGeo { int b, ty0, tx0, H, W; const float* img; float* out; float* pool; unsigned* out_amax; unsigned* pool_amax; };
  auto decode = [&](int tile) {
    int pt = tile;
    const ConvMember& m = p.m[conv_find_member(p, pt)];
    pt -= m.tile_start;
    int b_, ty_, tx_;
    conv_split_tile(m, pt, b_, ty_, tx_);
    return TileGeo{b_, ty_ * TH, tx_ * TW, m.H, m.W, m.img + (size_t)b_ * 3 * m.H * m.W, m.out, m.pool, m.out_amax, m.pool_amax};
  };
  auto decode_tab = [&](int k) {   // the k-th tile of this block's walk, from the LDS table
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)tabL[k]);
    const ConvMember& m = p.m[w & 15u];
    const int b_ = (int)((w >> 4) & 255u), ty_ = (int)((w >> 12) & 1023u), tx_ = (int)(w >> 22);
    return TileGeo{b_, ty_ * TH, tx_ * TW, m.H, m.W, m.img + (size_t)b_ * 3 * m.H * m.W, m.out, m.pool, m.out_amax, m.pool_amax};
  };
  int tile = bid, k_walk = 0;
  const int ntiles = PERSIST ? p.ntile_blocks : 0, gstride = (int)gridDim.x;
  TileGeo mem = decode(tile);
  int b = mem.b, ty0 = mem.ty0, tx0 = mem.tx0, H = mem.H, W = mem.W;
  float* gout = mem.out;
#ifdef SHF_CONV_TIMING
  asm volatile("" :: "s"(H), "s"(W), "s"(ty0), "s"(tx0));
  const unsigned long long t_dec = __builtin_amdgcn_s_memtime();
#endif

  constexpr int SLAB_B = BN * ROWB;          // 9 KiB
  constexpr int PCS_SLAB = SLAB_B / 1024;    // 9
  constexpr int PCS = 3 * PCS_SLAB;          // 27 one-KiB pieces per stage
  const size_t slab = (size_t)p.Cout * 72;
  const _Float16* wbase = (const _Float16*)p.wp;
  auto dma_w = [&](int stage, int buf) {     // producer waves only: 7 rounds of 4 pieces (the last one ragged)
    // (opaque base: the 42 source addresses of a tile are formed where they are used, on the scalar unit -- as loop
    // invariants of the persistent walk they would occupy 84 scalar registers, i.e. be spilled)
    const _Float16* wb = wbase;
    asm volatile("" : "+s"(wb));
    const unsigned char* ws_ = (const unsigned char*)(wb + (size_t)stage * 3 * slab);
    unsigned char* bd_ = Bs + buf * (3 * SLAB_B);
#pragma unroll
    for (int j = 0; j < (PCS + 3) / 4; ++j) {
      int pc = (wave_u - 4) + 4 * j;
      pc = pc < PCS ? pc : PCS - 1;
      const int sl = pc / PCS_SLAB, within = pc - sl * PCS_SLAB;
      const unsigned char* src = ws_ + (size_t)sl * slab * 2 + within * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(bd_ + pc * 1024), 16, 0, 0);
    }
  };
  if (!consumer) dma_w(0, 0);
#ifdef SHF_CONV_TIMING
  const unsigned long long t_dma = __builtin_amdgcn_s_memtime();
#endif

  float amax1 = 0.f;  // fp16 range guard for conv1_1's outputs (split right here, never seen by another epilogue)
  half2v amax1h = {(_Float16)0, (_Float16)0};   // ... its packed form, raised by conv1_tile on the hi halves
  auto patch_word = [](float x) -> unsigned {   // fp16 hi | fp16 lo (x 2^11) << 16; bf16 mode: the bf16 pattern
    if constexpr (BF) {
      return (unsigned)__builtin_bit_cast(unsigned short, bf16_as_half(x));
    } else {
      const _Float16 h = (_Float16)x;
      const _Float16 l = (_Float16)((x - (float)h) * LO_SCALE);
      return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
    }
  };
  // prologue, all eight waves (first tile of a walk; later tiles: the producers, under the previous tile's epilogue -- below):
  // conv1_1 + ReLU of the whole halo tile ON THE MATRIX CORES
    // [324 halo px x 27 taps (padded to 32)] x [32 x 64 couts] as split-fp16 MFMAs: 11 row tiles of 32 pixels,
    // 12 MFMAs each; a lane builds its A fragments (pixel lane&31, 8 taps) from the LDS image patch, the B
    // fragments (weights) come pre-packed from global memory.  N tile 0 / 1 = channel chunk 0 / 1 = halo tile
    // As0 / As1.  (On the vector ALUs this was 15-18 k cycles per tile, a third of the block.)
    static_assert(PH == 20 && PW == 20 && HTW == 18, "the multiply-shift divisions below are exact for these sizes");
    {   // (a later tile's patch, flags and first weights are fetched by the producers under the previous tile's K loop)
    const float* img = mem.img;
    constexpr int NPATCH = (3 * PH * PW + 511) / 512;   // 3 values per thread (the last round ragged): all requested, then parked
    float pv[NPATCH];
#pragma unroll
    for (int k = 0; k < NPATCH; ++k) {
      // (integer division is a ~40-instruction sequence: n / 400, n / 20 and n / 18 as multiply + shift, exact below 1300 / 420 / 400)
      const int idx = tid + 512 * k;
      const int ci = (idx * 2622) >> 20, r = idx - ci * (PH * PW);
      const int py = (r * 52429) >> 20, pxx = r - py * PW;
      const int gy = ty0 - 2 + py, gx = tx0 - 2 + pxx;
      const bool in = idx < 3 * PH * PW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      pv[k] = in ? img[((size_t)ci * H + gy) * W + gx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NPATCH; ++k) {
      const int idx = tid + 512 * k;
      if (idx < 3 * PH * PW) patch[idx] = patch_word(pv[k]);
      amax1 = conv_absmax_bits(amax1, pv[k]);  // the image itself is split to fp16 hi/lo for conv1_1's MFMAs
    }
    if (tid < BN) bias2L[tid] = p.bias ? bias2v : 0.f;
    if (tid < 8) patch[3 * PH * PW + tid] = 0u;
    if (tid < HPP) {
      const int qy = (tid * 58255) >> 20, qx = tid - qy * HTW;
      valid[tid] = (tid < HP && (unsigned)(ty0 - 1 + qy) < (unsigned)H && (unsigned)(tx0 - 1 + qx) < (unsigned)W) ? 1 : 0;
    }
    }
    PC_T();
    // conv1_1 runs as D[cout][pixel] (weights = A operand): a lane owns ONE halo pixel and the 16 couts
    // (r & 3) + 8 (r >> 2) + 4 kh of each 32-channel chunk -- one validity flag per lane, and after the half-wave
    // exchange 16 consecutive couts = two 16-byte LDS stores each for hi and lo (the D[pixel][cout] form wrote 32 two-byte
    // values per lane and chunk and read 16 flags).
    __syncthreads();
    PC_T();
    constexpr int NMT = (HP + 31) / 32;  // 11 row tiles
    // work items: tiles 0..7 whole (one per wave), tiles 8..10 split by N tile over waves 0..5: the longest
    // wave does 1.5 tiles instead of 2
    auto conv1_tile = [&](int m, int n_lo, int n_hi, bool halo_inside) {
      const int hp = m * 32 + i1 < HP ? m * 32 + i1 : HP - 1;
      const int hy = (hp * 58255) >> 20, hx = hp - hy * HTW;
      // sixteen packed patch words at base(kk, kh) + a compile-time offset (first_conv_slot_tap), then one byte permute per
      // fragment register: the low halves of two words are two hi values, the high halves the two lo values
      const unsigned* pb = patch + hy * PW + hx;
      const unsigned* b0 = pb + kh1 * (PH * PW);
      const unsigned* b1 = pb + kh1;
      unsigned e0[8], e1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e0[j] = b0[(j / 3) * PW + j % 3];
        constexpr int T1[8] = {18, 21, 24, 20, 23, 26, 8, 17};   // first_conv_slot_tap(1, 0, j)
        static_assert(first_conv_slot_tap(1, 0, 3) == 20 && first_conv_slot_tap(1, 0, 7) == 17 && first_conv_slot_tap(1, 1, 2) == 25, "slot map");
        e1[j] = b1[((T1[j] / 9) * PH + (T1[j] % 9) / 3) * PW + T1[j] % 3];
      }
      half8 ah[2], al[2];
      {
        unsigned h0[4], l0[4], h1[4], l1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h0[r] = __builtin_amdgcn_perm(e0[2 * r + 1], e0[2 * r], 0x05040100u);
          l0[r] = __builtin_amdgcn_perm(e0[2 * r + 1], e0[2 * r], 0x07060302u);
          h1[r] = __builtin_amdgcn_perm(e1[2 * r + 1], e1[2 * r], 0x05040100u);
          l1[r] = __builtin_amdgcn_perm(e1[2 * r + 1], e1[2 * r], 0x07060302u);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        ah[0] = __builtin_bit_cast(half8, u32x4{h0[0], h0[1], h0[2], h0[3]});
        al[0] = __builtin_bit_cast(half8, u32x4{l0[0], l0[1], l0[2], l0[3]});
        ah[1] = __builtin_bit_cast(half8, u32x4{h1[0], h1[1], h1[2], h1[3]});
        al[1] = __builtin_bit_cast(half8, u32x4{l1[0], l1[1], l1[2], l1[3]});
      }
      const bool row_ok = m + 1 < NMT || m * 32 + i1 < HP;   // (the last row tile is ragged: 324 = 10 x 32 + 4)
      // THREE PHASES, each over both channel chunks: every LDS operand read (weight fragments, biases) issued up front, then
      // all the MFMAs (two independent chains), then the two epilogues.  Written chunk by chunk -- operands, MFMAs, epilogue,
      // stores, next chunk -- the compiler waited for each bias quad on its own (eight serial LDS round trips) and could not
      // start chunk 1's reads before chunk 0's LDS stores: a row tile was one ~2.8 k-cycle dependent chain.
      half8 bwn[2][2][2];   // [n][kk][hi / lo]
      float4 bq[2][4];      // [n][register quad]: biases of couts 8q + 4kh .. + 3
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n < n_lo || n >= n_hi) continue;  // wave-uniform
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl)
            bwn[n][kk][hl] = *(const half8*)(w1L + ((size_t)((n * 2 + kk) * 2 + hl) * 64 + (i1 + 32 * kh1)) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[n][q] = *(const float4*)(b1L + n * 32 + 8 * q + 4 * kh1);
      }
      unsigned char okb = 1;
      const bool need_ok = !(halo_inside && m + 1 < NMT);   // (wave-uniform: most row tiles have every halo pixel inside the image)
      if (need_ok) okb = valid[m * 32 + i1];               // 0 outside the image (conv1_2's zero padding, not conv1_1 evaluated out there)
      f32x16 cm[2], cc[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n < n_lo || n >= n_hi) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) { cm[n][r] = 0.f; cc[n][r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          cm[n] = mma16<BF>(bwn[n][kk][0], ah[kk], cm[n]);
          if constexpr (!BF) cc[n] = mma16<BF>(bwn[n][kk][1], ah[kk], cc[n]);
        }
        if constexpr (!BF) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) cc[n] = mma16<BF>(bwn[n][kk][0], al[kk], cc[n]);
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n < n_lo || n >= n_hi) continue;
        // C row (cout) = (r & 3) + 8 (r >> 2) + 4 kh, C column (halo pixel) = lane & 31: registers 4q .. 4q + 3 are the FOUR
        // CONSECUTIVE couts 8q + 4kh .. + 3 -- 8 bytes of hi and 8 bytes of lo in the pixel's LDS row, stored as they are (the
        // half-wave exchange that made 16-byte stores of them cost eight permlane swaps with their wait states)
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * q] = fmaxf(cm[n][4 * q] + cc[n][4 * q] * LO_INV + bq[n][q].x, 0.f);
          v[4 * q + 1] = fmaxf(cm[n][4 * q + 1] + cc[n][4 * q + 1] * LO_INV + bq[n][q].y, 0.f);
          v[4 * q + 2] = fmaxf(cm[n][4 * q + 2] + cc[n][4 * q + 2] * LO_INV + bq[n][q].z, 0.f);
          v[4 * q + 3] = fmaxf(cm[n][4 * q + 3] + cc[n][4 * q + 3] * LO_INV + bq[n][q].w, 0.f);
        }
        if (need_ok) {
          const bool ok = okb != 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = ok ? v[r] : 0.f;
        }
        unsigned char* row = (n ? As1 : As0) + (m * 32 + i1) * ROWB + kh1 * 8;
        float2 sh[4], sl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half2v h0, h1, l0, l1;
          const f32x2 x0 = {v[4 * q], v[4 * q + 1]}, x1 = {v[4 * q + 2], v[4 * q + 3]};
          if constexpr (BF) {
            h0 = __builtin_bit_cast(half2v, pk_bf16(x0[0], x0[1]));
            h1 = __builtin_bit_cast(half2v, pk_bf16(x1[0], x1[1]));
            l0 = l1 = half2v{(_Float16)0, (_Float16)0};
          } else {
            h0 = __builtin_convertvector(x0, half2v);
            h1 = __builtin_convertvector(x1, half2v);
            l0 = __builtin_convertvector((x0 - __builtin_convertvector(h0, f32x2)) * LO_SCALE, half2v);
            l1 = __builtin_convertvector((x1 - __builtin_convertvector(h1, f32x2)) * LO_SCALE, half2v);
            // fp16 range guard of conv1_1's outputs, on the PACKED hi halves (values >= 0; an overflow is an inf there)
            amax1h = __builtin_elementwise_max(amax1h, __builtin_elementwise_max(h0, h1));
          }
          sh[q] = make_float2(__builtin_bit_cast(float, h0), __builtin_bit_cast(float, h1));
          sl[q] = make_float2(__builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1));
        }
        if (row_ok) {   // (one branch for the row's eight stores)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *(float2*)(row + q * 16) = sh[q];
            if constexpr (!BF) *(float2*)(row + 64 + q * 16) = sl[q];
          }
        }
      }
    };
    {
      const bool halo_inside = ty0 >= 1 && tx0 >= 1 && ty0 + TH < H && tx0 + TW < W;   // (wave-uniform)
      conv1_tile(wave_u, 0, 2, halo_inside);
      if (wave_u < 2 * (NMT - 8)) conv1_tile(8 + wave_u % (NMT - 8), wave_u / (NMT - 8), wave_u / (NMT - 8) + 1, halo_inside);
      amax1 = conv_absmax_bits(amax1, fmaxf((float)amax1h[0], (float)amax1h[1]));
    }

  TileGeo nxt_pre = mem;
  if (PERSIST && !consumer && tile + gstride < ntiles) nxt_pre = decode(tile + gstride);
#ifdef SHF_CONV_TIMING
  unsigned long long ts_k = 0, ts_bar = 0, ts_role = 0, ts_tail = 0, t_role_end = 0, ts_st[6] = {0, 0, 0, 0, 0, 0}, ts_own[6] = {0, 0, 0, 0, 0, 0};
  int n_walk = 0;
#endif
  for (;;) {   // (PERSIST: the walk over this block's tiles; otherwise one turn)
#ifdef SHF_CONV_TIMING
  nt = 3;
#endif
  // consumer geometry: wave wm = rows 4 wm .. 4 wm + 3 (two 2x16-pixel MFMA row tiles), all 64 couts
  const int i = lane & 31, kh = lane >> 5;
  int dy, px;
  row_to_pixel(i, dy, px);
  const int wm = wave_u & 3;
  int a_off[MT], b_off[2];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = ((wm * 2 * MT + t * 2 + dy) * HTW + px) * ROWB + kh * 16;
#pragma unroll
  for (int t = 0; t < 2; ++t) b_off[t] = (t * 32 + i) * ROWB + kh * 16;
  f32x16 accm[MT][2], accc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][c][r] = 0.f; accc[a][c][r] = 0.f; }

  auto mma_stage = [&](const unsigned char* Atile, int ky, int buf) {
    const unsigned char* Arow = Atile + (ky * HTW) * ROWB;
    const unsigned char* Bst = Bs + buf * (3 * BN * ROWB);
    half8 fa[2][2 * MT], fb[2][4];
    auto load_frag = [&](int s_, half8* a, half8* bf) {
      const unsigned char* Ap = Arow + (s_ >> 1) * ROWB + (s_ & 1) * 32;
      const unsigned char* Bp = Bst + (s_ >> 1) * (BN * ROWB) + (s_ & 1) * 32;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        a[2 * t] = *(const half8*)(Ap + a_off[t]);
        a[2 * t + 1] = *(const half8*)(Ap + a_off[t] + 64);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf[2 * t] = *(const half8*)(Bp + b_off[t]);
        bf[2 * t + 1] = *(const half8*)(Bp + b_off[t] + 64);
      }
    };
    load_frag(0, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      half8* a = fa[s_ & 1];
      half8* bf = fb[s_ & 1];
      if (s_ + 1 < 6) load_frag(s_ + 1, fa[(s_ + 1) & 1], fb[(s_ + 1) & 1]);
#pragma unroll
      for (int tm = 0; tm < MT; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          accm[tm][tn] = mma16<BF>(bf[2 * tn], a[2 * tm], accm[tm][tn]);   // weights = A operand: D[cout][pixel]
      if constexpr (NP >= 2) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            accc[tm][tn] = mma16<BF>(bf[2 * tn + 1], a[2 * tm], accc[tm][tn]);
      }
      if constexpr (NP >= 3) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            accc[tm][tn] = mma16<BF>(bf[2 * tn], a[2 * tm + 1], accc[tm][tn]);
      }
      if (s_ + 1 < 6) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read of the next step
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  unsigned seen = 0xffffffffu, seenp = 0xffffffffu;
  // PERSIST, producers: the next tile of the walk -- its patch is requested in stage 1 and parked (with the validity flags)
  // in stage 2: the patch and the flags are only read by conv1_1 -- this tile's ended before stage 0, the next tile's starts
  // behind the post-K barrier
  constexpr int NPF2 = (3 * PH * (PW / 2) + 255) / 256;   // 3 x-pairs per producer thread
  static_assert(PW % 2 == 0, "x-pairs");
  const bool has_next = PERSIST && tile + gstride < ntiles;
  TileGeo nxt = nxt_pre;   // (decoded a tile ago by the producers, under stage 4: the decode is ~2 k cycles of dependent scalar loads,
                           // and in stage 0 -- in front of the patch requests -- it held up the stage's barrier: 5.1 k cycles, not 2.7)
  float pvn[2 * NPF2];
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    // producers: their share of W(st) has landed (PERSIST, stage 2: and the next tile's image patch, requested in stage 1
    // in front of W(2)'s pieces)
#ifdef SHF_CONV_TIMING
    if (st > 0) { asm volatile("s_nop 0" ::: "memory"); ts_own[st - 1] += __builtin_amdgcn_s_memtime() - tt[2 + st]; }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PC_T();
    if (consumer) {
      if (st == 5) {   // the unit's max |output| slots, read under the last stage (conv_amax_peek)
        seen = conv_amax_peek(mem.out_amax);
        seenp = conv_amax_peek(mem.pool ? mem.pool_amax : nullptr);
      }
      mma_stage(st < 3 ? As0 : As1, st % 3, st & 1);
    } else {
      if (!(PERSIST && st == 1) && st + 1 < 6) dma_w(st + 1, (st + 1) & 1);   // (stage 1: behind the patch requests, below)
      if constexpr (PERSIST) {
        // (measured, not kept: s_setprio 3 around these chores -- no change: what made a producer's stage-1 work 4.3 k cycles
        // was not issue arbitration but the patch loads queueing behind the stage's seven 1-KiB weight pieces)
        int lane_p = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (not kept across the stages)
        asm volatile("" : "+v"(lane_p));
        const int ptid = (wave_u - 4) * 64 + lane_p;
        if (st == 4 && tile + 2 * gstride < ntiles) nxt_pre = p.pc_tab ? decode_tab(k_walk + 2) : decode(tile + 2 * gstride);
        if (st == 1 && has_next) {   // (stage 1: the producers' lightest -- stage 0 carries the walk's bookkeeping)
          // x-PAIRS of patch elements, one 8-byte load each (a load instruction costs a producer wave 100-200 cycles beside
          // the consumers' stream: 12 wave-level loads instead of 20).  With an even level width a pair is inside or outside
          // the image as a whole and 8-byte aligned: the patch starts at column tx0 - 2 (even).
#pragma unroll
          for (int k = 0; k < NPF2; ++k) {
            const int pi = ptid + 256 * k;                       // pair index: row (ci, py) = pi / 10, column pair pi % 10
            const int row = (pi * 6554) >> 16, c2 = pi - row * (PW / 2);
            const int ci = (row * 3277) >> 16, py = row - ci * PH;
            const int gy = nxt.ty0 - 2 + py, gx = nxt.tx0 - 2 + 2 * c2;
            const bool in = pi < 3 * PH * (PW / 2) && (unsigned)gy < (unsigned)nxt.H && (unsigned)gx < (unsigned)nxt.W;
            const unsigned off = (unsigned)((ci * nxt.H + gy) * nxt.W + gx);
            const bool in1 = in && gx + 1 < nxt.W;
            float2 v2 = make_float2(0.f, 0.f);
            if (in1 && !(off & 1u)) {
              v2 = *(const float2*)(nxt.img + off);
            } else {   // (an odd level width -- never the detector's, whose levels are padded to multiples of 16: element by element)
              if (in) v2.x = nxt.img[off];
              if (in1) v2.y = nxt.img[off + 1];
            }
            pvn[2 * k] = v2.x;
            pvn[2 * k + 1] = v2.y;
          }
        }
        // (the patch requests go out FIRST in their stage: issued behind the stage's seven 1-KiB weight pieces they queued for
        // 2-3 k cycles with the wave stuck at the issue -- a stage 1 of 4.3 k cycles instead of 3.7; 2.7 is the consumers')
        if (st == 1) dma_w(2, 0);
        if (st == 3 && has_next && ptid == 0) {   // (the previous tile's claims ended before stage 0; read behind the post-K barrier)
          ctrL[0] = 0u;
          ctrL[1] = (nxt.ty0 >= 1 && nxt.tx0 >= 1 && nxt.ty0 + TH < nxt.H && nxt.tx0 + TW < nxt.W) ? 1u : 0u;
          // the tile's geometry for every wave of the block (the decode is ~2 k cycles of dependent scalar loads: it ran on
          // this wave under stage 0; the others used to repeat it at the end of their tile)
          geoL[0] = (unsigned)nxt.b; geoL[1] = (unsigned)nxt.ty0; geoL[2] = (unsigned)nxt.tx0; geoL[3] = (unsigned)nxt.H;
          geoL[4] = (unsigned)nxt.W;
          const unsigned long long q0 = (unsigned long long)nxt.img, q1 = (unsigned long long)nxt.out, q2 = (unsigned long long)nxt.pool,
                                   q3 = (unsigned long long)nxt.out_amax, q4 = (unsigned long long)nxt.pool_amax;
          geoL[6] = (unsigned)q0; geoL[7] = (unsigned)(q0 >> 32); geoL[8] = (unsigned)q1; geoL[9] = (unsigned)(q1 >> 32);
          geoL[10] = (unsigned)q2; geoL[11] = (unsigned)(q2 >> 32); geoL[12] = (unsigned)q3; geoL[13] = (unsigned)(q3 >> 32);
          geoL[14] = (unsigned)q4; geoL[15] = (unsigned)(q4 >> 32);
        }
        if (st == 2 && has_next) {   // (stage 2: the producers' lightest; the loads were waited for at its top)
#pragma unroll
          for (int k = 0; k < NPF2; ++k) {
            const int pi = ptid + 256 * k;
            if (pi < 3 * PH * (PW / 2)) *(uint2*)(patch + 2 * pi) = make_uint2(patch_word(pvn[2 * k]), patch_word(pvn[2 * k + 1]));
            amax1 = conv_absmax_bits(conv_absmax_bits(amax1, pvn[2 * k]), pvn[2 * k + 1]);
          }
#pragma unroll
          for (int k = 0; k < (HPP + 255) / 256; ++k) {
            const int hp = ptid + 256 * k;
            const int qy = (hp * 58255) >> 20, qx = hp - qy * HTW;
            if (hp < HPP)
              valid[hp] = (hp < HP && (unsigned)(nxt.ty0 - 1 + qy) < (unsigned)nxt.H && (unsigned)(nxt.tx0 - 1 + qx) < (unsigned)nxt.W) ? 1 : 0;
          }
        }
        // (buffer 0 held stage 4's weights; every consumer is past them behind this stage's barrier)
        if (st == 5 && has_next) dma_w(0, 0);
      }
    }
  }

  PC_T();
  // PERSIST (round 4): conv1_1 of the NEXT tile runs on the four producer waves WHILE the consumers store this tile -- both
  // are vector-ALU phases, the epilogue latency-bound on one wave per SIMD (~10 cycles per instruction), so the two streams
  // share a SIMD's issue slots instead of queueing (measured: a matrix stream leaves a partner wave ~3 vector issues per
  // MFMA, tools/scratch/coissue.hip -- conv1_1 under the K loop was the wrong place).  The halo tiles are free once every
  // consumer has issued its last fragment read: one more barrier; the next tile's patch and flags were parked in stage 2.
  if (PERSIST && has_next) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // every consumer has issued its last fragment read: the halo tiles are the producers'
  }
#ifdef SHF_CONV_TIMING
  const unsigned long long t_barx = __builtin_amdgcn_s_memtime();
#endif
  // epilogue: the four consumer waves store from registers (conv_common.h conv_epilogue_regs: half-wave exchange, 16
  // consecutive couts per lane, fused 2x2 max-pool as a DPP quad max) -- no LDS round trip, no barrier
  float amax = 0.f;  // this layer's stored outputs: fp16 range guard (with conv1_1's, amax1) + activation exponent
  if (consumer) {
    const bool relu = (p.relu & 1) != 0, write_main = !(p.relu & 8), main_split = (p.relu & 32) != 0,
               pool_split = (p.relu & 64) != 0;
    // (the lane's coordinates are formed AGAIN here, from the lane id the hardware hands out: kept alive across the K loop
    // they are spilled, and a scratch reload between the stores waits for every store issued so far)
    int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_e));
    const int i_e = lane_e & 31, kh_e = lane_e >> 5;
    int px_e, dy_e;
    row_to_pixel(i_e, dy_e, px_e);
    asm volatile("" : "+v"(seen), "+v"(seenp));   // (the compiler's wait for the peeks goes here, before the first store)
    seen = __builtin_amdgcn_readfirstlane(seen);
    seenp = __builtin_amdgcn_readfirstlane(seenp);
    // (kernel arguments every accumulator tile uses: kept in vector registers, not re-read from the argument segment)
    int out_stride_e = p.out_stride, pool_stride_e = p.pool_stride;
    asm volatile("" : "+v"(out_stride_e), "+v"(pool_stride_e));
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const bool interior = ty0 + TH <= H && tx0 + TW <= W;
    const int x = tx0 + px_e;
    if (relu && !write_main && mem.pool) {
      // the un-pooled map is not stored (conv1_2 -> pool1 of VGG-16): the pool-only epilogue (conv_common.h), same bits
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float4 bias_q[4];   // (one LDS round trip per cout half, not one per register quad of every tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_q[q] = *(const float4*)(bias2L + tn * 32 + 8 * q + 4 * kh_e);
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
          int y = ty0 + wm * 2 * MT + tm * 2 + dy_e;
          asm volatile("" : "+v"(y));
          const bool vld = y < H && x < W;
          const bool window = y - dy_e < H && x - (i_e & 1) < W;
          const unsigned pix_q = (unsigned)((b * Hp + (y >> 1)) * Wp + (x >> 1));
          float* pq = mem.pool + (size_t)pix_q * (unsigned)pool_stride_e;
          const f32x16 am_ = accm[tm][tn], ac_ = accc[tm][tn];
          conv_epilogue_pool_only<true>([&](int r) { return __builtin_fmaf(ac_[r], LO_INV, am_[r]); }, [&](int q) { return bias_q[q]; },
                                        vld, window, interior, pq, tn * 32, kh_e, i_e & 3, pool_split, amax);
        }
        PC_T();
      }
    } else {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int cout16 = tn * 32 + kh_e * 16;
      float4 bias16[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bias16[g] = *(const float4*)(bias2L + cout16 + 4 * g);   // (LDS: no vmcnt wait between the tiles' stores)
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) {
        int y = ty0 + wm * 2 * MT + tm * 2 + dy_e;
        asm volatile("" : "+v"(y));   // (this tile's address arithmetic starts here: see the dual-tile kernel's epilogue)
        const bool vld = y < H && x < W;
        const unsigned pix_m = (unsigned)((b * H + y) * W + x), pix_q = (unsigned)((b * Hp + (y >> 1)) * Wp + (x >> 1));
        float* pm = write_main ? gout + (size_t)pix_m * (unsigned)out_stride_e : nullptr;
        float* pq = mem.pool ? mem.pool + (size_t)pix_q * (unsigned)pool_stride_e : nullptr;
        if (relu)
          conv_epilogue_regs<true>(accm[tm][tn], accc[tm][tn], LO_INV, bias16, vld, interior, pm, cout16, main_split, pq,
                                   vld && (i_e & 3) == 0, pool_split, amax);
        else
          conv_epilogue_regs<false>(accm[tm][tn], accc[tm][tn], LO_INV, bias16, vld, interior, pm, cout16, main_split, pq,
                                    vld && (i_e & 3) == 0, pool_split, amax);
      }
      PC_T();
    }
    }
    conv_amax_commit(mem.out_amax, seen, mem.pool ? mem.pool_amax : nullptr, seenp, amax);   // (producers hold no outputs)
  }
  if (PERSIST && has_next) {
    // the next tile's conv1_1: its 11 row tiles are CLAIMED one at a time (an LDS counter) by whichever wave is free -- the
    // producers from the barrier on, the consumers once their epilogue is out.  (Measured and dropped: row tiles split per
    // channel chunk -- 22 finer items -- cost more than their better balance gives, 1633 vs 1595 us: the fragments are built
    // twice and an item is one dependent chain; channel chunk 0 under stages 3-5 on the producer waves lengthens the K
    // loop by exactly what the producers run, with or without s_setprio: a matrix stream leaves its SIMD partner ~3 vector
    // issues per MFMA, tools/scratch/coissue.hip.)
    int lane_p = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (not kept across the K loop)
    asm volatile("" : "+v"(lane_p));
    i1 = lane_p & 31;
    kh1 = lane_p >> 5;
    const bool halo_inside = __builtin_amdgcn_readfirstlane((int)ctrL[1]) != 0;
#pragma unroll 1
    for (;;) {
      unsigned got = 0u;
      if (lane_p == 0) got = atomicAdd(ctrL, 1u);
      const int m = __builtin_amdgcn_readfirstlane((int)got);
      if (m >= NMT) break;
      conv1_tile(m, 0, 2, halo_inside);
    }
    amax1 = conv_absmax_bits(amax1, fmaxf((float)amax1h[0], (float)amax1h[1]));
  }
  conv_raise_range_flag(p.range_flag, conv_absmax_bits(amax, amax1));
#ifdef SHF_CONV_TIMING
  {
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();
    // per tile: first stage's barrier passed (tt[3]) .. K loop done (tt[9]) .. barrier X .. role work done; tail = from the
    // previous tile's role end to this tile's first stage start (the wait for the other role)
    if (n_walk > 0) ts_tail += tt[3] - t_role_end;
    ts_k += tt[9] - tt[3];
    for (int q = 0; q < 6; ++q) ts_st[q] += tt[4 + q] - tt[3 + q];
    ts_bar += t_barx - tt[9];
    ts_role += t_now - t_barx;
    t_role_end = t_now;
    ++n_walk;
  }
#endif
  if (!PERSIST || !has_next) break;
  // the walk's next tile: its patch, flags and first weight stage are in flight or parked; conv1_1 may overwrite the halo
  // tiles once every consumer is out of the K loop (they are: the epilogue is behind it)
  tile += gstride;
  ++k_walk;
  {
    auto rd = [&](int k) { return (unsigned)__builtin_amdgcn_readfirstlane((int)geoL[k]); };
    auto rd64 = [&](int k) { return (unsigned long long)rd(k) | ((unsigned long long)rd(k + 1) << 32); };
    mem = TileGeo{(int)rd(0), (int)rd(1), (int)rd(2), (int)rd(3), (int)rd(4), (const float*)rd64(6), (float*)rd64(8), (float*)rd64(10),
                  (unsigned*)rd64(12), (unsigned*)rd64(14)};
  }
  b = mem.b; ty0 = mem.ty0; tx0 = mem.tx0; H = mem.H; W = mem.W; gout = mem.out;
  amax1 = 0.f;
  amax1h = half2v{(_Float16)0, (_Float16)0};
  // (opaque per tile: what conv1_1 derives from the lane's coordinates -- 16 patch offsets, row addresses -- is formed again
  // for every tile instead of living in registers across the K loop)
  asm volatile("" : "+v"(lane));
  // (no barrier here: stage 0's orders the producers' conv1_1 stores before the consumers' first fragment reads)
  }
  PC_T();
#ifdef SHF_CONV_TIMING
  // tt: 0 entry, 1 patch requested + parked, 2 barrier, 3..8 the six stages' starts, 9 K loop done, (consumers: 10, 11 the
  // two cout halves stored,) last: flags published.  A first-round block (100) and two steady-state ones.
  if ((bid == 100 || bid == 9000 || bid == 20000) && lane == 0 && (wave == 0 || wave == 4))
    printf("[pc-own] blk%d wave%d own work per stage (before its closing barrier) %llu %llu %llu %llu %llu\n", bid, wave,
           ts_own[0] / n_walk, ts_own[1] / n_walk, ts_own[2] / n_walk, ts_own[3] / n_walk, ts_own[4] / n_walk);
  if ((bid == 100 || bid == 9000 || bid == 20000) && lane == 0 && (wave == 0 || wave == 4))
    printf("[pc] blk%d wave%d tiles %d | per tile: K loop %llu, wait at the post-K barrier %llu, role work (wave 0: epilogue, wave 4: next tile's conv1_1) %llu, wait for stage 0 %llu | mean stages %llu %llu %llu %llu %llu %llu\n",
           bid, wave, n_walk, ts_k / n_walk, ts_bar / n_walk, ts_role / n_walk,
           n_walk > 1 ? ts_tail / (n_walk - 1) : 0ull, ts_st[0] / n_walk, ts_st[1] / n_walk, ts_st[2] / n_walk, ts_st[3] / n_walk, ts_st[4] / n_walk, ts_st[5] / n_walk);
#endif
#undef PC_T
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
size_t split16_conv_weight_halfs(int Cout, int Cin, int k) { return (size_t)Cout * (Cin / 32) * k * k * 72; }

// host-side round-to-nearest-even fp32 -> bf16, returned as the fp16-typed bit pattern the packs store
static _Float16 host_bf16_as_half(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;                 // NaN stays a NaN
  else u += 0x7fffu + ((u >> 16) & 1u);
  const uint16_t b = (uint16_t)(u >> 16);
  _Float16 h;
  memcpy(&h, &b, 2);
  return h;
}

// (Cout,Cin,3,3) fp32 -> [Cin/32][ky][kx][Cout][hi 32 | lo 32 | 8 pad] fp16 (144-B rows = the LDS image)
// bf: the bf16 mode's pack -- hi = bf16(w) bit patterns, lo = 0
void pack_conv_weights_split16(const float* w, int Cout, int Cin, int k, void* dst_, bool bf) {
  _Float16* dst = (_Float16*)dst_;
  const int taps = k * k;
  memset(dst_, 0, split16_conv_weight_halfs(Cout, Cin, k) * 2);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const float x = w[((size_t)co * Cin + ci) * taps + t];
        const _Float16 h = bf ? host_bf16_as_half(x) : (_Float16)x;
        const _Float16 l = bf ? (_Float16)0 : (_Float16)((x - (float)h) * f16x3::LO_SCALE);
        const size_t row = (((size_t)(ci / 32) * taps + t) * Cout + co) * 72;
        dst[row + (ci % 32)] = h;
        dst[row + 32 + (ci % 32)] = l;
      }
}

size_t split16h_conv_weight_halfs(int Cout, int Cin, int k) { return (size_t)Cout * (Cin / 16) * k * k * 32; }

// (Cout,Cin,3,3) fp32 -> [Cin/16][ky][kx][Cout][4 x 8 halfs] fp16: 64-B rows, NO padding (every byte of the pack is
// fetched by every block: padding is weight traffic) = the dual-tile kernel's LDS image.  The four 16-byte pieces of a row
// -- hi k 0-7, hi k 8-15, lo k 0-7, lo k 8-15 -- are rotated by (row / 4) mod 4 so that the 16 lanes of a ds_read_b128
// group (16 consecutive rows, one piece each) still fall on 16 different bank groups.
// lo is NOT scaled here: lo = fp16(w s - hi) with one power of two s per layer that lifts the weights to [8, 16) at the top,
// so that the low parts of all but the tiniest weights are normal fp16 numbers (the MFMA honours subnormals anyway:
// tools/mfma_denorm.hip) and the three products share one accumulator.  Returns 1 / s for the epilogue.
float pack_conv_weights_split16h(const float* w, int Cout, int Cin, int k, void* dst_, bool bf) {
  _Float16* dst = (_Float16*)dst_;
  const int taps = k * k;
  float amax = 0.f;
  for (size_t i = 0; i < (size_t)Cout * Cin * taps; ++i) amax = std::max(amax, std::fabs(w[i]));
  int e = 0;
  if (amax > 0.f) e = (int)std::floor(std::log2(8.0 / (double)amax));
  e = std::max(-14, std::min(14, e));
  if (bf) e = 0;   // bf16 has fp32's exponent range: nothing to lift
  const float s = std::ldexp(1.0f, e);
  memset(dst_, 0, split16h_conv_weight_halfs(Cout, Cin, k) * 2);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const float x = w[((size_t)co * Cin + ci) * taps + t] * s;
        const _Float16 h = bf ? host_bf16_as_half(x) : (_Float16)x;
        const _Float16 l = bf ? (_Float16)0 : (_Float16)(x - (float)h);
        const size_t row = (((size_t)(ci / 16) * taps + t) * Cout + co) * 32;
        const int kk = ci % 16, rot = ((co & 127) >> 2) & 3;
        dst[row + (((kk >> 3) + rot) & 3) * 8 + (kk & 7)] = h;
        dst[row + ((2 + (kk >> 3) + rot) & 3) * 8 + (kk & 7)] = l;
      }
  return 1.0f / s;
}


// Environment knobs (experiments; the defaults are the measured best), read ONCE: none of them is consulted per launch.
namespace {
struct Knobs {
  int w4_mode;         // SHF_F16X3_W4: -1 auto (Cin >= 64), 0 never, 1 always -- which layers take the 4-wave dual-tile family
  int w4_mt;           // SHF_F16X3_W4_MT: 0 auto, 2 / 4 force 8- / 16-row tiles
  int w4d_ntile;       // SHF_F16X3_W4D_NTILE: 0 auto (hybrid launches), 1 / 2 force single- / two-tile blocks
  int xcd_remap;       // SHF_F16X3_XCD_REMAP: 1 = all cout tiles of a pixel tile on one XCD (experiment)
  bool pc, dilated, k1, scalar_epilogue;   // SHF_F16X3_PC, SHF_F16X3_DILATED, SHF_F16X3_1X1 (default on), SHF_CONV_SCALAR_EPILOGUE (off)
  bool pc_persist;     // SHF_F16X3_PC_PERSIST (default on): the fused first pair as one block per CU walking the tiles
  int cus;
};
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
const Knobs& knobs() {
  static const Knobs k = [] {
    Knobs q;
    q.w4_mode = env_int("SHF_F16X3_W4", -1);
    q.w4_mt = env_int("SHF_F16X3_W4_MT", 0);
    q.w4d_ntile = env_int("SHF_F16X3_W4D_NTILE", 0);
    q.xcd_remap = env_int("SHF_F16X3_XCD_REMAP", 0);
    q.pc = env_int("SHF_F16X3_PC", 1) != 0;
    q.pc_persist = env_int("SHF_F16X3_PC_PERSIST", 1) != 0;
    q.dilated = env_int("SHF_F16X3_DILATED", 1) != 0;
    q.k1 = env_int("SHF_F16X3_1X1", 1) != 0;
    q.scalar_epilogue = env_int("SHF_CONV_SCALAR_EPILOGUE", 0) != 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 4)
      cus = 256;
    q.cus = cus;
    return q;
  }();
  return k;
}
// 16-byte aligned channel views: what the vector epilogues (LDS-transposed and register) need
bool views_aligned(const ConvArgs* as, int n) {
  for (int i = 0; i < n; ++i) {
    const ConvArgs& q = as[i];
    if ((q.out.cstride % 4) || (q.out.coff % 4) || ((uintptr_t)q.out.p & 15)) return false;
    if (q.pool.p && ((q.pool.cstride % 4) || (q.pool.coff % 4) || ((uintptr_t)q.pool.p & 15))) return false;
  }
  return true;
}
}  // namespace

// (net.cpp: will launch_conv_f16x3_group(as, n) take the dual-tile family?  Then the sub-launch hook does the profiling.)
// The family addresses its input with 32-bit BYTE offsets from the member's base: inputs of 4 GiB and more, and
// unaligned views (scalar epilogue), take the 8-wave kernel.
bool conv_f16x3_group_is_dual(const ConvArgs* as, int n) {
  if (!as[0].wsplit16h || as[0].img || as[0].k != 3 || as[0].dil != 1 || as[0].out.C % 128) return false;
  if (!conv_f16x3_uses_w4(as[0].in.C) || knobs().scalar_epilogue || !views_aligned(as, n)) return false;
  for (int i = 0; i < n; ++i)
    if ((unsigned long long)as[i].in.B * as[i].in.H * as[i].in.W * as[i].in.cstride * 4ull >= (1ull << 32)) return false;
  return true;
}

void pack_first_conv_frags(const float* w, void* dst_, bool bf) {
  _Float16* dst = (_Float16*)dst_;
  for (int n = 0; n < 2; ++n)
    for (int kk = 0; kk < 2; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int i = lane & 31, kh = lane >> 5, k = first_conv_slot_tap(kk, kh, j);
          const float x = k >= 0 ? w[(size_t)(n * 32 + i) * 27 + k] : 0.f;
          const _Float16 h = bf ? host_bf16_as_half(x) : (_Float16)x;
          const _Float16 l = bf ? (_Float16)0 : (_Float16)((x - (float)h) * f16x3::LO_SCALE);
          dst[(((size_t)(n * 2 + kk) * 2 + 0) * 64 + lane) * 8 + j] = h;
          dst[(((size_t)(n * 2 + kk) * 2 + 1) * 64 + lane) * 8 + j] = l;
        }
}

bool conv_f16x3_uses_pc() { return knobs().pc; }
bool conv_f16x3_pc_persistent() { return knobs().pc_persist; }

// (Cin 64 -- conv2_1 -- joined in round 3: as two single-tile 8-row blocks per CU it beats the 8-wave kernel, 0.67 vs 0.81 ms)
bool conv_f16x3_uses_w4(int Cin) { return knobs().w4_mode < 0 ? Cin >= 64 : knobs().w4_mode != 0; }

bool conv_f16x3_eligible(int Cin, int Cout, int k, int pad, int dil) {
  const bool dil_ok = dil == 1 || (knobs().dilated && (dil == 2 || dil == 4));
  if (k == 1) return knobs().k1 && pad == 0 && Cin % 32 == 0 && Cout % 64 == 0;
  return k == 3 && dil_ok && pad == dil && Cin % 32 == 0 && Cout % 64 == 0;
}

// 4-wave family: 16-row tiles (MT 4) or 8-row tiles (MT 2)?  A launch runs in ceil(blocks / CUs) rounds of one block
// per CU; an 8-row block costs ~0.56 of a 16-row one (half the MFMAs, the same weight traffic per stage and the same
// prologue / epilogue latencies).  SHF_F16X3_W4_MT = 2 / 4 forces the choice (experiments).
// Short K loops (Cin <= 128: 24 stages) are the exception: there a block's prologue (first ~50 KB of weights and halo)
// and epilogue (the output tile's store burst) are a third of its life, and the single-tile 8-row variant -- 66 KB of
// LDS, 200 registers: TWO blocks per CU, one's epilogue under the other's K loop -- wins although it moves four times
// the weight bytes per MFMA of a two-tile 16-row block.  Measured per layer on one box (tools/variant_layers.sh, us under
// rocprofv3): conv2_2 1212 vs 1278, conv3_1 625 vs 672, head_1 102 vs 109; from Cin 256 up it loses (conv3_2 1186 vs 1134,
// conv4_2 1196 vs 1072).
static bool w4_short_k(const ConvArgs* as) { return as[0].in.C <= 128; }

static int w4_pick_mt(const ConvArgs* as, int n, int nct) {
  if (knobs().w4_mt == 2 || knobs().w4_mt == 4) return knobs().w4_mt;
  if (w4_short_k(as)) return 2;
  const int cus = knobs().cus;
  long long t4 = 0, t2 = 0;
  for (int i = 0; i < n; ++i) {
    const long long tx = (as[i].in.W + f16x3::TW - 1) / f16x3::TW, B = as[i].in.B;
    t4 += B * tx * ((as[i].in.H + 15) / 16);
    t2 += B * tx * ((as[i].in.H + 7) / 8);
  }
  const double c4 = (double)((t4 * nct + cus - 1) / cus), c2 = 0.56 * (double)((t2 * nct + cus - 1) / cus);
  return c2 < c4 ? 2 : 4;
}

int conv_f16x3_w4_mt(const ConvArgs* as, int n) { return w4_pick_mt(as, n, as[0].out.C / 128); }

template <int BN, bool FUSE1, int DIL = 1, int KS = 3>
static int launch_f16x3_t(const ConvArgs* as, int n, hipStream_t s) {
  using namespace f16x3;
  constexpr int PADH = KS == 3 ? DIL : 0;
  constexpr int HP = (TH + 2 * PADH) * (TW + 2 * PADH);
  const ConvArgs& a = as[0];
  ConvK p;
  p.wp = (const float*)a.wsplit16;
  p.wph = nullptr;
  p.wscale_inv = 1.f;
  p.tile_base = 0;
  p.ntile_blocks = 0;
  p.xcd_remap = knobs().xcd_remap;
  p.pc_tab = 0;
  p.bias = a.bias;
  p.Cin = a.in.C; p.Cout = a.out.C;
  p.in_stride = a.in.cstride; p.out_stride = a.out.cstride;
  p.dil = DIL; p.relu = a.relu | (a.pool.p && !a.write_main ? 8 : 0);
  p.pool_stride = a.pool.p ? a.pool.cstride : 0;
  p.nct = p.Cout / BN;
  p.nmem = n;
  for (int i = 0; i < MAX_GROUP; ++i) p.tile_starts[i] = 0x7fffffff;
  p.dbg = nullptr;
  p.range_flag = a.range_flag;
  p.w1t = a.w1t;
  p.w1f = a.w1f;
  p.b1 = a.b1;
  long long tiles = 0;
  const bool vec_ok = !knobs().scalar_epilogue && views_aligned(as, n);
  // the dual-tile 4-wave family (16- or 8-row tiles, w4_pick_mt) or this template's 8-wave kernel
  const bool dual = BN == 128 && !FUSE1 && DIL == 1 && KS == 3 && conv_f16x3_group_is_dual(as, n);
  const int mt = dual ? w4_pick_mt(as, n, p.nct) : 4;
  const int th = 4 * mt;
  for (int i = 0; i < n; ++i) {
    const ConvArgs& q = as[i];
    if (FUSE1 && !q.img) { set_error("conv f16x3: fused first layer needs the image pointer"); return -1; }
    if (q.in.C != p.Cin || q.out.C != p.Cout || q.in.cstride != p.in_stride || q.out.cstride != p.out_stride ||
        q.wsplit16 != a.wsplit16 || q.in_split != a.in_split || q.out_split != a.out_split ||
        q.pool_split != a.pool_split) {
      set_error("conv group: members must share the layer");
      return -1;
    }
    ConvMember& m = p.m[i];
    m.in = q.in.p + q.in.coff;
    m.out = q.out.p + q.out.coff;
    m.pool = q.pool.p ? q.pool.p + q.pool.coff : nullptr;
    m.img = q.img;
    m.in_amax = q.in_amax; m.out_amax = q.out_amax; m.pool_amax = q.pool.p ? q.pool_amax : nullptr;
    m.B = q.in.B; m.H = q.in.H; m.W = q.in.W;
    m.tiles_x = (m.W + TW - 1) / TW;
    m.tiles_per_img = m.tiles_x * ((m.H + th - 1) / th);
    m.inv_tiles_x = conv_inv32(m.tiles_x);
    m.inv_tiles_per_img = conv_inv32(m.tiles_per_img);
    m.tile_start = (int)tiles;
    p.tile_starts[i] = (int)tiles;
    tiles += (long long)m.tiles_per_img * m.B;
    // conv_split_tile's multiply-high quotients are exact while tile index x divisor < 2^32
    if ((unsigned long long)m.tiles_per_img * m.B * (unsigned long long)m.tiles_per_img >= (1ull << 32)) {
      set_error("conv f16x3: more than 2^32 / tiles-per-image pixel tiles in one member (shrink the batch or the map)");
      return -1;
    }
  }
  if (tiles * p.nct >= (1ll << 31)) { set_error("conv f16x3: grid too large"); return -1; }
  if (vec_ok) p.relu |= 16;
  if (a.out_split || a.pool_split) {
    if (!vec_ok) { set_error("conv f16x3: split-format output needs the aligned epilogue"); return -1; }
    p.relu |= (a.out_split ? 32 : 0) | (a.pool_split ? 64 : 0);
  }
  // (the transposed epilogue needs 256 x (BN + 4) floats: the 1x1 variant's K-loop buffers are smaller than that)
  const size_t lds = std::max((size_t)HP * ROWB + 2 * KS * (size_t)BN * ROWB, (size_t)256 * (BN + CS_PAD) * sizeof(float)) +
                     (FUSE1 ? (3 * (TH + 4) * (TW + 4) + 27 * 64 + 64) * sizeof(float) : 0);
#ifdef SHF_CONV_TIMING
  static unsigned long long* dbg_dev = nullptr;
  if (!dbg_dev) hipMalloc((void**)&dbg_dev, 16 * 5 * 8);
  hipMemset(dbg_dev, 0, 16 * 5 * 8);
  p.dbg = dbg_dev;
#endif
  if (FUSE1 && BN == 64 && conv_f16x3_uses_pc() && vec_ok && p.Cin == 64 && p.Cout == 64 && p.w1f) {
    // two halo tiles (both channel chunks of conv1_1's output) + the weight double buffer + the image patch
    constexpr size_t HPP = (HP + 31) / 32 * 32;
    // (+ conv1_1's weight fragments 8 KiB, its 64 biases, the row-tile counter: 162 512 B of the 160 KiB)
    const size_t lds_pc = 2 * (size_t)HP * ROWB + 2 * 3 * (size_t)BN * ROWB + (3 * (TH + 4) * (TW + 4) + 8) * sizeof(float) + HPP +
                          BN * sizeof(float) + 8192 + 64 * sizeof(float) + 16 + 64 + 300 * 4;
if (lds_pc > 160 * 1024) { set_error("conv f16x3: the fused first pair does not fit the LDS"); return -1; }
    if (knobs().pc_persist) {
      // one block per CU walks the tiles (tile = block, block + grid, ...)
      p.ntile_blocks = (int)tiles;
      const dim3 gp((unsigned)std::min<long long>(tiles, knobs().cus));
      {   // the per-block tile table: fits (300 tiles per block) and packs (image < 256, tile row / column < 1024)?
        bool ok = (tiles + gp.x - 1) / gp.x <= 300;
        for (int i = 0; i < n; ++i)
          ok = ok && as[i].in.B <= 255 && p.m[i].tiles_x <= 1023 && p.m[i].tiles_per_img / std::max(1, p.m[i].tiles_x) <= 1023;
        p.pc_tab = ok ? 1 : 0;
      }
      if (a.bf16) hipLaunchKernelGGL((conv_mfma_f16x3_pc_kernel<1, true, true>), gp, dim3(512), lds_pc, s, p);
      else if (a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_pc_kernel<3, false, true>), gp, dim3(512), lds_pc, s, p);
      else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_pc_kernel<2, false, true>), gp, dim3(512), lds_pc, s, p);
      else hipLaunchKernelGGL((conv_mfma_f16x3_pc_kernel<1, false, true>), gp, dim3(512), lds_pc, s, p);
    } else if (a.bf16) hipLaunchKernelGGL((conv_mfma_f16x3_pc_kernel<1, true>), dim3((unsigned)tiles), dim3(512), lds_pc, s, p);
    else if (a.nprod >= 3) hipLaunchKernelGGL(conv_mfma_f16x3_pc_kernel<3>, dim3((unsigned)tiles), dim3(512), lds_pc, s, p);
    else if (a.nprod == 2) hipLaunchKernelGGL(conv_mfma_f16x3_pc_kernel<2>, dim3((unsigned)tiles), dim3(512), lds_pc, s, p);
    else hipLaunchKernelGGL(conv_mfma_f16x3_pc_kernel<1>, dim3((unsigned)tiles), dim3(512), lds_pc, s, p);
  } else if (dual) {
    // dual-tile family (conv_mfma_f16x3_w4d_kernel<.., MT, NTILE, ..>): every variant forms an output with the same
    // operations in the same order, so the choice below -- two tiles per block where that fills whole rounds of one
    // block per CU, single tiles for the rest -- never changes a result.
    p.wph = a.wsplit16h;
    p.wscale_inv = a.wscale_inv;
    const long long per_round = knobs().cus / p.nct > 0 ? knobs().cus / p.nct : 1;   // pixel tiles (single) or pairs (dual) per round
    const long long pairs = (tiles + 1) / 2;
    const double c1 = mt == 4 ? 1.0 : 0.56, c2 = mt == 4 ? 1.82 : 1.02;   // block cost: one / two tiles (w4_pick_mt's unit)
    const long long full2 = pairs / per_round;                  // whole rounds of dual blocks
    const long long rest = std::max(0LL, tiles - 2 * full2 * per_round);
    const double cost_all1 = c1 * (double)((tiles + per_round - 1) / per_round);
    const double cost_all2 = c2 * (double)((pairs + per_round - 1) / per_round);
    const double cost_hyb = c2 * (double)full2 + c1 * (double)((rest + per_round - 1) / per_round);
    long long n2 = 0;                                           // pixel tiles covered by the dual launch
    if (cost_all2 <= cost_all1 && cost_all2 <= cost_hyb) n2 = tiles;
    else if (cost_hyb < cost_all1) n2 = 2 * full2 * per_round;
    if (w4_short_k(as) && knobs().w4_mt == 0) n2 = 0;          // two single-tile blocks per CU (w4_pick_mt)
    if (knobs().w4d_ntile == 1) n2 = 0;
    if (knobs().w4d_ntile == 2) n2 = tiles;
    // two buffer sets of halo tiles (4 planes of (th + 2) rows x 24 pixels x 16 B, + 32 B), the weight double buffer, biases
    const size_t as_b = 4 * ((size_t)(th + 2) * 24 * 16 + 32);
    const size_t lds1 = 2 * as_b + 2 * 3 * (size_t)BN * 64 + BN * sizeof(float), lds2 = lds1 + 2 * as_b;
#define SHF_W4D_LAUNCH(SPLIT, MTV, NTV, GRID, LDS)                                                                        \
    {                                                                                                                    \
      if (a.bf16) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<false, MTV, NTV, 1, true>), GRID, dim3(256), LDS, s, p);  \
      else if (a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 3>), GRID, dim3(256), LDS, s, p);    \
      else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 2>), GRID, dim3(256), LDS, s, p); \
      else hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 1>), GRID, dim3(256), LDS, s, p);                 \
    }
#define SHF_W4D_PICK(NTV, GRID, LDS)                                       \
    {                                                                     \
      if (mt == 4 && a.in_split) SHF_W4D_LAUNCH(true, 4, NTV, GRID, LDS)  \
      else if (mt == 4) SHF_W4D_LAUNCH(false, 4, NTV, GRID, LDS)          \
      else if (a.in_split) SHF_W4D_LAUNCH(true, 2, NTV, GRID, LDS)        \
      else SHF_W4D_LAUNCH(false, 2, NTV, GRID, LDS)                       \
    }
    const int vbase = (a.in_split ? 4 : 0) + (mt == 2 ? 2 : 0);
    if (n2 > 0) {
      p.tile_base = 0;
      p.ntile_blocks = (int)(n2 * p.nct);
      const dim3 g2((unsigned)(((n2 + 1) / 2) * p.nct));
      if (a.sub_hook) a.sub_hook(a.sub_ctx, 0, vbase, (double)n2 / (double)tiles);
      SHF_W4D_PICK(2, g2, lds2)
      if (a.sub_hook) a.sub_hook(a.sub_ctx, 1, vbase, (double)n2 / (double)tiles);
    }
    if (n2 < tiles) {
      p.tile_base = (int)n2;
      p.ntile_blocks = (int)(tiles * p.nct);
      const dim3 g1((unsigned)((tiles - n2) * p.nct));
      if (a.sub_hook) a.sub_hook(a.sub_ctx, 0, vbase + 1, (double)(tiles - n2) / (double)tiles);
      SHF_W4D_PICK(1, g1, lds1)
      if (a.sub_hook) a.sub_hook(a.sub_ctx, 1, vbase + 1, (double)(tiles - n2) / (double)tiles);
    }
#undef SHF_W4D_PICK
#undef SHF_W4D_LAUNCH
  } else if (a.in_split) {
    set_error("conv f16x3: split-format input reached a kernel other than the 4-wave family (unaligned views, or an input of 4 GiB or more)");
    return -1;
  } else {
    const dim3 g8((unsigned)(tiles * p.nct));
    if (a.bf16 && FUSE1) { set_error("conv f16x3: bf16 mode runs the first pair on the producer/consumer kernel only"); return -1; }
    if (a.bf16) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<BN, false, DIL, KS, 1, true>), g8, dim3(512), lds, s, p);
    else if (FUSE1 || a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<BN, FUSE1, DIL, KS, 3>), g8, dim3(512), lds, s, p);
    else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_kernel<BN, false, DIL, KS, 2>), g8, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((conv_mfma_f16x3_kernel<BN, false, DIL, KS, 1>), g8, dim3(512), lds, s, p);
  }
  SHF_HIP_OK(hipGetLastError());
#ifdef SHF_CONV_TIMING
  {
    unsigned long long h[80];
    hipStreamSynchronize(s);
    hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 16; w += 3)
      if (h[w * 5 + 4])
        fprintf(stderr, "[f16x3 timing] blk%d wave%d stages %llu: per-stage cycles barrier %.0f issue %.0f compute %.0f tail %.0f\n",
                w / 8 ? 100 : 0, w % 8, h[w * 5 + 4], (double)h[w * 5] / h[w * 5 + 4], (double)h[w * 5 + 1] / h[w * 5 + 4],
                (double)h[w * 5 + 2] / h[w * 5 + 4], (double)h[w * 5 + 3] / h[w * 5 + 4]);
  }
#endif
  return 0;
}

int conv_f16x3_init_attributes() {
  (void)knobs();
#define SHF_LDS_ATTR(K) SHF_HIP_OK(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define SHF_8W_ATTR(BNV, DILV, KSV)                                       \
  SHF_LDS_ATTR((conv_mfma_f16x3_kernel<BNV, false, DILV, KSV, 3>))         \
  SHF_LDS_ATTR((conv_mfma_f16x3_kernel<BNV, false, DILV, KSV, 2>))         \
  SHF_LDS_ATTR((conv_mfma_f16x3_kernel<BNV, false, DILV, KSV, 1>))         \
  SHF_LDS_ATTR((conv_mfma_f16x3_kernel<BNV, false, DILV, KSV, 1, true>))
  SHF_8W_ATTR(128, 1, 3) SHF_8W_ATTR(128, 1, 1) SHF_8W_ATTR(64, 1, 1) SHF_8W_ATTR(64, 2, 3) SHF_8W_ATTR(64, 4, 3) SHF_8W_ATTR(64, 1, 3)
#undef SHF_8W_ATTR
  SHF_LDS_ATTR((conv_mfma_f16x3_kernel<64, true, 1, 3, 3>))
  SHF_LDS_ATTR(conv_mfma_f16x3_pc_kernel<3>)
  SHF_LDS_ATTR(conv_mfma_f16x3_pc_kernel<2>)
  SHF_LDS_ATTR(conv_mfma_f16x3_pc_kernel<1>)
  SHF_LDS_ATTR((conv_mfma_f16x3_pc_kernel<1, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_pc_kernel<3, false, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_pc_kernel<2, false, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_pc_kernel<1, false, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_pc_kernel<1, true, true>))
#define SHF_W4D_ATTR(SPLIT, MTV, NTV)                                      \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 3>))           \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 2>))           \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<SPLIT, MTV, NTV, 1>))
  SHF_W4D_ATTR(false, 4, 2) SHF_W4D_ATTR(true, 4, 2) SHF_W4D_ATTR(false, 4, 1) SHF_W4D_ATTR(true, 4, 1)
  SHF_W4D_ATTR(false, 2, 2) SHF_W4D_ATTR(true, 2, 2) SHF_W4D_ATTR(false, 2, 1) SHF_W4D_ATTR(true, 2, 1)
#undef SHF_W4D_ATTR
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 2, 1, true>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 1, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 2, 2, 1, true>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 2, 1, 1, true>))
#undef SHF_LDS_ATTR
  return 0;
}

int launch_conv_f16x3_group(const ConvArgs* as, int n, hipStream_t s) {
  if (n < 1 || n > MAX_GROUP) { set_error("conv group: 1..16 members"); return -1; }
  for (int i = 0; i < n; ++i)
    if ((as[i].in.cstride % 4) || (as[i].in.coff % 4)) { set_error("conv: input view not 16-byte aligned"); return -1; }
  if (!as[0].wsplit16) { set_error("conv f16x3: split weights not packed"); return -1; }
  if (as[0].img) {
    if (as[0].in.C != 64 || !as[0].w1t) { set_error("conv f16x3: fused first layer needs 64 channels + transposed weights"); return -1; }
    return launch_f16x3_t<64, true>(as, n, s);  // conv1_1 computed in place (BN=64 tile: Cout 64 or any multiple of 64)
  }
  if (as[0].k == 1)
    return (as[0].out.C % 128 == 0) ? launch_f16x3_t<128, false, 1, 1>(as, n, s) : launch_f16x3_t<64, false, 1, 1>(as, n, s);
  if (as[0].dil == 2) return launch_f16x3_t<64, false, 2>(as, n, s);
  if (as[0].dil == 4) return launch_f16x3_t<64, false, 4>(as, n, s);
  return (as[0].out.C % 128 == 0) ? launch_f16x3_t<128, false>(as, n, s) : launch_f16x3_t<64, false>(as, n, s);
}

}  // namespace shf
