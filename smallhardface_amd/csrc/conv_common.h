// Shared by the convolution translation units (conv.hip: exact fp32 MFMA; conv_f16x3.hip: split-fp16 MFMA).
#pragma once
#include "shf_internal.h"

namespace shf {

constexpr int MAX_GROUP = 16;

// One launch covers a GROUP of independent problems that share the layer (weights,
// channels, dilation) but not the spatial size: the 10 (level, flip) units of an image's
// test pyramid go through each conv layer in ONE grid, so the latency-bound small levels
// ride along with the large ones instead of paying a whole K loop on a mostly idle chip.
struct ConvMember {
  const float* in;   // already offset to the view's first channel
  float* out;        // already offset to the view's first channel
  const float* img;  // FUSE1 (conv_f16x3.hip): the raw NCHW image this layer's input is computed from, else null
  float* pool;       // fused 2x2/2 max-pool output (ceil(H/2) x ceil(W/2) x Cout, NHWC) or null
  int B, H, W;
  int tiles_x, tiles_per_img, tile_start;  // tile_start: first pixel-tile index of this member
  unsigned inv_tiles_x, inv_tiles_per_img; // floor(2^32 / d) + 1: a tile index is split with two multiply-highs, not divisions
  // activation exponent (split-fp16 modes): one u32 slot per (lane, blob) = the bit pattern of max |value| of this unit's
  // blob, raised by every producer's epilogue (conv_publish_amax) and read by the single-accumulator kernels, which lift
  // their input to the top of the fp16 range before the unscaled low parts are formed (conv_act_exponent).  Null = unknown.
  const unsigned* in_amax;
  unsigned* out_amax;
  unsigned* pool_amax;
  // the three shared-weight dilated heads in one launch (conv_f16x3_h3.h): `out` is dilation 1's output, these are
  // dilation 2's and dilation 4's (null for every other kernel)
  float* out2;
  float* out3;
  unsigned* out2_amax;
  unsigned* out3_amax;
};

struct ConvK {
  const float* wp;   // packed weights
  const void* wph;   // dual-tile 4-wave kernel: its own split-fp16 pack (16-channel chunks, unscaled low parts)
  float wscale_inv;  // ... and the power of two its epilogue multiplies back
  const float* bias;
  int Cin, Cout;
  int in_stride, out_stride;
  int dil, relu;     // relu bit 0: ReLU; bit 3 (8): do NOT write the un-pooled output; bit 4 (16): views are
                     // 16-byte aligned -> LDS-transposed float4 epilogue; bit 5 (32) / bit 6 (64): write the
                     // main / the pooled output in the split-fp16 activation format (ConvArgs::out_split)
  int pool_stride;   // floats per pixel of the pool buffer
  int nct, nmem;
  int ntile_blocks;  // persistent / dual-tile kernels: (pixel tiles of the group this launch ends at) x cout tiles
  int tile_base;     // dual-tile kernels: first pixel tile of this launch (a group may be covered by two launches)
  int pc_tab;        // fused first pair, persistent form: 1 = every block keeps the packed geometry of the tiles it walks in LDS
                     // (member 4 bits | image 8 | tile row 10 | tile column 10), decoded once by all its lanes in parallel
  const float* w1t;  // FUSE1: first-layer weights transposed to [27][64]
  const void* w1f;   // FUSE1 (producer/consumer kernel): the same as split-fp16 MFMA B fragments (pack_first_conv_frags)
  const float* b1;   // FUSE1: first-layer bias [64]
  unsigned long long* dbg;  // SHF_CONV_TIMING builds only: per-wave phase cycle sums
  int* range_flag;   // split-fp16 kernels: set to 1 when an output value leaves the fp16 range (|x| > 65504), or null
  int tile_starts[MAX_GROUP];  // m[q].tile_start again, contiguous (unused entries INT_MAX): ONE scalar load finds a
                               // block's member instead of a chain of dependent ones (~1.5 k cycles per block)
  ConvMember m[MAX_GROUP];
};

// row i (0..31) of a 32-row MFMA tile -> pixel inside the wave's 2x16 strip.
// The two low bits walk a 2x2 window so that the 4 consecutive C rows a lane owns
// form one pooling window (kept for a fused 2x2 max-pool epilogue).
// which member of the group does pixel tile `pt` belong to?
__device__ __forceinline__ int conv_find_member(const ConvK& p, int pt) {
  int mi = 0;
#pragma unroll
  for (int q = 1; q < MAX_GROUP; ++q) mi += (pt >= p.tile_starts[q]) ? 1 : 0;
  return mi;
}

// pixel tile `pt` of a member -> image b, tile row ty, tile column tx.  Integer division on the scalar unit is a
// ~40-instruction float sequence (two of them were ~1.5 k cycles of every block's prologue); n / d = umulhi(n, floor(2^32
// / d) + 1) is exact while n * d < 2^32 (tile counts are < 2^20).
__device__ __forceinline__ unsigned conv_div(unsigned n, unsigned d, unsigned inv) { return d == 1 ? n : __umulhi(n, inv); }
__device__ __forceinline__ void conv_split_tile(const ConvMember& mem, int pt, int& b, int& ty, int& tx) {
  b = (int)conv_div((unsigned)pt, (unsigned)mem.tiles_per_img, mem.inv_tiles_per_img);
  pt -= b * mem.tiles_per_img;
  ty = (int)conv_div((unsigned)pt, (unsigned)mem.tiles_x, mem.inv_tiles_x);
  tx = pt - ty * mem.tiles_x;
}
inline unsigned conv_inv32(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned long long)d) + 1u; }

__device__ __forceinline__ void row_to_pixel(int i, int& dy, int& px) {
  dy = (i >> 1) & 1;
  px = ((i >> 2) << 1) | (i & 1);
}

// Shared epilogue: bias + ReLU, store, and the optional fused MAX 2x2/2 pool (pooling_layer.cu:11-47).
// Registers 4q..4q+3 of a lane are C rows s + 8q + 4*(lane>>5), s = 0..3: with row_to_pixel they are
// the (dy,dx) = (s>>1, s&1) corners of ONE pooling window, so the pool is a max over four registers
// of the same lane -- no cross-lane traffic.  Windows on a ragged edge are clipped like Caffe's.
template <typename GetV>
__device__ __forceinline__ void conv_store_tile(GetV getv, float bv, int relu_flags, int gy0, int gx0, int kh,
                                                int H, int W, int b, int cout, float* __restrict__ gout,
                                                int out_stride, float* __restrict__ gpool, int pool_stride,
                                                float* amax = nullptr) {
  const bool write_main = !(relu_flags & 8);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int gx = gx0 + 2 * (2 * q + kh);
    float m = -3.402823466e+38f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int y = gy0 + (s >> 1), x = gx + (s & 1);
      float v = getv(4 * q + s) + bv;
      if (relu_flags & 1) v = fmaxf(v, 0.f);
      if (y < H && x < W) {
        // (only what is stored counts: the unit's max must not depend on how the launch tiles the map)
        if (amax) *amax = fmaxf(*amax, v != v ? __builtin_inff() : fabsf(v));
        if (write_main) gout[((size_t)(b * H + y) * W + x) * out_stride + cout] = v;
        m = v > m ? v : m;
      }
    }
    if (gpool && gy0 < H && gx < W) {
      const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
      gpool[((size_t)(b * Hp + (gy0 >> 1)) * Wp + (gx >> 1)) * pool_stride + cout] = m;
    }
  }
}

// LDS-transposed epilogue for a 16x16-pixel x BN-cout block tile.  An MFMA accumulator lane holds ONE
// cout for 16 pixels, so storing it directly is 4-byte stores, 128 B per wave-instruction, hundreds of
// them per thread (measured: ~50 k cycles per tile, a third of a short-K layer).  Instead the tile is
// parked in LDS as Cs[pixel][BN + 4] (the K-loop buffers are dead by then) and written out along the
// channel axis: float4 per lane, 512 contiguous bytes per pixel, and the fused 2x2 pool becomes a max
// over four float4 rows.  Needs 16-byte aligned channel views (the launcher checks and otherwise keeps
// the scalar conv_store_tile path).
constexpr int CS_PAD = 16;  // 2 x (BN + CS_PAD) words = 32 (mod 64 banks): the two half-waves of a staging ds_write_b32 (pixels x, x+2) never share a bank

// stage the 2x16-pixel x 32-cout MFMA tile of one lane: local rows ly0, ly0+1; `cl` = local cout
template <int BN, typename GetV>
__device__ __forceinline__ void conv_stage_tile(float* __restrict__ Cs, GetV getv, float bv, int relu_flags, int ly0,
                                                int kh, int cl) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lx = 2 * (2 * q + kh);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v = getv(4 * q + s) + bv;
      if (relu_flags & 1) v = fmaxf(v, 0.f);
      Cs[((ly0 + (s >> 1)) * 16 + lx + (s & 1)) * (BN + CS_PAD) + cl] = v;
    }
  }
}

// The same for the split-fp16 kernels' accumulator pair (out = main + corr * inv + bias), two pixels per
// instruction: registers r, r+1 of a lane are x-adjacent pixels, so v_pk_fma / v_pk_add / v_pk_max do the
// arithmetic and one ds_write2_b32 parks the pair (2.5 instead of 6.5 VALU ops per value).
typedef float cs_f32x2 __attribute__((ext_vector_type(2)));
typedef float cs_f32x16 __attribute__((ext_vector_type(16)));
// (ylim, xlim) = rows / columns of the block tile that are inside the image: only those count for `amax` -- the unit's
// max must not depend on how the launch tiles the map
// max(a, |v|) on the bit patterns (a >= 0): |NaN| > inf > every finite value, so a NaN output survives into the
// range flag (conv_raise_range_flag tests !(amax <= 65504)) instead of being dropped by fmaxf
__device__ __forceinline__ float conv_absmax_bits(float a, float v) {
  const unsigned ua = __builtin_bit_cast(unsigned, a), uv = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
  return __builtin_bit_cast(float, uv > ua ? uv : ua);
}

template <int BN, bool RELU>
__device__ __forceinline__ void conv_stage_tile_pk(float* __restrict__ Cs, const cs_f32x16 am, const cs_f32x16 ac,
                                                   float inv, float bv, int ly0, int kh, int cl, float& amax, int ylim,
                                                   int xlim) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lx = 2 * (2 * q + kh);
#pragma unroll
    for (int sp = 0; sp < 4; sp += 2) {
      const int r = 4 * q + sp;
      cs_f32x2 v = __builtin_elementwise_fma(cs_f32x2{ac[r], ac[r + 1]}, cs_f32x2{inv, inv}, cs_f32x2{am[r], am[r + 1]});
      v = v + cs_f32x2{bv, bv};
      if (RELU) v = __builtin_elementwise_max(v, cs_f32x2{0.f, 0.f});
      if (ly0 + (sp >> 1) < ylim) {  // fp16 range guard + activation exponent (v_max3 with abs modifiers)
        if (lx < xlim) amax = conv_absmax_bits(amax, v[0]);   // (on the bit patterns: a NaN outranks everything
        if (lx + 1 < xlim) amax = conv_absmax_bits(amax, v[1]);  // and raises the flag; fmaxf would drop it)
      }
      float* d = Cs + ((ly0 + (sp >> 1)) * 16 + lx) * (BN + CS_PAD) + cl;
      d[0] = v[0];
      d[BN + CS_PAD] = v[1];
    }
  }
}

// fp16 range guard of the split-fp16 kernels: `amax` = largest |output| this lane produced.  hi = fp16(x) of the
// consumer's split overflows above 65504; inf compares greater.  (fmaxf drops NaNs, but a NaN only ever appears
// downstream of an inf, which has raised the flag of the same pass already.)
__device__ __forceinline__ void conv_raise_range_flag(int* flag, float amax) {
  if (flag && !(amax <= 65504.0f)) atomicOr(flag, 1);
}

// ... and the unit's running max |output| for the consumer's activation exponent: one wave reduction, then at most one
// atomic per wave and slot (none once the slot has caught up: the max converges within the first blocks).  amax >= 0, so
// the float bit patterns order like unsigned integers; a pooled output is bounded by the un-pooled one (slot2).
__device__ __forceinline__ unsigned conv_wave_umax(unsigned v) {
  // four DPP steps make every 16-lane row uniform (quad xor 1, quad xor 2, half-row mirror, row mirror), four readlanes
  // and scalar maxes finish -- no lane id, no LDS crossbar (__shfl_xor needs the lane id in a vector register, and one
  // kept alive to the end of an epilogue is spilled and reloaded behind a wait for all the epilogue's stores)
  auto step = [](unsigned x, unsigned w) { return x > w ? x : w; };
  v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = step(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));   // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                 c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return step(step(a, b), step(c, d));
}
// The same in two halves for kernels whose epilogue ends in a burst of stores: PEEK the slots early (the load completes
// under the last MFMAs; pin the value with an empty asm before the first store so that no later `vmcnt` wait -- which on
// gfx9 also waits for every store issued since -- is generated for it), COMMIT after the epilogue: a stale peek only
// costs one atomic that changes nothing.
__device__ __forceinline__ unsigned conv_amax_peek(const unsigned* slot) {
  return slot ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
}
__device__ __forceinline__ void conv_amax_commit(unsigned* slot, unsigned seen, unsigned* slot2, unsigned seen2, float amax) {
  if (!slot && !slot2) return;
  const unsigned m = conv_wave_umax(__builtin_bit_cast(unsigned, amax));
  // (lane 0 by the hardware's lane id, formed here: threadIdx.x -- or a lane id -- kept alive to the end of an epilogue
  // is spilled, and reloaded behind a wait for every store of the epilogue)
  unsigned zero = 0u;
  asm volatile("" : "+v"(zero));   // (opaque, so that the lane id is formed HERE and not shared with an earlier one)
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero)) == 0) {
    if (slot && m > seen) atomicMax(slot, m);
    if (slot2 && m > seen2) atomicMax(slot2, m);
  }
}

__device__ __forceinline__ void conv_publish_amax(unsigned* slot, unsigned* slot2, float amax) {
  if (!slot && !slot2) return;
  const unsigned m = conv_wave_umax(__builtin_bit_cast(unsigned, amax));
  if ((threadIdx.x & 63) == 0) {
    if (slot && m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
    if (slot2 && m > __hip_atomic_load(slot2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot2, m);
  }
}

// exponent e >= 0 with amax * 2^e in [2^13, 2^14) (fp16 tops out at 65504): what a single-accumulator kernel multiplies
// its input by (exactly, in fp16) so that lo = fp16(x - hi) keeps its bits whatever the layer's magnitude -- unscaled,
// lo of |x| < 0.25 is an fp16 subnormal (quantum 2^-24) and a layer that lives around 1e-3 would keep 14 bits, not 22.
// 0 when the slot is unknown, zero, or not finite (the range flag is up in that case).  Capped at 15 -- the largest power of
// two fp16 holds, so the lift is ONE exact multiplication: a unit whose max is below 0.25 lands lower than [2^13, 2^14),
// at worst (max 2^-16 x a few) around 1, where the low parts still carry 2^-23 of the top (a bound 2^12 too large
// passes every magnitude test: tools/experiments/exponent_bias_run.sh).
__device__ __forceinline__ int conv_act_exponent_of_bits(unsigned b) {
  if (b == 0u || b >= 0x7f800000u) return 0;
  const int e = 13 - ((int)(b >> 23) - 127);
  return e < 0 ? 0 : (e > 15 ? 15 : e);
}
__device__ __forceinline__ int conv_act_exponent(const unsigned* slot) {
  if (!slot) return 0;
  return conv_act_exponent_of_bits(*slot);
}
// The same read through the SCALAR cache, in two halves: the slot pointer is wave-uniform and nothing in this launch
// writes the slot (its producers ran in earlier launches; the scalar cache starts a kernel empty).  As a vector load the
// read sat in vmcnt next to the kernels' first weight DMA: the compiler's wait for it drained those pieces too, and -- in
// front of the halo requests, once per tile -- put two or three serial memory round trips into every block's prologue.
// conv_act_slot_request early, conv_act_slot_bits (one lgkmcnt(0) wait for all requests) where the exponent is needed.
__device__ __forceinline__ unsigned conv_act_slot_request(const unsigned* slot) {
  unsigned b = 0u;   // (a null slot: "unknown", exponent 0)
  if (slot) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(b) : "s"(slot) : "memory");
  return b;
}
__device__ __forceinline__ void conv_act_slot_wait(unsigned& b0, unsigned& b1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1)::"memory");
}
// 2^k as an fp16 bit pattern, twice (v_pk_mul_f16 operand); k in [-14, 15]
__device__ __forceinline__ unsigned conv_pk_pow2_f16(int k) {
  const unsigned h = (unsigned)(k + 15) << 10;
  return h | (h << 16);
}

// write the staged tile (and its 2x2/2 max-pool) to global memory; all NT threads of the block.
// A block-wide store covers PPI = NT / (BN/4) pixels; the walk over the 256 pixels is fully unrolled
// with compile-time (row, column) steps so that an iteration is one LDS read, one predicate and one
// float4 store off a running pointer.
// The low halves of a split pair from the packed hi halves: lo = fp16((x - f32(hi)) * 2048) formed as
// fma(f32(hi), -2048, x * 2048) with ONE rounding to fp16 -- v_fma_mixlo / mixhi_f16 read the fp16 source straight out of the
// packed register, so the two conversions back to fp32, the subtraction and the second pack of the plain form become two
// instructions (3 instead of 5 per pair after the hi pack).  Bit for bit the plain form: x - hi is exact in fp32, so are both
// products, and the fma's exact result is rounded once like the conversion's (tools/scratch/mix_split.hip: 0 mismatches over
// 8.4 M pairs from 2^-30 to the fp16 ceiling, zeros and subnormal results included).
typedef _Float16 cs_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cs_h2 conv_split_lo(const cs_f32x2 x, const cs_h2 hi) {
  const cs_f32x2 x2 = x * 2048.0f;
  const float k = -2048.0f;
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(__builtin_bit_cast(unsigned, hi)), "s"(k), "v"(x2[0]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(__builtin_bit_cast(unsigned, hi)), "s"(k), "v"(x2[1]));
  return __builtin_bit_cast(cs_h2, d);
}

// four consecutive channels (c % 4 == 0) of one pixel in the split-fp16 activation format:
// [chunk c/32][hi 32 halfs | lo 32 halfs], x = hi + lo / 2048
__device__ __forceinline__ void conv_store_split4(float* __restrict__ pixel, int c, const float4 v) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
  const h2 h01 = __builtin_convertvector(x01, h2), h23 = __builtin_convertvector(x23, h2);
  const h2 l01 = conv_split_lo(x01, h01), l23 = conv_split_lo(x23, h23);
  unsigned char* d = (unsigned char*)(pixel + (c >> 5) * 32) + (c & 31) * 2;
  *(float2*)d = make_float2(__builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23));
  *(float2*)(d + 64) = make_float2(__builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23));
}

// eight consecutive channels (c % 8 == 0): the hi and the lo halves are ONE 16-byte store each (the epilogue is
// store-issue bound: half as many store instructions as two conv_store_split4)
__device__ __forceinline__ void conv_store_split8(float* __restrict__ pixel, int c, const float4 v0, const float4 v1) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
  float hi[4], lo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const h2 h = __builtin_convertvector(x[q], h2);
    const h2 l = conv_split_lo(x[q], h);
    hi[q] = __builtin_bit_cast(float, h);
    lo[q] = __builtin_bit_cast(float, l);
  }
  unsigned char* d = (unsigned char*)(pixel + (c >> 5) * 32) + (c & 31) * 2;
  *(float4*)d = make_float4(hi[0], hi[1], hi[2], hi[3]);
  *(float4*)(d + 64) = make_float4(lo[0], lo[1], lo[2], lo[3]);
}

// split-format outputs: 8 channels per thread, NT / (BN / 8) pixels per block-wide round
template <int BN, int NT, int ROWS>
__device__ __forceinline__ void conv_flush_tile_split8(const float* __restrict__ Cs, int tid, int ty0, int tx0, int H,
                                                       int W, int b, int cout0, float* __restrict__ gout,
                                                       int out_stride, float* __restrict__ gpool, int pool_stride,
                                                       bool write_main, bool pool_split) {
  constexpr int CG = BN / 8;
  constexpr int PPI = NT / CG;     // 16 (4 waves x BN 128), 32 or 64
  static_assert(PPI >= 16 && PPI % 16 == 0, "whole tile rows per store round");
  constexpr int YS = PPI / 16;     // tile rows per round
  const int cg = tid % CG, p0 = tid / CG;
  if (write_main) {
    const int y0 = ty0 + (p0 >> 4), x = tx0 + (p0 & 15);
    const float* cs = Cs + p0 * (BN + CS_PAD) + cg * 8;
    float* g = gout + ((size_t)(b * H + y0) * W + x) * out_stride;   // the pixel's first channel
    const size_t row_pitch = (size_t)W * out_stride;
#pragma unroll
    for (int r = 0; r < ROWS / YS; ++r) {
      if (y0 + r * YS < H && x < W) {
        const float* c = cs + r * YS * 16 * (BN + CS_PAD);
        conv_store_split8(g + r * YS * row_pitch, cout0 + cg * 8, *(const float4*)c, *(const float4*)(c + 4));
      }
    }
  }
  if (gpool) {
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
#pragma unroll
    for (int k = 0; k < (ROWS * 4 + PPI - 1) / PPI; ++k) {
      const int pp = p0 + k * PPI;
      const int ly = (pp >> 3) * 2, lx = (pp & 7) * 2;
      const int y = ty0 + ly, x = tx0 + lx;
      if (pp < ROWS * 4 && y < H && x < W) {
        const float* c0 = Cs + (ly * 16 + lx) * (BN + CS_PAD) + cg * 8;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -3.402823466e+38f;
        auto mx = [&](const float* q) {
          const float4 a = *(const float4*)q, bq = *(const float4*)(q + 4);
          const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
        };
        mx(c0);
        if (x + 1 < W) mx(c0 + (BN + CS_PAD));
        if (y + 1 < H) {
          mx(c0 + 16 * (BN + CS_PAD));
          if (x + 1 < W) mx(c0 + 17 * (BN + CS_PAD));
        }
        float* pq = gpool + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * pool_stride;
        if (pool_split) {
          conv_store_split8(pq, cout0 + cg * 8, make_float4(m[0], m[1], m[2], m[3]), make_float4(m[4], m[5], m[6], m[7]));
        } else {
          *(float4*)(pq + cout0 + cg * 8) = make_float4(m[0], m[1], m[2], m[3]);
          *(float4*)(pq + cout0 + cg * 8 + 4) = make_float4(m[4], m[5], m[6], m[7]);
        }
      }
    }
  }
}

template <int BN, int NT, int ROWS = 16>
__device__ __forceinline__ void conv_flush_tile(const float* __restrict__ Cs, int tid, int ty0, int tx0, int H, int W,
                                                int b, int cout0, float* __restrict__ gout, int out_stride,
                                                float* __restrict__ gpool, int pool_stride, bool write_main,
                                                bool main_split = false, bool pool_split = false) {
#ifndef SHF_FLUSH_SPLIT8
#define SHF_FLUSH_SPLIT8 1
#endif
  if constexpr (SHF_FLUSH_SPLIT8 && NT / (BN / 8) >= 16) {
    if ((write_main && main_split) || (!write_main && gpool && pool_split)) {
      conv_flush_tile_split8<BN, NT, ROWS>(Cs, tid, ty0, tx0, H, W, b, cout0, gout, out_stride, gpool, pool_stride,
                                           write_main, pool_split);
      return;
    }
  }
  constexpr int CG = BN / 4;       // float4 groups per pixel
  constexpr int PPI = NT / CG;     // pixels per block-wide store instruction: 8 (4 waves x BN 128), 16 or 32
  static_assert(PPI == 8 || PPI == 16 || PPI == 32, "tile walk assumes 8, 16 or 32 pixels per store round");
  const int cg = tid % CG, p0 = tid / CG;
  if (write_main) {
    constexpr int XS = PPI < 16 ? 16 / PPI : 1;   // store rounds per tile row
    constexpr int YS = PPI < 16 ? 1 : PPI / 16;   // tile rows per store round
    const int y0 = ty0 + (p0 >> 4), x0 = tx0 + (p0 & 15);
    const float* cs = Cs + p0 * (BN + CS_PAD) + cg * 4;
    float* g = gout + ((size_t)(b * H + y0) * W + x0) * out_stride + cout0 + cg * 4;
    const size_t row_pitch = (size_t)W * out_stride;
#pragma unroll
    for (int r = 0; r < ROWS / YS; ++r) {
#pragma unroll
      for (int c = 0; c < XS; ++c) {
        const int y = y0 + r * YS, x = x0 + c * PPI;
        if (y < H && x < W) {
          const float4 v = *(const float4*)(cs + (r * YS * 16 + c * PPI) * (BN + CS_PAD));
          float* gp = g + r * YS * row_pitch + (size_t)(c * PPI) * out_stride;
          if (main_split)
            conv_store_split4(gp - (cout0 + cg * 4), cout0 + cg * 4, v);
          else
            *(float4*)gp = v;
        }
      }
    }
  }
  if (gpool) {
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
#pragma unroll
    for (int k = 0; k < (ROWS * 4 + PPI - 1) / PPI; ++k) {
      const int pp = p0 + k * PPI;
      const int ly = (pp >> 3) * 2, lx = (pp & 7) * 2;
      const int y = ty0 + ly, x = tx0 + lx;
      if (pp < ROWS * 4 && y < H && x < W) {
        // windows on a ragged edge are clipped like Caffe's (pooling_layer.cu:24-27)
        const float* c0 = Cs + (ly * 16 + lx) * (BN + CS_PAD) + cg * 4;
        float4 m = make_float4(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
        auto mx = [&](const float* q) {
          const float4 v = *(const float4*)q;
          m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
          m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
        };
        mx(c0);
        if (x + 1 < W) mx(c0 + (BN + CS_PAD));
        if (y + 1 < H) {
          mx(c0 + 16 * (BN + CS_PAD));
          if (x + 1 < W) mx(c0 + 17 * (BN + CS_PAD));
        }
        float* pp = gpool + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * pool_stride;
        if (pool_split)
          conv_store_split4(pp, cout0 + cg * 4, m);
        else
          *(float4*)(pp + cout0 + cg * 4) = m;
      }
    }
  }
}

// The persistent 4-wave kernel's epilogue unit: one QUARTER of the 16x16 tile -- tile rows {2q, 2q+1, 8+2q, 8+2q+1},
// staged as Cs rows 0..3 (64 pixels x BN couts) -- written by 256 threads, 8 channels each: a block-wide round is one
// tile row (16 pixels x 128 couts), four rounds; the 16 pooling windows of the quarter are one per 16 threads.
template <int BN>
__device__ __forceinline__ void conv_flush_quarter(const float* __restrict__ Cs, int tid, int q, int ty0, int tx0, int H,
                                                   int W, int b, int cout0, float* __restrict__ gout, int out_stride,
                                                   float* __restrict__ gpool, int pool_stride, bool write_main,
                                                   bool main_split, bool pool_split) {
  constexpr int CG = BN / 8;
  static_assert(CG == 16, "256 threads = 16 pixels x 16 channel groups");
  const int cg = tid % CG, p0 = tid / CG;
  if (write_main) {
    const int x = tx0 + p0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty0 + (r >> 1) * 8 + 2 * q + (r & 1);
      if (y < H && x < W) {
        const float* c = Cs + (r * 16 + p0) * (BN + CS_PAD) + cg * 8;
        float* g = gout + ((size_t)(b * H + y) * W + x) * out_stride;
        const float4 v0 = *(const float4*)c, v1 = *(const float4*)(c + 4);
        if (main_split) {
          conv_store_split8(g, cout0 + cg * 8, v0, v1);
        } else {
          *(float4*)(g + cout0 + cg * 8) = v0;
          *(float4*)(g + cout0 + cg * 8 + 4) = v1;
        }
      }
    }
  }
  if (gpool) {
    const int Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const int pair = p0 >> 3, lx = (p0 & 7) * 2;
    const int y = ty0 + pair * 8 + 2 * q, x = tx0 + lx;
    if (y < H && x < W) {
      // windows on a ragged edge are clipped like Caffe's (pooling_layer.cu:24-27)
      const float* c0 = Cs + (pair * 2 * 16 + lx) * (BN + CS_PAD) + cg * 8;
      float m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = -3.402823466e+38f;
      auto mx = [&](const float* qq) {
        const float4 a = *(const float4*)qq, bq = *(const float4*)(qq + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
      };
      mx(c0);
      if (x + 1 < W) mx(c0 + (BN + CS_PAD));
      if (y + 1 < H) {
        mx(c0 + 16 * (BN + CS_PAD));
        if (x + 1 < W) mx(c0 + 17 * (BN + CS_PAD));
      }
      float* pq = gpool + ((size_t)(b * Hp + (y >> 1)) * Wp + (x >> 1)) * pool_stride;
      if (pool_split) {
        conv_store_split8(pq, cout0 + cg * 8, make_float4(m[0], m[1], m[2], m[3]), make_float4(m[4], m[5], m[6], m[7]));
      } else {
        *(float4*)(pq + cout0 + cg * 8) = make_float4(m[0], m[1], m[2], m[3]);
        *(float4*)(pq + cout0 + cg * 8 + 4) = make_float4(m[4], m[5], m[6], m[7]);
      }
    }
  }
}

// REGISTER epilogue for kernels that run the MFMA with the WEIGHTS as the A operand, i.e. D[cout][pixel]: a lane then
// owns ONE pixel (column lane & 31 -> row_to_pixel) and 16 couts of the 32-cout tile, rows (r & 3) + 8 (r >> 2) + 4 kh.
// Four v_permlane32_swap pairs exchange register quads between the two half-waves (same pixel, kh = 0 / 1) so that
// kh = 0 ends up with couts 0..15 and kh = 1 with couts 16..31, in register order 0-3, 8-11, 4-7, 12-15 -- 64
// contiguous bytes of the pixel (fp32 output) or 32 B of hi + 32 B of lo (split-fp16 output): four 16-byte stores per
// accumulator tile, no LDS round trip and no barrier.  The fused 2x2 max-pool is a max over the four lanes of a quad
// (row_to_pixel walks a 2x2 window with the two low lane bits): two DPP quad permutes; windows on a ragged edge are
// clipped like Caffe's (pooling_layer.cu:24-27) by feeding -FLT_MAX for pixels outside the image.
// the half-wave exchange on its own (used by the fused first pair for conv1_1's way into LDS): afterwards kh = 0 holds
// rows 0..15 and kh = 1 rows 16..31 of the 32-row tile, in register order 0-3, 8-11, 4-7, 12-15
__device__ __forceinline__ void conv_swap_halves(float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int x = (s & 3) + 4 * (s >> 2), y = x + 8;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[x]), "+v"(v[y]));
  }
}

template <bool RELU>
__device__ __forceinline__ void conv_epilogue_regs_tail(float (&v)[16], const float4* bias16, bool valid, bool interior,
                                                        float* __restrict__ pix_main, int cout16, bool main_split,
                                                        float* __restrict__ pix_pool, bool pool_writer, bool pool_split,
                                                        float& amax) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int x = (s & 3) + 4 * (s >> 2), y = x + 8;   // (0..3 <-> 8..11), (4..7 <-> 12..15)
    // (inline asm: hipcc 7.2 drops the SECOND result of __builtin_amdgcn_permlane32_swap -- both outputs come back as
    // the new vdst; the s_nop covers the VALU-write -> permlane-read wait states the compiler would have inserted)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[x]), "+v"(v[y]));
  }
  constexpr int ORD[4] = {0, 8, 4, 12};
  float4 o[4];
  // max |output| of the tile on the BIT PATTERNS (sign cleared, they order like the magnitudes): one and / max3 per two
  // values instead of fmaxf's NaN-quieting float sequence -- and a NaN, which fmaxf would drop, reaches the range flag
  unsigned tmax = 0u;
  auto umax3 = [](unsigned t, float x, float y) {
    const unsigned p = __builtin_bit_cast(unsigned, x) & 0x7fffffffu, q = __builtin_bit_cast(unsigned, y) & 0x7fffffffu;
    const unsigned m = p > q ? p : q;
    return t > m ? t : m;           // v_max3_u32
  };
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cs_f32x2 a = cs_f32x2{v[ORD[g]], v[ORD[g] + 1]} + cs_f32x2{bias16[g].x, bias16[g].y};
    cs_f32x2 b = cs_f32x2{v[ORD[g] + 2], v[ORD[g] + 3]} + cs_f32x2{bias16[g].z, bias16[g].w};
    o[g] = make_float4(a[0], a[1], b[0], b[1]);
    if (RELU) {
      // max(x, 0) on the bit patterns (signed): one v_max_i32 per value, no NaN-quieting pre-pass; -0 and negatives -> +0
      auto relu1 = [](float x) { const int q = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, q > 0 ? q : 0); };
      o[g] = make_float4(relu1(o[g].x), relu1(o[g].y), relu1(o[g].z), relu1(o[g].w));
    }
    tmax = umax3(umax3(tmax, o[g].x, o[g].y), o[g].z, o[g].w);
  }
  {
    // (only pixels inside the image count: the unit's max must not depend on how the launch tiles the map)
    const unsigned am = __builtin_bit_cast(unsigned, amax);
    amax = __builtin_bit_cast(float, valid && tmax > am ? tmax : am);
  }
  if (pix_main && valid) {
    if (main_split) {
      conv_store_split8(pix_main, cout16, o[0], o[1]);
      conv_store_split8(pix_main, cout16 + 8, o[2], o[3]);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) *(float4*)(pix_main + cout16 + 4 * g) = o[g];
    }
  }
  if (pix_pool) {
    float4 m[4];
    if (RELU) {
      // values are >= 0: their bit patterns order like unsigned integers, 0 is the neutral element for pixels outside
      // the image, and the quad max is two v_max_u32 with a DPP operand each
      auto quad_max = [&](float x) {
        unsigned u = __builtin_bit_cast(unsigned, x);
        if (!interior) u = valid ? u : 0u;
        unsigned w = (unsigned)__builtin_amdgcn_mov_dpp((int)u, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
        u = u > w ? u : w;
        w = (unsigned)__builtin_amdgcn_mov_dpp((int)u, 0x4E, 0xf, 0xf, true);           // quad_perm [2,3,0,1]
        u = u > w ? u : w;
        return __builtin_bit_cast(float, u);
      };
#pragma unroll
      for (int g = 0; g < 4; ++g) m[g] = make_float4(quad_max(o[g].x), quad_max(o[g].y), quad_max(o[g].z), quad_max(o[g].w));
    } else {
      auto quad_max = [&](float x) {
        x = valid ? x : -3.402823466e+38f;
        int xi = __builtin_bit_cast(int, x);
        x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0xB1, 0xf, 0xf, false)));
        xi = __builtin_bit_cast(int, x);
        return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x4E, 0xf, 0xf, false)));
      };
#pragma unroll
      for (int g = 0; g < 4; ++g) m[g] = make_float4(quad_max(o[g].x), quad_max(o[g].y), quad_max(o[g].z), quad_max(o[g].w));
    }
    if (pool_writer) {
      if (pool_split) {
        conv_store_split8(pix_pool, cout16, m[0], m[1]);
        conv_store_split8(pix_pool, cout16 + 8, m[2], m[3]);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(pix_pool + cout16 + 4 * g) = m[g];
      }
    }
  }
}

// POOL-ONLY register epilogue (round 4): the layer's un-pooled map is not stored (conv1_2, conv2_2, conv3_3 of VGG-16 feed
// their 2x2/2 MAX pool and nothing else).  The generic tail above forms all 16 pooled values in every lane of a pooling quad
// and lets one lane split and store them: the split of a pooled value was done four times over, after a half-wave exchange
// nothing here needs.  Here: bias, then the quad max on the raw accumulator order -- registers 4q..4q+3 of a lane are the
// couts 8q + 4kh .. + 3 of the 32-cout tile, the same in all four lanes of a quad --, then lane ql of the quad keeps register
// quad q = ql (three selects per value), applies ReLU to those FOUR values, splits them and stores 8 B of hi + 8 B of lo:
// the eight lanes (4 of the quad x 2 half-waves) of a pooled pixel write its 32 couts.  ~80 vector instructions per
// accumulator tile instead of ~175, and the SAME bits: max commutes with the ReLU (max over the quad and 0, as signed
// integers on the bit patterns: any positive value beats every negative one and 0 beats them all), the bias is added before
// the max like before, and the unit's max |output| over the pooled values equals that over all in-image outputs (every
// in-image pixel lies in exactly one clipped window).  getv(r): register r's value before the bias; getb(q): the biases of couts 8q + 4kh .. + 3 (from LDS, or preloaded);
// `window` = the quad's top-left pixel is inside the image (the pooled pixel exists); `valid` = this lane's own pixel is.
template <bool RELU, typename GetV, typename GetB>
__device__ __forceinline__ void conv_epilogue_pool_only(GetV getv, GetB getb, bool valid, bool window,
                                                        bool interior, float* __restrict__ pix_pool, int cout32, int kh,
                                                        int ql, bool pool_split, float& amax) {
  auto quad_max = [&](float xf) {
    if (RELU) {
      int x = __builtin_bit_cast(int, xf);
      if (!interior) x = valid ? x : 0;        // (a pixel outside the image: 0 is neutral under the ReLU that follows)
      int y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
      x = x > y ? x : y;
      y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
      return __builtin_bit_cast(float, x > y ? x : y);
    } else {
      float x = valid ? xf : -3.402823466e+38f;
      int xi = __builtin_bit_cast(int, x);
      x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0xB1, 0xf, 0xf, false)));
      xi = __builtin_bit_cast(int, x);
      return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x4E, 0xf, 0xf, false)));
    }
  };
  // a register quad at a time (four values + the four kept so far are live, not sixteen): bias from LDS, the quad max by
  // ALL lanes (a DPP operand must not be read inside a lane-dependent branch), then lane ql keeps quad q = ql
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bq = getb(q);   // biases of couts 8q + 4kh .. + 3
    const cs_f32x2 a = cs_f32x2{getv(4 * q), getv(4 * q + 1)} + cs_f32x2{bq.x, bq.y};
    const cs_f32x2 b = cs_f32x2{getv(4 * q + 2), getv(4 * q + 3)} + cs_f32x2{bq.z, bq.w};
    const float m0 = quad_max(a[0]), m1 = quad_max(a[1]), m2 = quad_max(b[0]), m3 = quad_max(b[1]);
    const bool mine = ql == q;
    w0 = mine ? m0 : w0;
    w1 = mine ? m1 : w1;
    w2 = mine ? m2 : w2;
    w3 = mine ? m3 : w3;
  }
  if (RELU) {   // on the bit patterns: negatives and -0 -> +0
    auto relu1 = [](float x) { const int q = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, q > 0 ? q : 0); };
    w0 = relu1(w0); w1 = relu1(w1); w2 = relu1(w2); w3 = relu1(w3);
  }
  {
    const unsigned q0 = __builtin_bit_cast(unsigned, w0) & 0x7fffffffu, q1 = __builtin_bit_cast(unsigned, w1) & 0x7fffffffu;
    const unsigned q2 = __builtin_bit_cast(unsigned, w2) & 0x7fffffffu, q3 = __builtin_bit_cast(unsigned, w3) & 0x7fffffffu;
    const unsigned t01 = q0 > q1 ? q0 : q1, t23 = q2 > q3 ? q2 : q3, t = t01 > t23 ? t01 : t23;
    const unsigned am = __builtin_bit_cast(unsigned, amax);
    amax = __builtin_bit_cast(float, window && t > am ? t : am);
  }
  if (window) {
    const int c = cout32 + 8 * ql + 4 * kh;
    if (pool_split)
      conv_store_split4(pix_pool, c, make_float4(w0, w1, w2, w3));
    else
      *(float4*)(pix_pool + c) = make_float4(w0, w1, w2, w3);
  }
}

template <bool RELU>
__device__ __forceinline__ void conv_epilogue_regs(const cs_f32x16 am, const cs_f32x16 ac, float inv, const float4* bias16,
                                                   bool valid, bool interior, float* __restrict__ pix_main, int cout16,
                                                   bool main_split, float* __restrict__ pix_pool, bool pool_writer,
                                                   bool pool_split, float& amax) {
  // `interior` (wave-uniform): every pixel of the block tile is inside the image -- no pooling window needs clipping
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const cs_f32x2 t = __builtin_elementwise_fma(cs_f32x2{ac[r], ac[r + 1]}, cs_f32x2{inv, inv}, cs_f32x2{am[r], am[r + 1]});
    v[r] = t[0];
    v[r + 1] = t[1];
  }
  conv_epilogue_regs_tail<RELU>(v, bias16, valid, interior, pix_main, cout16, main_split, pix_pool, pool_writer, pool_split, amax);
}

// one accumulator (dual-tile kernel: unscaled low parts, weights pre-scaled by a power of two): out = acc * scale
template <bool RELU>
__device__ __forceinline__ void conv_epilogue_regs1(const cs_f32x16 acc, float scale, const float4* bias16, bool valid,
                                                    bool interior, float* __restrict__ pix_main, int cout16, bool main_split,
                                                    float* __restrict__ pix_pool, bool pool_writer, bool pool_split,
                                                    float& amax) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const cs_f32x2 t = cs_f32x2{acc[r], acc[r + 1]} * cs_f32x2{scale, scale};
    v[r] = t[0];
    v[r + 1] = t[1];
  }
  conv_epilogue_regs_tail<RELU>(v, bias16, valid, interior, pix_main, cout16, main_split, pix_pool, pool_writer, pool_split, amax);
}

}  // namespace shf
