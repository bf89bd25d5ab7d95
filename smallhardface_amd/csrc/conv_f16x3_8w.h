// Split-fp16 convolution, 8-wave kernel: Cin 64 with Cout 64, the dilated heads, the 1x1 convolutions, unaligned views.
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

#include "conv_f16x3_types.h"

namespace shf {

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


}  // namespace shf
