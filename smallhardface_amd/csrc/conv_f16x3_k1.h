// Split-fp16 1x1 convolutions (conv4_256 / conv5_256 of the detector: 512 -> 256 + ReLU) as a plain GEMM.
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

#include "conv_f16x3_types.h"

namespace shf {

// A 1x1 convolution has no halo and no taps to reuse an input tile over: per MFMA it moves NINE times the activation
// bytes of a 3x3 layer, and on the 8-wave kernel's KS = 1 form -- fp32 input, split and parked in LDS chunk by chunk, two
// barriers per 24 MFMAs of a wave -- it ran at 0.12-0.22 matrix-pipe busy (round 4: 162 + 81 us per image).  Here:
//  * a block is 256 PIXELS of the member's flat pixel list (no 2-D tile: nothing ragged but the last block) x ALL 256
//    couts, so an activation byte is fetched once; a wave owns 64 pixels x 256 couts = 16 accumulator tiles; the MFMA
//    runs D[cout][pixel] like the dual-tile family and the family's register epilogue is reused as it is;
//  * a 32-channel chunk of a pixel is ONE 128-byte line in either input format (split: hi 64 B | lo 64 B; fp32: 32
//    floats).  Eight lanes of an LDS-DMA instruction gather one pixel's line -- full lines, no registers, no vector
//    instructions -- into [pixel][128 B] rows of LDS, each wave its own 64 pixels, THREE chunk buffers deep (a fetch has
//    one and a half to two chunks = 2-3 us to land).  The 16-byte piece a lane fetches is chosen so that slot j of pixel
//    q's row holds piece j ^ ((q >> 1) & 7): the sixteen lanes of a ds_read_b128 lane group -- sixteen pixels, one piece
//    index -- then cover all 64 banks.  (Measured on the way, conv5_256 + conv4_256 per image: fragments loaded straight
//    into registers, 16 B per lane from 32 lines per instruction, one chunk ahead: 60 + 140 us against 43 + 97 with the
//    loads removed; full lines by plain loads, parked with ds_write_b128 a chunk later: 60 + 135; this form 52 + 114,
//    and no longer sensitive to the fetches: what was left was the EPILOGUE, below.);
//  * the weights come from the family's pack (pack_conv_weights_split16h with k = 1: [16-channel slab][cout][64 B, pieces
//    rotated by row / 4]) by LDS DMA, 32 KiB per chunk, double-buffered: ONE barrier per 96 MFMAs of a wave; the sixteen
//    DMA issues of a wave and chunk (8 weight pieces, then 8 pixel groups) are spread over the chunk's eight MFMA groups;
//  * single-accumulator arithmetic with the activation exponent, exactly the family's: hi * 2^e, the format's lo (which
//    carries 2^11) * 2^(e - 11), weights pre-scaled by the pack's power of two, products hi*hi, lo*hi, hi*lo in that order.
//  * the output of a block is 256 KiB -- per MFMA nine times a 3x3 layer's -- and the family's register epilogue writes it
//    as 16-byte pieces of 64 different lines per instruction: 31-75 k cycles per block against 81 k for the whole K loop
//    (in-kernel cycle counters, -DSHF_K1_TIMING).  Here the tile goes through the (now idle) LDS: a wave parks its 32
//    pixels x 256 couts in rows of 1 KiB + 16 B (the eight lanes of a ds_write_b128 group are eight pixels: 16 B apart
//    mod 256) and writes each pixel's 1 KiB with ONE instruction of 64 consecutive lanes, bias and ReLU applied on the way
//    out (a lane's four couts are the same in every row: one bias quad per lane, no loads between the tiles): ~20 k cycles,
//    what is left is the write burst of every block storing at once.  (A split-format output keeps the register form.)
//    Per block (cycles): prologue 8-11 k, K loop 16 x 3.85 k (MFMA-only: 3.07 k), epilogue 20 k; per image 45 + 98 us.
// LDS: 3 x 32 KiB of activations + 2 x 32 KiB of weights = the 160 KiB of a CU; the biases are read from global memory in the
// epilogue.
template <bool IN_SPLIT, int NP = 3>
__global__ __launch_bounds__(256) void conv_mfma_f16x3_k1_kernel(ConvK p) {
  constexpr int BN = 256, PXB = 256, NTN = BN / 32, WROWB = 64;
  constexpr int SLAB_B = BN * WROWB;                 // 16 KiB: one 16-channel slab of the block's couts
  constexpr int BUF_B = 2 * SLAB_B;                  // a 32-channel chunk of weights
  constexpr int ABUF_B = PXB * 128;                  // a 32-channel chunk of the block's pixels
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                          // [3 buffers][256 pixels][128 B]
  unsigned char* Bs = smem + 3 * ABUF_B;             // [2 buffers][2 slabs][BN][64 B]

#ifdef SHF_K1_TIMING
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int ct = bid % p.nct;
  int pt = bid / p.nct;
  const int nchunks = p.Cin / 32;
  const unsigned char* wbase = (const unsigned char*)p.wph + (size_t)ct * BN * WROWB;
  const size_t slab_b = (size_t)p.Cout * WROWB;      // bytes per 16-channel slab of the whole layer
  const unsigned lane16 = (unsigned)lane * 16u;
  // (inline asm for the reasons given at the family's dma_w: the compiler would drain vmcnt before unrelated LDS accesses,
  // and M0 has to be written in the statement that uses it)
  auto dma = [&](unsigned lds, unsigned voff, const unsigned char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase));
  };
  // weight DMA: a chunk is 32 one-KiB pieces, piece q = wave + 4 r (slab q / 16, KiB q % 16 of the block's rows), r = 0..7
  auto dma_w = [&](int c32, int buf, int r) {
    const int q = wave_u + 4 * r;
    const unsigned char* ub = wbase + (size_t)(2 * c32 + (q >> 4)) * slab_b + (size_t)(q & 15) * 1024;
    dma((unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(Bs + buf * BUF_B + q * 1024), lane16, ub);
  };
#pragma unroll
  for (int r = 0; r < 8; ++r) dma_w(0, 0, r);

  const int mi = conv_find_member(p, pt);
  const ConvMember mem = p.m[mi];
  pt -= mem.tile_start;
  const int npix = mem.B * mem.H * mem.W;
  const int i = lane & 31, kh = lane >> 5;
  const int pixw = pt * PXB + wave_u * 64;           // the wave's first pixel
  // activation DMA: group r (0..7) of a wave = its pixels 8 r .. 8 r + 7, lane -> pixel 8 r + lane / 8, LDS slot lane % 8 of
  // the pixel's row; a pixel past the member's end fetches the last one's bytes (and stores nothing)
  unsigned a_voff[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int ql = r * 8 + (lane >> 3);              // pixel within the wave's 64
    const int P = pixw + ql;
    a_voff[r] = (unsigned)(P < npix ? P : npix - 1) * (unsigned)p.in_stride * 4u + (unsigned)(((lane & 7) ^ ((ql >> 1) & 7)) * 16);
  }
  // (before the activation requests: the compiler waits for this load by ITS count of what is in flight)
  const unsigned seen = conv_amax_peek(mem.out_amax);   // (early, so possibly stale: costs an atomic, never a result)
  const unsigned char* gin = (const unsigned char*)mem.in;
  auto dma_a = [&](int c32, int buf, int r) {
    dma((unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(As + buf * ABUF_B + (wave_u * 64 + r * 8) * 128),
        a_voff[r], gin + (size_t)c32 * 128);
  };
#pragma unroll
  for (int r = 0; r < 8; ++r) dma_a(0, 0, r);
#pragma unroll
  for (int r = 0; r < 8; ++r) dma_a(nchunks > 1 ? 1 : 0, 1, r);
  // activation exponent (conv_common.h): the unit's max |input| -> hi * 2^e, lo * 2^(e - 11), exact in fp16
  unsigned slot_bits = conv_act_slot_request(mem.in_amax), slot_none = 0u;   // (through the scalar cache: conv_common.h)
  conv_act_slot_wait(slot_bits, slot_none);
  const int e_act = conv_act_exponent_of_bits(slot_bits);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 f_hi = __builtin_bit_cast(h2, conv_pk_pow2_f16(e_act)), f_lo = __builtin_bit_cast(h2, conv_pk_pow2_f16(e_act - 11));
  // the two 16-byte pieces a lane reads per (pixel tile, k-step) -> its hi and lo fragments
  auto frag = [&](const float4 r0, const float4 r1, half8& hi, half8& lo) {
    if constexpr (IN_SPLIT) {
      float4 h = r0, l = r1;
      float* hp = &h.x;
      float* lp = &l.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hp[k] = __builtin_bit_cast(float, __builtin_bit_cast(h2, hp[k]) * f_hi);
        lp[k] = __builtin_bit_cast(float, __builtin_bit_cast(h2, lp[k]) * f_lo);
      }
      hi = __builtin_bit_cast(half8, h);
      lo = __builtin_bit_cast(half8, l);
    } else {
      // (lo through the split activation format's 2^11, like a producer's epilogue + the scaling above would: the two input
      // formats give the same bits)
      float hw[4], lw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = (k >> 1) ? r1 : r0;
        const f32x2 x = (k & 1) ? f32x2{v.z, v.w} : f32x2{v.x, v.y};
        const h2 hh = __builtin_convertvector(x, h2);
        const h2 ll = conv_split_lo(x, hh);
        hw[k] = __builtin_bit_cast(float, hh * f_hi);
        lw[k] = __builtin_bit_cast(float, ll * f_lo);
      }
      hi = __builtin_bit_cast(half8, make_float4(hw[0], hw[1], hw[2], hw[3]));
      lo = __builtin_bit_cast(half8, make_float4(lw[0], lw[1], lw[2], lw[3]));
    }
  };
  // LDS offsets of the lane's pieces: pixel row (wave's 64 + 32 tm + i) x 128 B, slot = piece ^ ((i >> 1) & 7)
  //   split input: hi piece 2 s + kh, lo piece 4 + 2 s + kh;   fp32 input: pieces 4 s + 2 kh and + 1 (eight floats)
  const int swz = (i >> 1) & 7;
  int a_off[2][2];                                    // [k-step][first | second piece], pixel tile 1 = + 32 rows
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int p0 = IN_SPLIT ? 2 * s + kh : 4 * s + 2 * kh, p1 = IN_SPLIT ? p0 + 4 : p0 + 1;
    a_off[s][0] = (wave_u * 64 + i) * 128 + ((p0 ^ swz) * 16);
    a_off[s][1] = (wave_u * 64 + i) * 128 + ((p1 ^ swz) * 16);
  }
  // weight fragment of cout row `row` (hi piece; the lo piece sits two rotated positions further: ^ 32)
  int b_off[NTN];
#pragma unroll
  for (int tn = 0; tn < NTN; ++tn) {
    const int row = tn * 32 + i;
    b_off[tn] = row * WROWB + ((kh + (row >> 2)) & 3) * 16;
  }

  f32x16 acc[2][NTN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < NTN; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

#ifdef SHF_K1_TIMING
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  int abuf = 0, abuf2 = 2;                            // activation buffers of chunk c / chunk c + 2 (c mod 3, (c + 2) mod 3)
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    // in flight, oldest first: ..., A(c), [A(c + 1) when c = 0], W(c), A(c + 1).  All but the last eight = W(c) and A(c) and
    // everything older have landed; the barrier publishes the weights (the activations are the wave's own)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    const int cw = c + 1 < nchunks ? c + 1 : c;      // past the end: re-fetch a valid chunk (unused) instead of branching
    const int ca = c + 2 < nchunks ? c + 2 : nchunks - 1;
    const unsigned char* Ab = As + abuf * ABUF_B;
    const unsigned char* Bst = Bs + (c & 1) * BUF_B;
    float4 ra[2][2][2];                               // [k-step][pixel tile][piece]
    half8 bfr[2][2][2];                               // [group parity][tile of the pair][hi | lo]
    auto read_a = [&](int s) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        ra[s][tm][0] = *(const float4*)(Ab + tm * 32 * 128 + a_off[s][0]);
        ra[s][tm][1] = *(const float4*)(Ab + tm * 32 * 128 + a_off[s][1]);
      }
    };
    auto read_b = [&](int g) {
      const unsigned char* Bp = Bst + (g >> 2) * SLAB_B;
      const int tp = (g & 3) * 2;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bfr[g & 1][t][0] = *(const half8*)(Bp + b_off[tp + t]);
        bfr[g & 1][t][1] = *(const half8*)(Bp + (b_off[tp + t] ^ 32));
      }
    };
    read_a(0);
    read_b(0);
    half8 ah[2], al[2];
    // eight groups of twelve MFMAs: k-step g / 4, cout tiles 2 (g % 4) and + 1
    auto group = [&](auto G_) {
      constexpr int g = decltype(G_)::value;
      constexpr int s = g >> 2, tp = (g & 3) * 2;
      if ((g & 3) == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) frag(ra[s][tm][0], ra[s][tm][1], ah[tm], al[tm]);
      }
      if (g + 1 < 8) read_b(g + 1);
      if (g == 2) read_a(1);
      // this group's two DMA issues: the next chunk's weights first (groups 0-3), then the activations of chunk c + 2
#pragma unroll
      for (int r = 2 * g; r < 2 * g + 2; ++r) {
        if (r < 8) dma_w(cw, (c + 1) & 1, r);
        else dma_a(ca, abuf2, r - 8);
      }
      // consecutive MFMAs never chain on one accumulator: the four tiles of the pair, product by product
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm][tp + t] = mma16<false>(bfr[g & 1][t][0], ah[tm], acc[tm][tp + t]);
      if constexpr (NP >= 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tm][tp + t] = mma16<false>(bfr[g & 1][t][1], ah[tm], acc[tm][tp + t]);
      }
      if constexpr (NP >= 3) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tm][tp + t] = mma16<false>(bfr[g & 1][t][0], al[tm], acc[tm][tp + t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    using std::integral_constant;
    group(integral_constant<int, 0>{}); group(integral_constant<int, 1>{}); group(integral_constant<int, 2>{});
    group(integral_constant<int, 3>{}); group(integral_constant<int, 4>{}); group(integral_constant<int, 5>{});
    group(integral_constant<int, 6>{}); group(integral_constant<int, 7>{});
    abuf = abuf == 2 ? 0 : abuf + 1;
    abuf2 = abuf2 == 2 ? 0 : abuf2 + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last chunks' (unused) re-fetches
#ifdef SHF_K1_TIMING
  const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
  unsigned long long te[4] = {0, 0, 0, 0}, tb = 0;
#endif

  {
    const bool relu = (p.relu & 1) != 0, main_split = (p.relu & 32) != 0;
    const float out_scale = p.wscale_inv * __builtin_bit_cast(float, (unsigned)(127 - e_act) << 23);   // 2^-e, exact
    float amax = 0.f;
    if (main_split) {
      // split-format output: the family's register epilogue (a lane ends up with 16 consecutive couts of its pixel per tile)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int P = pixw + tm * 32 + i;
        const bool valid = P < npix;
        float* pm = mem.out + (size_t)(unsigned)(valid ? P : 0) * (unsigned)p.out_stride;
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) {
          const int cout16 = ct * BN + tn * 32 + kh * 16;
          float4 bias16[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) bias16[g] = p.bias ? *(const float4*)(p.bias + cout16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (relu)
            conv_epilogue_regs1<true>(acc[tm][tn], out_scale, bias16, valid, false, pm, cout16, true, nullptr, false, false, amax);
          else
            conv_epilogue_regs1<false>(acc[tm][tn], out_scale, bias16, valid, false, pm, cout16, true, nullptr, false, false, amax);
        }
      }
    } else {
      // fp32 output through LDS: the wave's 32 pixels x 256 couts of one pixel tile in rows of 1 KiB + 16 B -- the scaled
      // accumulators as they lie in the registers, four 16-byte pieces per tile (couts 8 q + 4 kh .. + 3: no half-wave
      // exchange needed) --, then per pixel ONE 1-KiB store of 64 consecutive lanes; a lane's four couts are the same in every
      // row, so bias, ReLU (on the bit patterns) and the max |output| happen there, with one bias quad per lane.  Same
      // operations on the same values as the register form.
      constexpr int EROW = BN * 4 + 16;
      const float4 bq = p.bias ? *(const float4*)(p.bias + ct * BN + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();                                // every wave is done with the K loop's buffers
#ifdef SHF_K1_TIMING
      tb = __builtin_amdgcn_s_memtime();
#endif
      unsigned char* Ew = smem + wave_u * (32 * EROW);
      unsigned tmax = 0u;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 a = f32x2{acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1]} * f32x2{out_scale, out_scale};
            const f32x2 b = f32x2{acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]} * f32x2{out_scale, out_scale};
            *(float4*)(Ew + i * EROW + (tn * 32 + 8 * q + 4 * kh) * 4) = make_float4(a[0], a[1], b[0], b[1]);
          }
        }
#ifdef SHF_K1_TIMING
        te[2 * tm] = __builtin_amdgcn_s_memtime();
#endif
        // the wave's own rows: LDS operations of a wave complete in order, no barrier
        const int prow0 = pixw + tm * 32;
        float* gout = mem.out + (size_t)(unsigned)prow0 * (unsigned)p.out_stride + ct * BN + lane * 4;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          const float4 w = *(const float4*)(Ew + r * EROW + lane * 16);
          const f32x2 a = f32x2{w.x, w.y} + f32x2{bq.x, bq.y};
          const f32x2 b = f32x2{w.z, w.w} + f32x2{bq.z, bq.w};
          float4 o = make_float4(a[0], a[1], b[0], b[1]);
          if (relu) {
            auto relu1 = [](float x) { const int q = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, q > 0 ? q : 0); };
            o = make_float4(relu1(o.x), relu1(o.y), relu1(o.z), relu1(o.w));
          }
          if (prow0 + r < npix) {                     // (wave-uniform; only pixels inside the member are stored and counted)
            auto um = [](unsigned t, float x) { const unsigned q = __builtin_bit_cast(unsigned, x) & 0x7fffffffu; return t > q ? t : q; };
            tmax = um(um(um(um(tmax, o.x), o.y), o.z), o.w);
            *(float4*)(gout + (size_t)r * (unsigned)p.out_stride) = o;
          }
        }
#ifdef SHF_K1_TIMING
        te[2 * tm + 1] = __builtin_amdgcn_s_memtime();
#endif
      }
      amax = __builtin_bit_cast(float, tmax);
    }
    conv_raise_range_flag(p.range_flag, amax);
    conv_amax_commit(mem.out_amax, seen, nullptr, 0xffffffffu, amax);
  }
#ifdef SHF_K1_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && (bid == 0 || bid == 100 || bid == 300) && p.Cin == 512)
    printf("[k1] blk%d wave%d prologue %llu loop %llu (%d chunks) epilogue %llu: barrier %llu park0 %llu store0 %llu park1 %llu store1 %llu drain %llu\n", bid, wave_u, t_loop - t_entry, t_epi - t_loop, nchunks,
           (unsigned long long)__builtin_amdgcn_s_memtime() - t_epi, tb - t_epi, te[0] - tb, te[1] - te[0], te[2] - te[1], te[3] - te[2], (unsigned long long)__builtin_amdgcn_s_memtime() - te[3]);
#endif
}

}  // namespace shf
