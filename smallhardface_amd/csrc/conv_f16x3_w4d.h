// Split-fp16 convolution, the dual-tile 4-wave family: every 3x3 / dilation-1 layer with Cin >= 64 and Cout % 128 == 0.
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

#include "conv_f16x3_types.h"

namespace shf {

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
// DIL (round 4): the dilated shared-weight heads (dilation 2 / 4, Cin = Cout = 128) run here too, as single 16-row tiles:
// the halo tile is (16 + 2 DIL)^2 pixels -- 20 or 24 per row, inside the 24-pixel plane rows --, a tap is DIL pixels / DIL
// plane rows further, nothing else changes (111 / 123 KB of LDS).  They used to take the 8-wave kernel's 64-cout form (a
// fragment read per MFMA: 0.29 issued).
template <bool IN_SPLIT, int MT_, int NTILE, int NP = 3, bool BF = false, int DIL = 1>
__global__ __launch_bounds__(256) void conv_mfma_f16x3_w4d_kernel(ConvK p) {
  static_assert(!BF || (NP == 1 && !IN_SPLIT), "bf16 mode: one product, fp32 activations in HBM");
  static_assert((MT_ == 4 || MT_ == 2) && (NTILE == 1 || NTILE == 2), "16- or 8-row tiles, one or two per block");
  static_assert(DIL == 1 || ((DIL == 2 || DIL == 4) && MT_ == 4 && NTILE == 1), "dilated layers: single 16-row tiles");
  constexpr int MT = MT_, TH = 4 * MT, TW = 16, HTW = TW + 2 * DIL, HTH = TH + 2 * DIL, HP = HTH * HTW;
  constexpr int UNUSED = 24 - HTW;                    // plane-row pixels no halo pixel uses (they take the ragged last piece)
  constexpr int KC = 16, BN = 128, NT = 256;
  constexpr int PROW = 24 * 16;                       // 384 B per halo-tile row of a plane (18 pixels used)
  constexpr int PLANE = HTH * PROW + 32;              // 6 944 B (16-row tiles) / 3 872 B (8-row tiles)
  constexpr int AS_B = 4 * PLANE;                     // 27 776 B / 15 488 B per halo tile
  constexpr int NB_B = NTILE * AS_B;                  // one buffer set (the tiles of one chunk)
  constexpr int WROWB = 64;                           // weight rows: no padding, the 16-byte pieces rotated by row / 4
  constexpr int SLAB_B = BN * WROWB;                  // 8 192 B per tap slab
  constexpr int ALD = (HP * 4 + NT - 1) / NT;         // 16-byte halo pieces per thread and tile: 6 (16 rows) or 3
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                           // [2 buffer sets][NTILE][4 planes][HTH][24 px][16 B]
  unsigned char* Bs = smem + 2 * NB_B;                // [2 buffers][3 taps][BN][64 B]
  float* biasL = (float*)(Bs + 2 * 3 * SLAB_B);       // [BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int bid = blockIdx.x;
#ifdef SHF_CONV_TIMING
  const unsigned long long tt_entry = __builtin_amdgcn_s_memtime();
#endif
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
  // (the block's biases: requested here, parked behind the halo pieces at the end of the prologue -- parked right behind
  // the halo requests, the compiler's wait for this one load was a wait for all of them)
  const float bias_v = (tid < BN && p.bias) ? p.bias[ct * BN + tid] : 0.f;

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
  unsigned slot_bits0 = conv_act_slot_request(g0.in_amax), slot_bits1 = NTILE == 2 ? conv_act_slot_request(g1.in_amax) : 0u;


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
      const int gy = g.ty0 - DIL + hy, gx = g.tx0 - DIL + hx;
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
  // (read through the scalar cache, requested with the geometry: conv_act_slot_request)
  conv_act_slot_wait(slot_bits0, slot_bits1);
  const int e_t0 = conv_act_exponent_of_bits(slot_bits0);
  const int e_t1 = NTILE == 2 ? conv_act_exponent_of_bits(slot_bits1) : 0;
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
      const h2 l01 = conv_split_lo(x01, h01), l23 = conv_split_lo(x23, h23);
      v = make_float4(__builtin_bit_cast(float, h01 * sc.hi1), __builtin_bit_cast(float, h23 * sc.hi1),
                      __builtin_bit_cast(float, l01 * sc.lo1), __builtin_bit_cast(float, l23 * sc.lo1));
    }
  };
  // (set_off: byte offset of the buffer set the pieces go to)
  auto store_piece = [&](const float4& v, int t, int j, unsigned set_off) {
    const int idx = tid + NT * j;
    const int hp = idx >> 2, q = idx & 3;
    int hy = hp / HTW, hx = hp - hy * HTW;
    if constexpr (UNUSED > 0) {
      if (NT * (j + 1) > HP * 4) {
        // the ragged last piece: threads past the tile's end store theirs in the unused columns (18..23 at dilation 1) of
        // the first rows (no branch inside the stage -- it would split the scheduling region)
        static_assert((ALD * NT - HP * 4 + 3) / 4 <= (UNUSED > 0 ? UNUSED : 1) * HTH, "the ragged piece's parking columns");
        const int hpd = hp - HP, ry = hpd / (UNUSED > 0 ? UNUSED : 1);
        const bool past = idx >= HP * 4;
        hy = past ? ry : hy;
        hx = past ? HTW + hpd - ry * UNUSED : hx;
      }
    } else {
      static_assert(UNUSED > 0 || (HP * 4) % NT == 0, "a full 24-pixel halo row leaves no parking column: no ragged piece allowed");
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
  if (tid < BN) biasL[tid] = bias_v;
#ifdef SHF_CONV_TIMING
  const unsigned long long tt_pro = __builtin_amdgcn_s_memtime();
#endif

  unsigned seen0 = 0xffffffffu, seen0p = 0xffffffffu, seen1 = 0xffffffffu, seen1p = 0xffffffffu;
  // one stage = kernel row KY of the 16-channel chunk c.  MODE 1 (kernel row 1 of a chunk that has a successor):
  // request the pieces of chunk c + 1's halo tiles; MODE 2 (kernel row 2): convert them and park them in the other
  // buffer set, spread over the half-steps
  auto stage = [&](int c, auto KY_, auto MODE_) {
    constexpr int ky = decltype(KY_)::value;
    constexpr int MODE = decltype(MODE_)::value;
    const int st = c * 3 + ky;
    // this wave's share of W(st) (and, MODE 2, its halo pieces) has landed; its parked pieces are in LDS
    // (the BUILTIN, not an asm string: hipcc keeps its own count of the loads in flight and cannot see a wait inside an asm
    // statement, nor the DMA issues inside dma_w's.  With an invisible wait it guarded every halo piece of the MODE 2 stage
    // -- requested a whole stage earlier, long landed -- with a vmcnt(11 - j) by ITS count, and since the counter really
    // holds this stage's DMA pieces as well, the later of those waited for weight pieces issued moments before: 1-2 % of
    // every two-tile and 16-row layer, same-box A/B.  The single 8-row tile form -- three pieces per thread, two blocks
    // per CU -- measured 3 % FASTER on conv2_1 with the asm form (699 / 713 against 734 / 725 us) and keeps it.)
    if constexpr (MT == 2 && NTILE == 1) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      __builtin_amdgcn_s_waitcnt(0x0070);             // vmcnt(0) lgkmcnt(0), expcnt untouched
      asm volatile("" ::: "memory");
    }
    __syncthreads();
    const int st_next = st + 1 < NST ? st + 1 : st;   // the last stage re-fetches itself (unused) instead of branching
    const int buf_next = (st + 1) & 1;
    const unsigned coff = chunk_off(c + 1);
    const unsigned char* Bst = Bs + (st & 1) * (3 * SLAB_B);
    half8 fa[2][2 * MT], fb[2][4];
    auto load_a = [&](int h, half8* a) {              // half-step h = NTILE kx + tile
      const unsigned char* Ap = As + (h % NTILE) * AS_B + ky * DIL * PROW + (h / NTILE) * DIL * 16;
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
      // (two-tile blocks of 8-row tiles -- conv5_x -- issue the stage's six weight pieces a half-step earlier: their half-steps
      // are half as long, and a piece issued in the last one had ~400 cycles to land before the next stage's wait; same-box
      // 3 x 300 vs 3 x 312 us.  For 16-row tiles the even spread stays: 2-2-2-0-0-0 measured 2 % slower there.)
      constexpr int DMA_N2[6] = {MT == 2 ? 2 : 1, 1, 1, 1, 1, MT == 2 ? 0 : 1}, DMA_J2[6] = {0, MT == 2 ? 2 : 1, MT == 2 ? 3 : 2, MT == 2 ? 4 : 3, MT == 2 ? 5 : 4, MT == 2 ? 6 : 5};
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
#ifdef SHF_CONV_TIMING
  const unsigned long long tt_k = __builtin_amdgcn_s_memtime();
#endif
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
#ifdef SHF_CONV_TIMING
  // per-block phase cycles of wave 0 (summed over all blocks of the launch): prologue (entry .. first stage), K loop, epilogue
  if (p.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tt_end = __builtin_amdgcn_s_memtime();
    atomicAdd(p.dbg + 0, tt_pro - tt_entry);
    atomicAdd(p.dbg + 1, tt_k - tt_pro);
    atomicAdd(p.dbg + 2, tt_end - tt_k);
    atomicAdd(p.dbg + 3, 1ull);
  }
#endif
  conv_raise_range_flag(p.range_flag, conv_absmax_bits(amax0, amax1));
  conv_amax_commit(g0.out_amax, seen0, g0.pool ? g0.pool_amax : nullptr, seen0p, amax0);
  if constexpr (NTILE == 2) {
    if (has1) conv_amax_commit(g1.out_amax, seen1, g1.pool ? g1.pool_amax : nullptr, seen1p, amax1);   // (wave-uniform)
  }
}


}  // namespace shf
