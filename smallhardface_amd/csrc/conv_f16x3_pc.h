// Split-fp16 convolution, the fused first pair conv1_1 -> conv1_2 (producer / consumer waves, persistent walk).
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

#include "conv_f16x3_types.h"

namespace shf {

// Producer / consumer kernel of the fused first pair (conv1_1 -> conv1_2, Cin = Cout = 64): conv1_1's 64-channel output
// (1.8 GB per image on the bench pyramid) never touches HBM.  With only 64 couts a wave of the 8-wave kernel owns ONE
// 32-pixel MFMA row tile (MT = 1) and needs a ds_read_b128 per MFMA -- LDS-bound at ~40 % matrix-pipe use.  Here:
//  * conv1_1 runs ON THE MATRIX CORES as D[cout][pixel]: [324 halo px x 27 taps (padded to 32)] x [32 x 64 couts] =
//    11 row tiles of 32 halo pixels, 12 split-fp16 MFMAs each.  The 20x20x3 image patch is parked in LDS ALREADY SPLIT
//    (fp16 hi | fp16 lo << 16 per dword), the K slots are ordered (first_conv_slot_tap) so that a lane's sixteen patch reads
//    are base(kk, kh) + immediate, and a fragment register is one byte permute of two patch words.  Weights fragments and
//    biases live in LDS; a row tile = all operand reads, then the twelve MFMAs, then two epilogues (bias, ReLU, zero
//    outside the image, split, four 8-byte stores of hi and of lo per chunk into the halo tiles As0 / As1).
//  * waves 0-3 are CONSUMERS (one per SIMD): 64 px x 64 couts = 4 accumulator tiles each, 8 fragment reads per 12 MFMAs,
//    the six k-steps of a stage software-pipelined (~2.65 k cycles per stage against 2.3 k of pure MFMA issue); their
//    epilogue stores from registers (conv_common.h conv_epilogue_pool_only when only the pooled map is kept).  Waves 4-7
//    are PRODUCERS: every weight DMA, and the walk's bookkeeping.
//  * PERSIST: one block per CU WALKS the tiles (tile = block, block + grid, ...), decoded once per block into a packed
//    LDS table.  Under tile t's K loop the producers read tile t + 2's record (stage 4), request tile t + 1's patch (a pair per
//    thread in each of stages 1-3, IN FRONT of that stage's weight pieces) and park it a stage later (stages 2-4), write its
//    validity flags (stage 5), hand tile t + 1's geometry to all waves through LDS (stage 3) and request its first weight
//    stage (stage 5): 2.0-2.7 k cycles of own work in every stage (round 5; stages 1 / 2 carried 2.9 k / 3.4 k before).
//  * A matrix stream and a vector stream do not overlap on a SIMD (tools/scratch/coissue.hip: a partner wave gets ~3 vector
//    issues per MFMA), so conv1_1 of tile t + 1 does NOT run under the K loop (measured: the K loop grows by what the
//    producers run) but BESIDE THE OTHER VECTOR PHASE: behind a post-K barrier its 11 row tiles are claimed one at a time
//    (an LDS counter) by whichever wave is free -- the producers at once, the consumers when their epilogue is out.
//    Per tile: K loop 6 x ~2.7 k cycles (stages 0 / 1 stretched to ~3.2 / 3.7 k by the producers' chores), epilogue +
//    conv1_1 phase 7.5-8.5 k, hand-over to stage 0 1.3-1.9 k: 1.84 -> 1.50 ms per image against round 3.
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
  constexpr int HPP = (HP + 31) / 32 * 32;          // 352: tile rows padded to whole 32-row MFMA tiles
  // HALO TILES (round 5): a pixel is ROWB = 144 B ([hi 32 | lo 32 | 16 B]: eight consecutive pixels of a row fall on eight
  // different 16-byte bank groups), a halo ROW is 18 pixels + 96 B = 2 688 B = 128 mod 256: the 16 lanes of a
  // ds_read_b128 group are 8 pixels of row y and 8 of row y + 1 (row_to_pixel), and with the plain 18 x 144 = 2 592 B rows
  // (32 mod 256) the second row's groups fell two slots beside the first's -- a 2-way conflict on every A-fragment read
  // (SQ_LDS_BANK_CONFLICT 0.32 of the kernel's LDS cycles in rounds 3-4, and the consumers' K loop is LDS-bound: 8 KiB of
  // fragments per 12 MFMAs and wave).  The 3.4 KB the padding costs come from the weight rows (below).
  constexpr int AROW = HTW * ROWB + 96;             // 2 688 B per halo row
  constexpr int AT_B = HTH * AROW;                  // 48 384 B per halo tile
  static_assert(AROW % 256 == 128, "consecutive halo rows on complementary halves of the bank row");
  // WEIGHT ROWS are 128 B without padding (the 8-wave kernel's pack has 144-byte rows): the eight 16-byte pieces of cout
  // row r -- hi k 0-7 .. 24-31, lo k 0-7 .. 24-31 -- are rotated by (r >> 1) mod 8 (pack_conv_weights_split16r), so the 16
  // lanes of a B-fragment read (16 consecutive rows, one logical piece) still cover all 16 bank groups; a tap slab is
  // 8 KiB = 8 DMA pieces, a stage 24 = six rounds of the four producer waves with no ragged one.
  constexpr int WROWB = 128;
  unsigned char* As0 = smem;                        // [HTH][AROW] channels  0..31 of conv1_1's output
  unsigned char* As1 = smem + AT_B;                 // [HTH][AROW] channels 32..63
  unsigned char* Bs = smem + 2 * AT_B;              // [2][3][BN][WROWB]
  // [3][PH][PW] image patch, already split: fp16 hi in the low half of a dword, fp16 lo (x 2^11) in the high half (bf16 mode:
  // the bf16 pattern | 0) -- conv1_1's fragments are then gathered with one byte permute per register, no conversion
  // (round 4; the conversions used to be redone for every fragment element: ~200 vector instructions per row tile).
  // (+ 8 dwords: half-wave 1's zero-weight slots read one element past a tap)
  unsigned* patch = (unsigned*)(Bs + 2 * 3 * BN * WROWB);
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

  constexpr int SLAB_B = BN * WROWB;         // 8 KiB
  constexpr int PCS_SLAB = SLAB_B / 1024;    // 8
  constexpr int PCS = 3 * PCS_SLAB;          // 24 one-KiB pieces per stage
  const size_t slab = (size_t)p.Cout * (WROWB / 2);
  const _Float16* wbase = (const _Float16*)p.wp;
  auto dma_w = [&](int stage, int buf) {     // producer waves only: 6 rounds of 4 pieces
    unsigned dma_l16 = (unsigned)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) * 16u;   // (formed here: not kept across the stages)
    asm volatile("" : "+v"(dma_l16));
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
#ifdef SHF_PC_BUILTIN_DMA
      const unsigned char* src = ws_ + (size_t)sl * slab * 2 + within * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(bd_ + pc * 1024), 16, 0, 0);
#else
      // (inline asm, like the family's dma_w: behind the BUILTIN the compiler -- which cannot tell the DMA's LDS destination
      // from any other LDS address -- drained vmcnt before the producers' next LDS access, so their chores of a stage (patch
      // parking, validity flags, the walk's bookkeeping, and after stage 5 the first conv1_1 row tile) started only when
      // the seven pieces just issued had landed, instead of running under their flight.  Completion is waited for by
      // hand at the top of the next stage.)
      const unsigned char* ub = ws_ + (size_t)sl * slab * 2 + within * 1024;
      const unsigned lds = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(bd_ + pc * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(dma_l16), "s"(ub));
#endif
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
    // conv1_1's operands -- weight fragments [n][kk][hi / lo] and the bias quads of couts 8q + 4kh .. + 3 -- are the SAME for
    // every row tile: a wave reads them from LDS ONCE per phase (before its first row tile of a tile's conv1_1) and keeps them
    // in registers over the 2-3 row tiles it gets through (round 6; until then every row tile re-read its 16 KiB: 16 of the 33
    // LDS instructions of a row tile and a full LDS round trip at its start).  They are not held across the K loop.
    half8 bwn[2][2][2];   // [n][kk][hi / lo]
    float4 bq[2][4];      // [n][register quad]
    auto conv1_operands = [&]() {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl)
            bwn[n][kk][hl] = *(const half8*)(w1L + ((size_t)((n * 2 + kk) * 2 + hl) * 64 + (i1 + 32 * kh1)) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[n][q] = *(const float4*)(b1L + n * 32 + 8 * q + 4 * kh1);
      }
    };
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
        // (packed: v_pk_fma_f32 + v_pk_add_f32 per register pair -- the same fma-then-add the scalar form compiled to -- then the
        // ReLU per value)
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 s01 = __builtin_elementwise_fma(f32x2{cc[n][4 * q], cc[n][4 * q + 1]}, f32x2{LO_INV, LO_INV}, f32x2{cm[n][4 * q], cm[n][4 * q + 1]}) +
                            f32x2{bq[n][q].x, bq[n][q].y};
          const f32x2 s23 = __builtin_elementwise_fma(f32x2{cc[n][4 * q + 2], cc[n][4 * q + 3]}, f32x2{LO_INV, LO_INV}, f32x2{cm[n][4 * q + 2], cm[n][4 * q + 3]}) +
                            f32x2{bq[n][q].z, bq[n][q].w};
          v[4 * q] = fmaxf(s01[0], 0.f);
          v[4 * q + 1] = fmaxf(s01[1], 0.f);
          v[4 * q + 2] = fmaxf(s23[0], 0.f);
          v[4 * q + 3] = fmaxf(s23[1], 0.f);
        }
        if (need_ok) {
          const bool ok = okb != 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = ok ? v[r] : 0.f;
        }
        unsigned char* row = (n ? As1 : As0) + hy * AROW + hx * ROWB + kh1 * 8;   // (row_ok lanes: hp is their own pixel)
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
            l0 = conv_split_lo(x0, h0);   // (conv_common.h: v_fma_mixlo / mixhi_f16)
            l1 = conv_split_lo(x1, h1);
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
      conv1_operands();
      conv1_tile(wave_u, 0, 2, halo_inside);
      if (wave_u < 2 * (NMT - 8)) conv1_tile(8 + wave_u % (NMT - 8), wave_u / (NMT - 8), wave_u / (NMT - 8) + 1, halo_inside);
      amax1 = conv_absmax_bits(amax1, fmaxf((float)amax1h[0], (float)amax1h[1]));
    }

  TileGeo nxt_pre = mem;
  if (PERSIST && !consumer && tile + gstride < ntiles) nxt_pre = decode(tile + gstride);
#ifdef SHF_CONV_TIMING
  unsigned long long ts_k = 0, ts_bar = 0, ts_role = 0, ts_tail = 0, t_role_end = 0, ts_st[6] = {0, 0, 0, 0, 0, 0}, ts_own[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long ts_epi = 0, n_claimed = 0;   // consumers: their epilogue alone; every wave: conv1_1 row tiles it claimed
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
  int a_off[MT], b_off[4];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = (wm * 2 * MT + t * 2 + dy) * AROW + px * ROWB + kh * 16;
  // B fragments: cout row t * 32 + i, logical piece kh + 2 (k half) [+ 4: lo] at slot (piece + (row >> 1)) & 7
#pragma unroll
  for (int x = 0; x < 4; ++x) b_off[x] = i * WROWB + ((kh + 2 * x + (i >> 1)) & 7) * 16;
  f32x16 accm[MT][2], accc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[a][c][r] = 0.f; accc[a][c][r] = 0.f; }

  auto mma_stage = [&](const unsigned char* Atile, int ky, int buf) {
    const unsigned char* Arow = Atile + ky * AROW;
    const unsigned char* Bst = Bs + buf * (3 * SLAB_B);
    half8 fa[2][2 * MT], fb[2][4];
    auto load_frag = [&](int s_, half8* a, half8* bf) {
      const unsigned char* Ap = Arow + (s_ >> 1) * ROWB + (s_ & 1) * 32;
      const unsigned char* Bp = Bst + (s_ >> 1) * SLAB_B;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        a[2 * t] = *(const half8*)(Ap + a_off[t]);
        a[2 * t + 1] = *(const half8*)(Ap + a_off[t] + 64);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf[2 * t] = *(const half8*)(Bp + t * (32 * WROWB) + b_off[s_ & 1]);
        bf[2 * t + 1] = *(const half8*)(Bp + t * (32 * WROWB) + b_off[2 + (s_ & 1)]);
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
#ifdef SHF_PC_BUILTIN_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    // (the builtin: the compiler's own count of the loads in flight must see this wait -- the patch registers requested in
    // stage 1 are parked in stage 2 BEHIND that stage's DMA issues, which it cannot see)
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
#endif
    __syncthreads();
    PC_T();
    if (consumer) {
      if (st == 5) {   // the unit's max |output| slots, read under the last stage (conv_amax_peek)
        seen = conv_amax_peek(mem.out_amax);
        seenp = conv_amax_peek(mem.pool ? mem.pool_amax : nullptr);
      }
      mma_stage(st < 3 ? As0 : As1, st % 3, st & 1);
    } else {
      if (!(PERSIST && st >= 1 && st <= 3) && st + 1 < 6) dma_w(st + 1, (st + 1) & 1);   // (stages 1-3: behind the patch requests, below)
      if constexpr (PERSIST) {
        // (measured, not kept: s_setprio 3 around these chores -- no change: what made a producer's stage-1 work 4.3 k cycles
        // was not issue arbitration but the patch loads queueing behind the stage's seven 1-KiB weight pieces)
        int lane_p = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (not kept across the stages)
        asm volatile("" : "+v"(lane_p));
        const int ptid = (wave_u - 4) * 64 + lane_p;
        if (st == 4 && tile + 2 * gstride < ntiles) nxt_pre = p.pc_tab ? decode_tab(k_walk + 2) : decode(tile + 2 * gstride);
        // THE PRODUCERS' CHORES ARE SPREAD OVER THE STAGES (round 5; per-role cycle counters, -DSHF_CONV_TIMING): with the patch
        // requested in stage 1 and parked -- with the validity flags -- in stage 2, the producers' own work was 2.9 k / 3.4 k
        // cycles in those stages against the consumers' 2.5-2.6 k, and 1.7 k / 1.7 k / 1.1 k in stages 3-5: stages 1 and 2 waited
        // for the producers (3.0 k / 3.4 k per stage instead of 2.6 k).  Now a thread's three patch pairs are requested one per
        // stage in stages 1-3, each is parked a stage later (stages 2-4: its load was waited for at that stage's top), and the
        // validity flags are written in stage 5: ~2.0-2.5 k of own work in every stage.
        static_assert(NPF2 == 3, "one patch pair per producer thread and stage in stages 1-3");
        if (st >= 1 && st <= 3 && has_next) {
          // x-PAIRS of patch elements, one 8-byte load each (a load instruction costs a producer wave 100-200 cycles beside
          // the consumers' stream: 12 wave-level loads instead of 20).  With an even level width a pair is inside or outside
          // the image as a whole and 8-byte aligned: the patch starts at column tx0 - 2 (even).
#pragma unroll
          for (int k = st - 1; k < st; ++k) {
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
        if (st >= 1 && st <= 3) dma_w(st + 1, (st + 1) & 1);
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
        if (st >= 2 && st <= 4 && has_next) {   // (pair st - 2: requested a stage ago, waited for at this stage's top)
#pragma unroll
          for (int k = st - 2; k < st - 1; ++k) {
            const int pi = ptid + 256 * k;
            if (pi < 3 * PH * (PW / 2)) *(uint2*)(patch + 2 * pi) = make_uint2(patch_word(pvn[2 * k]), patch_word(pvn[2 * k + 1]));
            amax1 = conv_absmax_bits(conv_absmax_bits(amax1, pvn[2 * k]), pvn[2 * k + 1]);
          }
        }
        if (st == 5 && has_next) {
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
#ifdef SHF_CONV_TIMING
  ts_epi += __builtin_amdgcn_s_memtime() - t_barx;
#endif
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
    conv1_operands();
#pragma unroll 1
    for (;;) {
      unsigned got = 0u;
      if (lane_p == 0) got = atomicAdd(ctrL, 1u);
      const int m = __builtin_amdgcn_readfirstlane((int)got);
      if (m >= NMT) break;
#ifdef SHF_CONV_TIMING
      ++n_claimed;
#endif
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
    // (the pointers come back from LDS as integers: cast through the GLOBAL address space, or every access through them is a
    // FLAT instruction -- counted in lgkmcnt as well as vmcnt, so that each LDS wait of the epilogue and of the row-tile
    // claims behind a store waited for that store to reach memory; found in the ISA in round 5: the persistent form was the
    // only conv kernel with flat_store)
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) const float gcfloat;
    typedef __attribute__((address_space(1))) unsigned gunsigned;
    mem = TileGeo{(int)rd(0), (int)rd(1), (int)rd(2), (int)rd(3), (int)rd(4), (const float*)(gcfloat*)rd64(6), (float*)(gfloat*)rd64(8),
                  (float*)(gfloat*)rd64(10), (unsigned*)(gunsigned*)rd64(12), (unsigned*)(gunsigned*)rd64(14)};
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
#ifdef SHF_CONV_TIMING
  if ((bid == 100) && lane == 0)
    printf("[pc-role] blk%d wave%d per tile: epilogue alone (consumers; producers: 0) %llu, conv1_1 row tiles claimed %.2f\n", bid, wave,
           ts_epi / (n_walk ? n_walk : 1), (double)n_claimed / (n_walk ? n_walk : 1));
#endif
#undef PC_T
}


}  // namespace shf
