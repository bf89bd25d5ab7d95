// Split-fp16 convolution, the three shared-weight dilated heads in ONE launch (head_1 / head_2 / head_4 of
// models/test_different_dilation_template.prototxt:480-552: three 3x3 convolutions 128 -> 128 + ReLU over the SAME input
// with the SAME weights `head_w` / `head_b`, dilation = pad = 1, 2, 4).
// (part of the one translation unit conv_f16x3.hip: see its header for the arithmetic and the kernel map)
#pragma once
#include "conv_common.h"

#include "conv_f16x3_types.h"

namespace shf {

// The dual-tile family's data path (conv_f16x3_w4d.h: one accumulator per output, 16-channel chunks, planar halo tiles
// in two buffer sets, weights by LDS DMA one stage ahead, register epilogue) around ONE 8 x 16-pixel tile per block
// whose halo tile is cut for dilation 4 -- (8 + 8) x (16 + 8) = 16 x 24 pixels, exactly the 24-pixel plane rows -- and
// serves all three dilations: dilation d's tap (ky, kx) is the fragment at halo row 4 + (ky - 1) d, column
// 4 + (kx - 1) d, i.e. lane offset + an immediate like every other fragment address.  A STAGE is still one kernel
// row of a chunk and its three weight slabs are fetched ONCE for the three dilations (a third of the three launches'
// weight traffic and B-fragment reads: the nine half-steps of a stage are (kx, d) with the B fragments of tap kx held
// over d = 1, 2, 4), the halo tile of a chunk is fetched once instead of three times (18^2 + 20^2 + 24^2 pixel tiles
// before), and a block pays one prologue for 3 x 24 stage-equivalents.  Three accumulator sets of 2 x 2 tiles = 192
// registers; 97 KB of LDS, one block per CU.
// Every output is formed by the same operations in the same order as in the family's own <.., DIL> forms (chunk, kernel
// row, tap; hi*hi, lo*hi, hi*lo), so the launch is bit-identical to the three it replaces (knob SHF_F16X3_HEADS3=0).
template <bool IN_SPLIT, int NP = 3>
__global__ __launch_bounds__(256) void conv_mfma_f16x3_heads3_kernel(ConvK p) {
  constexpr int MT = 2, TH = 4 * MT, TW = 16, PADH = 4, HTW = TW + 2 * PADH, HTH = TH + 2 * PADH, HP = HTH * HTW;
  static_assert(HTW == 24, "the halo rows are exactly the 24-pixel plane rows");
  constexpr int KC = 16, BN = 128, NT = 256;
  constexpr int PROW = 24 * 16;
  constexpr int PLANE = HTH * PROW + 32;
  constexpr int AS_B = 4 * PLANE;                     // 24 704 B per halo tile
  constexpr int WROWB = 64;
  constexpr int SLAB_B = BN * WROWB;
  constexpr int ALD = HP * 4 / NT;                    // 6 sixteen-byte halo pieces per thread and chunk
  static_assert(HP * 4 == ALD * NT, "no ragged piece");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                           // [2 buffer sets][4 planes][HTH][24 px][16 B]
  unsigned char* Bs = smem + 2 * AS_B;                // [2 buffers][3 taps][BN][64 B]
  float* biasL = (float*)(Bs + 2 * 3 * SLAB_B);       // [BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wm = wave >> 1;
  const int nchunks = p.Cin / KC;
  const int NST = nchunks * 3;
  const size_t slab = (size_t)p.Cout * 32;            // halfs per tap slab of the whole layer
  const _Float16* wbase = (const _Float16*)p.wph;

  // weight DMA (the family's): round r (0..5) of a wave moves 1-KiB piece q = wave + 4 r of the stage's 24
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
      const unsigned char* ub = ws_ + w_goff(r);
      const unsigned lds = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(bd_ + w_loff(r));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane16), "s"(ub));
    }
  };
  dma_w(0, 0, 0, W_ROUNDS);
  const float bias_v = (tid < BN && p.bias) ? p.bias[tid] : 0.f;

  // the block's tile (nct == 1: block = pixel tile)
  int pt = (int)blockIdx.x;
  const int mi = conv_find_member(p, pt);
  const ConvMember mem = p.m[mi];
  pt -= mem.tile_start;
  int gb, ty_, tx_;
  conv_split_tile(mem, pt, gb, ty_, tx_);
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, H = mem.H, W = mem.W;
  unsigned slot_bits = conv_act_slot_request(mem.in_amax), slot_none = 0u;

  unsigned a_goff[ALD];
  unsigned a_valid = 0;
  int in_stride_v = p.in_stride;
  asm volatile("" : "+v"(in_stride_v));
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    const int idx = tid + NT * j;
    const int hp = idx >> 2, q = idx & 3;
    const int hy = hp / HTW, hx = hp - hy * HTW;
    const int gy = ty0 - PADH + hy, gx = tx0 - PADH + hx;
    const bool in = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
    const unsigned pix = (unsigned)(((gb * H + gy) * W + gx) * in_stride_v) * 4u;
    a_goff[j] = in ? pix + (IN_SPLIT ? (unsigned)((q >> 1) * 64 + (q & 1) * 16) : (unsigned)(q * 16)) : 0u;
    a_valid |= in ? (1u << j) : 0u;
  }
  auto chunk_off = [&](int c16) -> unsigned {
    return IN_SPLIT ? (unsigned)((c16 >> 1) * 128 + (c16 & 1) * 32) : (unsigned)(c16 * 64);
  };
  // activation exponent of the unit (conv_f16x3_w4d.h)
  conv_act_slot_wait(slot_bits, slot_none);
  const int e_t = conv_act_exponent_of_bits(slot_bits);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const int lo_shift = (tid & 2) ? 11 : 0;
  const unsigned sc_pk_f1 = conv_pk_pow2_f16(e_t - lo_shift), sc_hi1 = conv_pk_pow2_f16(e_t), sc_lo1 = conv_pk_pow2_f16(e_t - 11);
  auto convert = [&](float4& v, int vbit) {
    const unsigned keep = (unsigned)((int)(a_valid << (31 - vbit)) >> 31);   // all ones / zero
    if constexpr (IN_SPLIT) {
      const h2 f1 = __builtin_bit_cast(h2, sc_pk_f1 & keep);
      float* e = &v.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h2 x = __builtin_bit_cast(h2, e[k]);
        x = x * f1;
        e[k] = __builtin_bit_cast(float, x);
      }
    } else {
      const h2 hi1 = __builtin_bit_cast(h2, sc_hi1 & keep), lo1 = __builtin_bit_cast(h2, sc_lo1 & keep);
      const f32x2 x01 = {v.x, v.y}, x23 = {v.z, v.w};
      const h2 h01 = __builtin_convertvector(x01, h2), h23 = __builtin_convertvector(x23, h2);
      const h2 l01 = conv_split_lo(x01, h01), l23 = conv_split_lo(x23, h23);
      v = make_float4(__builtin_bit_cast(float, h01 * hi1), __builtin_bit_cast(float, h23 * hi1),
                      __builtin_bit_cast(float, l01 * lo1), __builtin_bit_cast(float, l23 * lo1));
    }
  };
  auto store_piece = [&](const float4& v, int j, unsigned set_off) {
    const int idx = tid + NT * j;
    const int hp = idx >> 2, q = idx & 3;
    const int hy = hp / HTW, hx = hp - hy * HTW;
    unsigned char* pix = As + set_off + hy * PROW + hx * 16;
    if constexpr (IN_SPLIT) {
      *(float4*)(pix + q * PLANE) = v;
    } else {
      *(float2*)(pix + (q >> 1) * PLANE + (q & 1) * 8) = make_float2(v.x, v.y);
      *(float2*)(pix + (2 + (q >> 1)) * PLANE + (q & 1) * 8) = make_float2(v.z, v.w);
    }
  };

  // prologue
  float4 areg[ALD];
#pragma unroll
  for (int j = 0; j < ALD; ++j) areg[j] = *(const float4*)((const char*)mem.in + a_goff[j]);

  const int i = lane & 31, kh = lane >> 5;
  int dy, px;
  row_to_pixel(i, dy, px);
  int a_off[MT], b_off[2];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = kh * PLANE + (wm * 2 * MT + t * 2 + dy) * PROW + px * 16;
  int a_delta = AS_B;
  unsigned park_off = AS_B;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    b_off[t] = (wn * 64 + t * 32 + i) * WROWB + ((kh + ((wn * 64 + t * 32 + i) >> 2)) & 3) * 16;
  f32x16 acc1[MT][2], acc2[MT][2], acc4[MT][2];     // dilation 1 / 2 / 4
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc1[a][c][r] = 0.f; acc2[a][c][r] = 0.f; acc4[a][c][r] = 0.f; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < ALD; ++j) {
    convert(areg[j], j);
    store_piece(areg[j], j, 0u);
  }
  if (tid < BN) biasL[tid] = bias_v;

  unsigned seen1 = 0xffffffffu, seen2 = 0xffffffffu, seen4 = 0xffffffffu;
  // one stage = kernel row KY of the 16-channel chunk c, for the three dilations.  MODE 1 (kernel row 1 of a chunk with a
  // successor): request the next chunk's halo pieces; MODE 2 (kernel row 2): convert + park them in the other buffer set
  auto stage = [&](int c, auto KY_, auto MODE_) {
    constexpr int ky = decltype(KY_)::value;
    constexpr int MODE = decltype(MODE_)::value;
    const int st = c * 3 + ky;
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): the builtin (conv_f16x3_w4d.h: the compiler's count)
    asm volatile("" ::: "memory");
    __syncthreads();
    const int st_next = st + 1 < NST ? st + 1 : st;
    const int buf_next = (st + 1) & 1;
    const unsigned coff = chunk_off(c + 1);
    const unsigned char* Bst = Bs + (st & 1) * (3 * SLAB_B);
    half8 fa[2][2 * MT], fb[2][4];
    auto load_a = [&](int h, half8* a) {              // half-step h = 3 kx + dilation index
      const int d = 1 << (h % 3), kx = h / 3;
      const unsigned char* Ap = As + (PADH + (ky - 1) * d) * PROW + (PADH + (kx - 1) * d) * 16;
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
        bf[2 * t + 1] = *(const half8*)(Bp + (b_off[t] ^ 32));
      }
    };
    load_a(0, fa[0]);
    load_b(0, fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NH = 9;
    constexpr int PARK_VALU = IN_SPLIT ? 4 : 7;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int kx = h / 3, di = h % 3;
      half8* a = fa[h & 1];
      half8* bf = fb[kx & 1];
      int n_ds = 0;
      if (h + 1 < NH) { load_a(h + 1, fa[(h + 1) & 1]); n_ds += 2 * MT; }
      if (di == 0 && kx + 1 < 3) { load_b(kx + 1, fb[(kx + 1) & 1]); n_ds += 4; }
      const int dma_n = h < W_ROUNDS ? 1 : 0;
      if (dma_n) dma_w(st_next, buf_next, h, 1);
      int n_vmem = dma_n;
      if constexpr (MODE == 1) {
        if (h == 0) {
#pragma unroll
          for (int j = 0; j < ALD; ++j) areg[j] = *(const float4*)((const char*)mem.in + (a_goff[j] + coff));
          n_vmem += ALD;
        }
      }
      if constexpr (MODE == 3) {
        if (h == 0) {
          seen1 = conv_amax_peek(mem.out_amax);
          seen2 = conv_amax_peek(mem.out2_amax);
          seen4 = conv_amax_peek(mem.out3_amax);
          n_vmem += 3;
        }
      }
      int n_park = 0;
      if constexpr (MODE == 2) {
        if (h < ALD) {
          convert(areg[h], h);
          store_piece(areg[h], h, park_off);
          ++n_park;
        }
      }
      auto mfmas = [&](f32x16 (&acc)[MT][2]) {
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = mma16<false>(bf[2 * tn], a[2 * tm], acc[tm][tn]);
        if constexpr (NP >= 2) {
#pragma unroll
          for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = mma16<false>(bf[2 * tn + 1], a[2 * tm], acc[tm][tn]);
        }
        if constexpr (NP >= 3) {
#pragma unroll
          for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = mma16<false>(bf[2 * tn], a[2 * tm + 1], acc[tm][tn]);
        }
      };
      if (di == 0) mfmas(acc1);
      else if (di == 1) mfmas(acc2);
      else mfmas(acc4);
      constexpr int NM = 2 * NP * MT;                 // MFMAs of the half-step
      if (h + 1 < NH || n_park > 0) {
        const int n_first = n_ds < NM ? n_ds : NM;
#pragma unroll
        for (int g = 0; g < n_first; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        const int n_rest = NM > n_first + 1 ? NM - n_first - 1 : 0;
#pragma unroll
        for (int g = 0; g < n_rest; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < n_vmem) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          if (n_park > 0) {
            __builtin_amdgcn_sched_group_barrier(0x002, PARK_VALU, 0);
            if (g == n_rest - 1) __builtin_amdgcn_sched_group_barrier(0x200, IN_SPLIT ? 1 : 2, 0);
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
    park_off = AS_B - park_off;
  }
  stage(nchunks - 1, integral_constant<int, 0>{}, integral_constant<int, 0>{});
  stage(nchunks - 1, integral_constant<int, 1>{}, integral_constant<int, 0>{});
  stage(nchunks - 1, integral_constant<int, 2>{}, integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last stage's (unused) self re-fetch, the slot peeks
  asm volatile("" : "+v"(seen1), "+v"(seen2), "+v"(seen4));
  seen1 = __builtin_amdgcn_readfirstlane(seen1);
  seen2 = __builtin_amdgcn_readfirstlane(seen2);
  seen4 = __builtin_amdgcn_readfirstlane(seen4);

  // register epilogue, one dilation after the other (the family's conv_epilogue_regs1)
  float amax1 = 0.f, amax2 = 0.f, amax4 = 0.f;
  {
    const bool relu = (p.relu & 1) != 0, main_split = (p.relu & 32) != 0;
    int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_e));
    const int i_e = lane_e & 31, kh_e = lane_e >> 5;
    int px_e, dy_e;
    row_to_pixel(i_e, dy_e, px_e);
    const int wn_e = wave_u & 1, wm_e = wave_u >> 1;
    int out_stride_e = p.out_stride;
    float wscale_inv_e = p.wscale_inv;
    asm volatile("" : "+v"(out_stride_e), "+v"(wscale_inv_e));
    float4 bias16[2][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
      bias16[g >> 2][g & 3] = *(const float4*)(biasL + wn_e * 64 + (g >> 2) * 32 + kh_e * 16 + 4 * (g & 3));
    const float out_scale = wscale_inv_e * __builtin_bit_cast(float, (unsigned)(127 - e_t) << 23);   // 2^-e, exact
    const bool interior = ty0 + TH <= H && tx0 + TW <= W;
    const int x = tx0 + px_e;
    auto tile_out = [&](f32x16 (&acc)[MT][2], float* outp, float& amax) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int cout16 = wn_e * 64 + tn * 32 + kh_e * 16;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
          int y = ty0 + wm_e * 2 * MT + tm * 2 + dy_e;
          asm volatile("" : "+v"(y));
          const bool valid = y < H && x < W;
          const unsigned pix_m = (unsigned)((gb * H + y) * W + x);
          float* pm = outp + (size_t)pix_m * (unsigned)out_stride_e;
          if (relu)
            conv_epilogue_regs1<true>(acc[tm][tn], out_scale, bias16[tn], valid, interior, pm, cout16, main_split, nullptr, false, false, amax);
          else
            conv_epilogue_regs1<false>(acc[tm][tn], out_scale, bias16[tn], valid, interior, pm, cout16, main_split, nullptr, false, false, amax);
        }
      }
    };
    tile_out(acc1, mem.out, amax1);
    tile_out(acc2, mem.out2, amax2);
    tile_out(acc4, mem.out3, amax4);
  }
  conv_raise_range_flag(p.range_flag, conv_absmax_bits(conv_absmax_bits(amax1, amax2), amax4));
  conv_amax_commit(mem.out_amax, seen1, nullptr, 0xffffffffu, amax1);
  conv_amax_commit(mem.out2_amax, seen2, nullptr, 0xffffffffu, amax2);
  conv_amax_commit(mem.out3_amax, seen4, nullptr, 0xffffffffu, amax4);
}

}  // namespace shf
