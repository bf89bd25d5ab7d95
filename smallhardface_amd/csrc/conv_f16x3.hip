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
// Kernels, one header each (this file is the host side: weight packs, knobs, launchers):
//   conv_f16x3_w4d.h   conv_mfma_f16x3_w4d_kernel -- the dual-tile 4-wave family: Cin >= 64, Cout % 128 == 0 (80 % of the time)
//   conv_f16x3_pc.h    conv_mfma_f16x3_pc_kernel  -- the fused first pair conv1_1 -> conv1_2, producer / consumer waves
//   conv_f16x3_k1.h    conv_mfma_f16x3_k1_kernel  -- the 1x1 layers with Cout % 256 == 0 as a plain GEMM over flat pixels
//   conv_f16x3_h3.h    conv_mfma_f16x3_heads3_kernel -- the three shared-weight dilated heads (dilation 1 / 2 / 4) in ONE launch
//   conv_f16x3_8w.h    conv_mfma_f16x3_kernel     -- 8 waves: what the others cannot take (Cout 64, other 1x1s, unaligned views)
//   conv_f16x3_types.h vector types, the hi / lo split, the MFMA wrapper, conv1_1's K-slot map
// Common structure: tile 256 px (16x16) x BN couts; a STAGE is one kernel row (3 taps) of one channel chunk: its weight
// slabs are double-buffered in LDS and arrive by LDS DMA; the halo tile is staged once per chunk and reused by all 9 taps.
// 8-wave / first-pair LDS rows are [hi: 32 halfs][lo: 32 halfs][16 B pad] = 144 B (conflict-free ds_read_b128 over
// consecutive rows); the dual-tile family has its own geometry: 16-channel chunks, planar halo tiles, unscaled low parts.
// Epilogues: the 4-wave and first-pair kernels run the MFMA as D[cout][pixel] and store from registers
// (conv_common.h conv_epilogue_regs*, conv_epilogue_pool_only); the 8-wave kernel runs D[pixel][cout] and transposes the
// tile through LDS (conv_stage_tile / conv_flush_tile).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "conv_common.h"

#include "conv_f16x3_types.h"
#include "conv_f16x3_8w.h"
#include "conv_f16x3_w4d.h"
#include "conv_f16x3_pc.h"
#include "conv_f16x3_k1.h"
#include "conv_f16x3_h3.h"

namespace shf {

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

// The fused first pair's pack (conv_f16x3_pc.h): (Cout,Cin,3,3) fp32 -> [Cin/32][ky][kx][Cout][8 x 8 halfs] fp16, 128-byte rows
// WITHOUT padding; the eight 16-byte pieces of cout row r -- hi k 0-7, 8-15, 16-23, 24-31, then lo (x 2^11) the same -- sit at
// slot (piece + (r >> 1)) mod 8, so that the 16 lanes of a ds_read_b128 group (16 consecutive rows, one logical piece)
// cover all 16 bank groups.  The 6 KB of LDS this saves against the 144-byte rows pay for the halo rows' padding there.
size_t split16r_conv_weight_halfs(int Cout, int Cin, int k) { return (size_t)Cout * (Cin / 32) * k * k * 64; }
void pack_conv_weights_split16r(const float* w, int Cout, int Cin, int k, void* dst_, bool bf) {
  _Float16* dst = (_Float16*)dst_;
  const int taps = k * k;
  memset(dst_, 0, split16r_conv_weight_halfs(Cout, Cin, k) * 2);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const float x = w[((size_t)co * Cin + ci) * taps + t];
        const _Float16 h = bf ? host_bf16_as_half(x) : (_Float16)x;
        const _Float16 l = bf ? (_Float16)0 : (_Float16)((x - (float)h) * f16x3::LO_SCALE);
        const size_t row = (((size_t)(ci / 32) * taps + t) * Cout + co) * 64;
        const int kk = ci % 32, rot = (co >> 1) & 7;
        dst[row + (((kk >> 3) + rot) & 7) * 8 + (kk & 7)] = h;
        dst[row + ((4 + (kk >> 3) + rot) & 7) * 8 + (kk & 7)] = l;
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
  int pc_tab;          // SHF_F16X3_PC_TAB: 0 = the persistent first pair decodes its tiles one by one (the path launches with more than
                       // 300 tiles per block take anyway); bit-identical
  int heads3;          // SHF_F16X3_HEADS3: 1 (default) = the three shared-weight dilated heads as ONE launch (conv_f16x3_h3.h), 0 = one
                       // launch per head; bit-identical
  bool pc;             // SHF_F16X3_PC (default on): the fused first pair on the producer / consumer kernel
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
    q.heads3 = env_int("SHF_F16X3_HEADS3", 1);
    q.pc_tab = env_int("SHF_F16X3_PC_TAB", 1);
    q.pc = env_int("SHF_F16X3_PC", 1) != 0;
    q.pc_persist = env_int("SHF_F16X3_PC_PERSIST", 1) != 0;
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

// (net_forward.cpp / net_detect.cpp: will launch_conv_f16x3_group(as, n) take the dual-tile family?  Then the sub-launch hook does the profiling.)
// The family addresses its input with 32-bit BYTE offsets from the member's base: inputs of 4 GiB and more, and
// unaligned views (scalar epilogue), take the 8-wave kernel.
bool conv_f16x3_group_is_dual(const ConvArgs* as, int n) {
  if (!as[0].wsplit16h || as[0].img || as[0].k != 3 || as[0].dil != 1 || as[0].out.C % 128) return false;
  if (!conv_f16x3_uses_w4(as[0].in.C) || !views_aligned(as, n)) return false;
  for (int i = 0; i < n; ++i)
    if ((unsigned long long)as[i].in.B * as[i].in.H * as[i].in.W * as[i].in.cstride * 4ull >= (1ull << 32)) return false;
  return true;
}

// the dilated shared-weight heads (dilation 2 / 4) on the family's DIL form: the same conditions but for the dilation
// (what they exclude -- Cin < 64, Cout % 128, unaligned views -- takes the 8-wave kernel's DIL form)
bool conv_f16x3_group_is_dilated_w4(const ConvArgs* as, int n) {
  if (!as[0].wsplit16h || as[0].img || as[0].k != 3 || (as[0].dil != 2 && as[0].dil != 4) || as[0].out.C % 128)
    return false;
  if (!conv_f16x3_uses_w4(as[0].in.C) || !views_aligned(as, n)) return false;
  for (int i = 0; i < n; ++i)
    if ((unsigned long long)as[i].in.B * as[i].in.H * as[i].in.W * as[i].in.cstride * 4ull >= (1ull << 32)) return false;
  return true;
}

// 1x1 layers on the GEMM kernel (conv_f16x3_k1.h): all 256 couts of a pixel in one block, the family's weight pack with
// k = 1, activations in the split format (net_graph.cpp counts such a layer as a split-format reader) or fp32
bool conv_f16x3_k1_gemm_shape(int Cin, int Cout) {
  return conv_f16x3_uses_w4(Cin) && Cin % 32 == 0 && Cout % 256 == 0;
}
bool conv_f16x3_group_is_k1_gemm(const ConvArgs* as, int n) {
  if (!as[0].wsplit16h || as[0].img || as[0].k != 1 || as[0].bf16 || as[0].pool.p || !conv_f16x3_k1_gemm_shape(as[0].in.C, as[0].out.C))
    return false;
  if (!views_aligned(as, n)) return false;
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
  const bool dil_ok = dil == 1 || dil == 2 || dil == 4;
  if (k == 1) return pad == 0 && Cin % 32 == 0 && Cout % 64 == 0;
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
  ConvK p = {};
  p.wp = (const float*)a.wsplit16;
  p.wph = nullptr;
  p.wscale_inv = 1.f;
  p.tile_base = 0;
  p.ntile_blocks = 0;
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
  const bool vec_ok = views_aligned(as, n);   // (unaligned channel views: the 8-wave kernel's scalar stores)
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
  if (FUSE1 && BN == 64 && conv_f16x3_uses_pc() && vec_ok && p.Cin == 64 && p.Cout == 64 && p.w1f && a.wsplit16r) {
    p.wp = (const float*)a.wsplit16r;   // its own pack: 128-byte rotated rows (pack_conv_weights_split16r)
    // two halo tiles (both channel chunks of conv1_1's output; rows of 18 pixels x 144 B + 96 B) + the weight double buffer
    // (128-byte rows) + the image patch
    constexpr size_t HPP = (HP + 31) / 32 * 32;
    // (+ conv1_1's weight fragments 8 KiB, its 64 biases, the row-tile counter)
    const size_t lds_pc = 2 * (size_t)(TH + 2) * ((TW + 2) * ROWB + 96) + 2 * 3 * (size_t)BN * 128 + (3 * (TH + 4) * (TW + 4) + 8) * sizeof(float) + HPP +
                          BN * sizeof(float) + 8192 + 64 * sizeof(float) + 16 + 64 + 300 * 4;
if (lds_pc > 160 * 1024) { set_error("conv f16x3: the fused first pair does not fit the LDS"); return -1; }
    if (knobs().pc_persist) {
      // one block per CU walks the tiles (tile = block, block + grid, ...)
      p.ntile_blocks = (int)tiles;
      const dim3 gp((unsigned)std::min<long long>(tiles, knobs().cus));
      {   // the per-block tile table: fits (300 tiles per block) and packs (image < 256, tile row / column < 1024)?
        bool ok = knobs().pc_tab != 0 && (tiles + gp.x - 1) / gp.x <= 300;
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
  } else if (BN == 128 && !FUSE1 && KS == 3 && DIL > 1 && conv_f16x3_group_is_dilated_w4(as, n)) {
    // the dilated heads on the family's DIL form: single 16-row tiles (halo tiles of (16 + 2 DIL)^2 pixels, two buffer sets)
    if constexpr (BN == 128 && !FUSE1 && KS == 3 && (DIL == 2 || DIL == 4)) {
      p.wph = a.wsplit16h;
      p.wscale_inv = a.wscale_inv;
      p.tile_base = 0;
      p.ntile_blocks = (int)(tiles * p.nct);
      const size_t as_b = 4 * ((size_t)(16 + 2 * DIL) * 24 * 16 + 32);
      const size_t ldsd = 2 * as_b + 2 * 3 * (size_t)BN * 64 + BN * sizeof(float);
      const dim3 gd((unsigned)(tiles * p.nct));
#define SHF_W4D_DIL(SPLIT)                                                                                                    \
      {                                                                                                                      \
        if (a.bf16) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 1, true, DIL>), gd, dim3(256), ldsd, s, p);     \
        else if (a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, 4, 1, 3, false, DIL>), gd, dim3(256), ldsd, s, p); \
        else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, 4, 1, 2, false, DIL>), gd, dim3(256), ldsd, s, p); \
        else hipLaunchKernelGGL((conv_mfma_f16x3_w4d_kernel<SPLIT, 4, 1, 1, false, DIL>), gd, dim3(256), ldsd, s, p);            \
      }
      if (a.in_split) SHF_W4D_DIL(true)
      else SHF_W4D_DIL(false)
#undef SHF_W4D_DIL
    }
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
  if (dual) {   // the family's per-block phase sums (conv_f16x3_w4d.h): wave 0 of every block of the layer's launch(es)
    unsigned long long h[4];
    hipStreamSynchronize(s);
    hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    if (h[3])
      fprintf(stderr, "[w4d timing] Cin %d Cout %d rows %d: %llu blocks, per block cycles: prologue %.0f, K loop %.0f (%d stages: %.0f each), epilogue %.0f\n",
              p.Cin, p.Cout, 4 * mt, h[3], (double)h[0] / h[3], (double)h[1] / h[3], p.Cin / 16 * 3, (double)h[1] / h[3] / (p.Cin / 16 * 3),
              (double)h[2] / h[3]);
  } else {
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

// 1x1 GEMM kernel: blocks of 256 pixels of each member's flat pixel list x 256 couts
static int launch_f16x3_k1(const ConvArgs* as, int n, hipStream_t s) {
  const ConvArgs& a = as[0];
  ConvK p = {};
  p.wp = (const float*)a.wsplit16;
  p.wph = a.wsplit16h;
  p.wscale_inv = a.wscale_inv;
  p.tile_base = 0;
  p.ntile_blocks = 0;
  p.pc_tab = 0;
  p.bias = a.bias;
  p.Cin = a.in.C; p.Cout = a.out.C;
  p.in_stride = a.in.cstride; p.out_stride = a.out.cstride;
  p.dil = 1; p.relu = a.relu | 16 | (a.out_split ? 32 : 0);
  p.pool_stride = 0;
  p.nct = p.Cout / 256;
  p.nmem = n;
  for (int i = 0; i < MAX_GROUP; ++i) p.tile_starts[i] = 0x7fffffff;
  p.dbg = nullptr;
  p.range_flag = a.range_flag;
  p.w1t = nullptr; p.w1f = nullptr; p.b1 = nullptr;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    const ConvArgs& q = as[i];
    if (q.in.C != p.Cin || q.out.C != p.Cout || q.in.cstride != p.in_stride || q.out.cstride != p.out_stride ||
        q.wsplit16h != a.wsplit16h || q.in_split != a.in_split || q.out_split != a.out_split || q.k != 1) {
      set_error("conv group: members must share the layer");
      return -1;
    }
    ConvMember& m = p.m[i];
    m.in = q.in.p + q.in.coff;
    m.out = q.out.p + q.out.coff;
    m.pool = nullptr;
    m.img = nullptr;
    m.in_amax = q.in_amax; m.out_amax = q.out_amax; m.pool_amax = nullptr;
    m.B = q.in.B; m.H = q.in.H; m.W = q.in.W;
    const long long npix = (long long)m.B * m.H * m.W;
    if (npix >= (1ll << 31)) { set_error("conv f16x3: 2^31 pixels or more in one member"); return -1; }
    m.tiles_x = 1;
    m.tiles_per_img = (int)((npix + 255) / 256);
    m.inv_tiles_x = 0; m.inv_tiles_per_img = 0;
    m.tile_start = (int)tiles;
    p.tile_starts[i] = (int)tiles;
    tiles += m.tiles_per_img;
  }
  if (tiles * p.nct >= (1ll << 31)) { set_error("conv f16x3: grid too large"); return -1; }
  const size_t lds = 3 * 256 * 128 + 2 * 2 * 256 * 64;   // three activation chunks + two weight chunks: all 160 KiB
  const dim3 g((unsigned)(tiles * p.nct));
#define SHF_K1_LAUNCH(SPLIT)                                                                                         \
  {                                                                                                                 \
    if (a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_k1_kernel<SPLIT, 3>), g, dim3(256), lds, s, p);            \
    else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_k1_kernel<SPLIT, 2>), g, dim3(256), lds, s, p);       \
    else hipLaunchKernelGGL((conv_mfma_f16x3_k1_kernel<SPLIT, 1>), g, dim3(256), lds, s, p);                         \
  }
  if (a.in_split) SHF_K1_LAUNCH(true)
  else SHF_K1_LAUNCH(false)
#undef SHF_K1_LAUNCH
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// The three shared-weight dilated heads as ONE launch (conv_f16x3_h3.h): a1 / a2 / a4 = the dilation-1 / 2 / 4 layers'
// arguments, member by member.  They must read the same input with the same weights and differ in dilation and output only.
bool conv_f16x3_group_is_heads3(const ConvArgs* a1, const ConvArgs* a2, const ConvArgs* a4, int n) {
  if (!knobs().heads3 || n < 1 || n > MAX_GROUP) return false;
  const ConvArgs& a = a1[0];
  if (!a.wsplit16h || a.bf16 || a.img || a.k != 3 || a.dil != 1 || a.out.C != 128 || a.in.C % 16 || !conv_f16x3_uses_w4(a.in.C)) return false;
  for (int i = 0; i < n; ++i) {
    const ConvArgs* q[3] = {&a1[i], &a2[i], &a4[i]};
    if (q[1]->dil != 2 || q[2]->dil != 4) return false;
    for (int d = 0; d < 3; ++d) {
      const ConvArgs& b = *q[d];
      if (b.k != 3 || b.pad != b.dil || b.wsplit16h != a.wsplit16h || b.bias != a.bias || b.nprod != a.nprod || b.bf16 || b.pool.p ||
          b.relu != a.relu || b.in_split != a.in_split || b.out_split != a.out_split || b.out.C != 128 || b.out.cstride != a.out.cstride ||
          b.in.p != q[0]->in.p || b.in.coff != q[0]->in.coff || b.in.cstride != a.in.cstride || b.in.C != a.in.C ||
          b.in.B != q[0]->in.B || b.in.H != q[0]->in.H || b.in.W != q[0]->in.W || b.range_flag != a.range_flag)
        return false;
    }
    if (!views_aligned(q[0], 1) || !views_aligned(q[1], 1) || !views_aligned(q[2], 1)) return false;
    if ((unsigned long long)a1[i].in.B * a1[i].in.H * a1[i].in.W * a1[i].in.cstride * 4ull >= (1ull << 32)) return false;
  }
  return true;
}

int launch_conv_f16x3_heads3(const ConvArgs* a1, const ConvArgs* a2, const ConvArgs* a4, int n, hipStream_t s) {
  if (!conv_f16x3_group_is_heads3(a1, a2, a4, n)) { set_error("conv f16x3: not a shared-weight dilation-1/2/4 triple"); return -1; }
  const ConvArgs& a = a1[0];
  ConvK p = {};
  p.wp = (const float*)a.wsplit16;
  p.wph = a.wsplit16h;
  p.wscale_inv = a.wscale_inv;
  p.bias = a.bias;
  p.Cin = a.in.C; p.Cout = a.out.C;
  p.in_stride = a.in.cstride; p.out_stride = a.out.cstride;
  p.dil = 1; p.relu = a.relu | 16 | (a.out_split ? 32 : 0);
  p.nct = 1;
  p.nmem = n;
  for (int i = 0; i < MAX_GROUP; ++i) p.tile_starts[i] = 0x7fffffff;
  p.range_flag = a.range_flag;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    ConvMember& m = p.m[i];
    m.in = a1[i].in.p + a1[i].in.coff;
    m.out = a1[i].out.p + a1[i].out.coff;
    m.out2 = a2[i].out.p + a2[i].out.coff;
    m.out3 = a4[i].out.p + a4[i].out.coff;
    m.in_amax = a1[i].in_amax;
    m.out_amax = a1[i].out_amax; m.out2_amax = a2[i].out_amax; m.out3_amax = a4[i].out_amax;
    m.B = a1[i].in.B; m.H = a1[i].in.H; m.W = a1[i].in.W;
    m.tiles_x = (m.W + 15) / 16;
    m.tiles_per_img = m.tiles_x * ((m.H + 7) / 8);
    m.inv_tiles_x = conv_inv32(m.tiles_x);
    m.inv_tiles_per_img = conv_inv32(m.tiles_per_img);
    m.tile_start = (int)tiles;
    p.tile_starts[i] = (int)tiles;
    tiles += (long long)m.tiles_per_img * m.B;
    if ((unsigned long long)m.tiles_per_img * m.B * (unsigned long long)m.tiles_per_img >= (1ull << 32)) {
      set_error("conv f16x3: more than 2^32 / tiles-per-image pixel tiles in one member");
      return -1;
    }
  }
  if (tiles >= (1ll << 31)) { set_error("conv f16x3: grid too large"); return -1; }
  p.ntile_blocks = (int)tiles;
  const size_t as_b = 4 * ((size_t)16 * 24 * 16 + 32);
  const size_t lds = 2 * as_b + 2 * 3 * (size_t)128 * 64 + 128 * sizeof(float);
  const dim3 g((unsigned)tiles);
#define SHF_H3_LAUNCH(SPLIT)                                                                                        \
  {                                                                                                                \
    if (a.nprod >= 3) hipLaunchKernelGGL((conv_mfma_f16x3_heads3_kernel<SPLIT, 3>), g, dim3(256), lds, s, p);       \
    else if (a.nprod == 2) hipLaunchKernelGGL((conv_mfma_f16x3_heads3_kernel<SPLIT, 2>), g, dim3(256), lds, s, p);  \
    else hipLaunchKernelGGL((conv_mfma_f16x3_heads3_kernel<SPLIT, 1>), g, dim3(256), lds, s, p);                    \
  }
  if (a.in_split) SHF_H3_LAUNCH(true)
  else SHF_H3_LAUNCH(false)
#undef SHF_H3_LAUNCH
  SHF_HIP_OK(hipGetLastError());
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
#define SHF_W4D_DIL_ATTR(DILV)                                                                                                  \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 3, false, DILV>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<true, 4, 1, 3, false, DILV>)) \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 2, false, DILV>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<true, 4, 1, 2, false, DILV>)) \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 1, false, DILV>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<true, 4, 1, 1, false, DILV>)) \
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 1, true, DILV>))
  SHF_W4D_DIL_ATTR(2) SHF_W4D_DIL_ATTR(4)
#undef SHF_W4D_DIL_ATTR
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 2, 1, true>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 4, 1, 1, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 2, 2, 1, true>)) SHF_LDS_ATTR((conv_mfma_f16x3_w4d_kernel<false, 2, 1, 1, true>))
  SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<true, 3>)) SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<true, 2>)) SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<true, 1>))
  SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<true, 3>)) SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<true, 2>)) SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<true, 1>))
  SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<false, 3>)) SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<false, 2>)) SHF_LDS_ATTR((conv_mfma_f16x3_heads3_kernel<false, 1>))
  SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<false, 3>)) SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<false, 2>)) SHF_LDS_ATTR((conv_mfma_f16x3_k1_kernel<false, 1>))
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
  if (as[0].k == 1 && conv_f16x3_group_is_k1_gemm(as, n)) return launch_f16x3_k1(as, n, s);
  if (as[0].k == 1)
    return (as[0].out.C % 128 == 0) ? launch_f16x3_t<128, false, 1, 1>(as, n, s) : launch_f16x3_t<64, false, 1, 1>(as, n, s);
  if (as[0].dil == 2) return conv_f16x3_group_is_dilated_w4(as, n) ? launch_f16x3_t<128, false, 2>(as, n, s) : launch_f16x3_t<64, false, 2>(as, n, s);
  if (as[0].dil == 4) return conv_f16x3_group_is_dilated_w4(as, n) ? launch_f16x3_t<128, false, 4>(as, n, s) : launch_f16x3_t<64, false, 4>(as, n, s);
  return (as[0].out.C % 128 == 0) ? launch_f16x3_t<128, false>(as, n, s) : launch_f16x3_t<64, false>(as, n, s);
}

}  // namespace shf
