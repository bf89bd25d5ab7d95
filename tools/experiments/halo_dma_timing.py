#!/usr/bin/env python3
"""Timing-only experiment (WRONG results): what would the dual-tile family's halo hand-over cost as LDS DMA?

DESIGN.md 8.7 priced "consumer-ready activation pairs + halo tiles by LDS DMA" without a measurement of the DMA
side.  This builds three patched copies of csrc/conv_f16x3_w4d.h into variants/*.so (tools/build_variant.sh; nothing
in the shipped sources changes) and tools/scratch/ab_lib.sh alternates them with the shipped library on ONE box:

  halo_contig  : the in-loop halo pieces still travel through registers (12 loads, 96 v_pk_mul, 12 ds_write_b128 per
                 thread and chunk) but every wave instruction reads ONE contiguous KiB near the tile -- what a
                 channel-blocked activation layout ([C/16][H][W][hi16|lo16]) would give the requests
  dma_contig   : the in-loop pieces by global_load_lds_dwordx4 straight into the other buffer set (7 instructions per
                 wave and tile), contiguous KiB each; no conversion, no parking; the third product reads b_hi * 2^-11
                 formed in registers (8 v_pk_mul_f16 per tap and wave) -- the scheme that needs no activation exponent
  dma_scatter  : the same with TODAY's activation format: lane -> (plane, halo pixel) of the planar LDS layout, i.e. 64
                 different 128-byte lines per instruction

    python3 tools/experiments/halo_dma_timing.py          # build the three variants (CPU container)
    gpurun -- tools/scratch/ab_lib.sh default exp_halo_contig exp_dma_contig exp_dma_scatter
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "smallhardface_amd", "csrc")


def rep(s, a, b, count=1):
    assert s.count(a) == count, (s.count(a), a[:70])
    return s.replace(a, b)


def contig_offsets(s):
    # every (thread, piece) reads 16 bytes of ONE contiguous run that starts at the tile's first row (clamped into the
    # member's tensor): 256 threads x 16 B = 4 KiB per piece index, 1 KiB per wave instruction
    return rep(s, """      a_goff[t][j] = in ? pix + (IN_SPLIT ? (unsigned)((q >> 1) * 64 + (q & 1) * 16) : (unsigned)(q * 16)) : 0u;""",
               """      {
        const unsigned total = (unsigned)(g.H * g.W * in_stride_v) * 4u;
        const int row0 = g.ty0 > 0 ? g.ty0 : 0;
        unsigned base = (unsigned)((g.b * g.H + row0) * g.W * in_stride_v) * 4u;
        base = base + 40000u < total ? base : (total > 40000u ? total - 40000u : 0u);
        a_goff[t][j] = (base & ~1023u) + (unsigned)(idx * 16) + (unsigned)(t * 0);
        (void)pix; (void)q;
      }""")


def dma_variant(s, scatter):
    # MODE 1: DMA issues instead of register loads
    s = rep(s, """        if (h == 0) {
#pragma unroll
          for (int j = 0; j < ALD; ++j) areg0[j] = *(const float4*)((const char*)g0.in + (a_goff[0][j] + coff));
          n_vmem += ALD;
        } else if (NTILE == 2 && h == 1) {
#pragma unroll
          for (int j = 0; j < ALD; ++j) areg1[j] = *(const float4*)((const char*)g1.in + (a_goff[NTILE - 1][j] + coff));
          n_vmem += ALD;
        }""", """        if (h == 0) {
          dma_halo(g0.in, 0, coff);
          n_vmem += HDMA;
        } else if (NTILE == 2 && h == 1) {
          dma_halo(g1.in, NTILE - 1, coff);
          n_vmem += HDMA;
        }""")
    # MODE 2: nothing to convert or park
    s = rep(s, """        for (int k = h * PP; k < (h + 1) * PP && k < NPC; ++k) {
          const int t = k / ALD, j = k % ALD;""", """        for (int k = h * PP; k < (h + 1) * PP && k < NPC && false; ++k) {
          const int t = k / ALD, j = k % ALD;""")
    # the DMA helper, defined right before the prologue (after store_piece)
    addr = ("(const unsigned char*)in_ + (size_t)(d_goff[t][k] + coff_)" if scatter else
            "(const unsigned char*)in_ + (size_t)(d_goff[t][k] + coff_)")
    s = rep(s, """  // prologue
  float4 areg0[ALD], areg1[ALD];""", """  constexpr int NPIECE = AS_B / 1024;               // whole KiB pieces of a halo tile (the 32-byte plane pads ride along)
  constexpr int HDMA = (NPIECE + 3) / 4;            // DMA instructions per wave and tile
  unsigned d_goff[NTILE][HDMA];
  {
    auto fill = [&](const Geo& g, int t) {
#pragma unroll
      for (int k = 0; k < HDMA; ++k) {
        int piece = wave + 4 * k;
        piece = piece < NPIECE ? piece : NPIECE - 1;
""" + ("""        const int sbyte = piece * 1024 + lane * 16;       // byte inside the tile's LDS image
        const int q = sbyte / PLANE, r = (sbyte - q * PLANE) >> 4;
        const int hy = r / 24, hx = r - hy * 24;
        const int gy = g.ty0 - DIL + hy, gx = g.tx0 - DIL + hx;
        const bool in = hy < HTH && hx < HTW && ((unsigned)gy < (unsigned)g.H) && ((unsigned)gx < (unsigned)g.W);
        const unsigned pix = (unsigned)(((g.b * g.H + gy) * g.W + gx) * in_stride_v) * 4u;
        d_goff[t][k] = in ? pix + (unsigned)((q >> 1) * 64 + (q & 1) * 16) : 0u;
""" if scatter else """        const unsigned total = (unsigned)(g.H * g.W * in_stride_v) * 4u;
        const int row0 = g.ty0 > 0 ? g.ty0 : 0;
        unsigned base = (unsigned)((g.b * g.H + row0) * g.W * in_stride_v) * 4u;
        base = base + 40000u < total ? base : (total > 40000u ? total - 40000u : 0u);
        d_goff[t][k] = (base & ~1023u) + (unsigned)(piece * 1024 + lane * 16);
""") + """      }
    };
    fill(g0, 0);
    if constexpr (NTILE == 2) fill(g1, 1);
  }
  unsigned park_off_s = NB_B;
  auto dma_halo = [&](const float* in_, int t, unsigned coff_) {
#pragma unroll
    for (int k = 0; k < HDMA; ++k) {
      int piece = wave_u + 4 * k;
      piece = piece < NPIECE ? piece : NPIECE - 1;
      const unsigned char* ga = """ + addr + """;
      const unsigned lds = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) unsigned char*)(As + park_off_s + t * AS_B + piece * 1024);
      asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(ga));
    }
  };

  // prologue
  float4 areg0[ALD], areg1[ALD];""")
    s = rep(s, """    park_off = NB_B - park_off;
  }""", """    park_off = NB_B - park_off;
    park_off_s = NB_B - park_off_s;
  }""")
    # third product against b_hi * 2^-11 formed in registers (the scheme without an activation exponent)
    s = rep(s, """      auto mfmas = [&](f32x16 (&acc)[MT][2]) {""", """      half8 bsc[2];
      {
        typedef _Float16 h8v __attribute__((ext_vector_type(8)));
        const _Float16 tiny = (_Float16)0.00048828125f;
        bsc[0] = __builtin_bit_cast(half8, __builtin_bit_cast(h8v, bf[0]) * tiny);
        bsc[1] = __builtin_bit_cast(half8, __builtin_bit_cast(h8v, bf[2]) * tiny);
      }
      auto mfmas = [&](f32x16 (&acc)[MT][2]) {""")
    s = rep(s, """              acc[tm][tn] = mma16<BF>(bf[2 * tn], a[2 * tm + 1], acc[tm][tn]);""",
            """              acc[tm][tn] = mma16<BF>(bsc[tn], a[2 * tm + 1], acc[tm][tn]);""")
    return s


def main():
    base = open(os.path.join(SRC, "conv_f16x3_w4d.h")).read()
    out = os.path.join(ROOT, "variants", "_src")
    os.makedirs(out, exist_ok=True)
    variants = {"exp_halo_contig": contig_offsets(base), "exp_dma_contig": dma_variant(base, False),
                "exp_dma_scatter": dma_variant(base, True)}
    hip = open(os.path.join(SRC, "conv_f16x3.hip")).read()
    only = sys.argv[1:]
    for name, text in variants.items():
        if only and name not in only:
            continue
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "conv_f16x3_w4d.h"), "w").write(text)
        open(os.path.join(d, "conv_f16x3.hip"), "w").write(hip)     # includes "conv_f16x3_w4d.h": the patched copy beside it wins
        env = dict(os.environ, SRC_OVERRIDE=os.path.join(d, "conv_f16x3.hip"))
        subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), name, "conv_f16x3.hip"], env=env, cwd=ROOT)


if __name__ == "__main__":
    sys.exit(main())
