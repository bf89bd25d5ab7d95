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
  int B, H, W;
  int tiles_x, tiles_per_img, tile_start;  // tile_start: first pixel-tile index of this member
};

struct ConvK {
  const float* wp;   // packed weights
  const float* bias;
  int Cin, Cout;
  int in_stride, out_stride;
  int dil, relu;
  int nct, nmem;
  unsigned long long* dbg;  // SHF_CONV_TIMING builds only: per-wave phase cycle sums
  ConvMember m[MAX_GROUP];
};

// row i (0..31) of a 32-row MFMA tile -> pixel inside the wave's 2x16 strip.
// The two low bits walk a 2x2 window so that the 4 consecutive C rows a lane owns
// form one pooling window (kept for a fused 2x2 max-pool epilogue).
__device__ __forceinline__ void row_to_pixel(int i, int& dy, int& px) {
  dy = (i >> 1) & 1;
  px = ((i >> 2) << 1) | (i & 1);
}

}  // namespace shf
