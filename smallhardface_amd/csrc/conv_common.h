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
};

struct ConvK {
  const float* wp;   // packed weights
  const float* bias;
  int Cin, Cout;
  int in_stride, out_stride;
  int dil, relu;     // relu bit 0: ReLU; bit 3 (8): do NOT write the un-pooled output
  int pool_stride;   // floats per pixel of the pool buffer
  int nct, nmem;
  const float* w1t;  // FUSE1: first-layer weights transposed to [27][64]
  const float* b1;   // FUSE1: first-layer bias [64]
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

// Shared epilogue: bias + ReLU, store, and the optional fused MAX 2x2/2 pool (pooling_layer.cu:11-47).
// Registers 4q..4q+3 of a lane are C rows s + 8q + 4*(lane>>5), s = 0..3: with row_to_pixel they are
// the (dy,dx) = (s>>1, s&1) corners of ONE pooling window, so the pool is a max over four registers
// of the same lane -- no cross-lane traffic.  Windows on a ragged edge are clipped like Caffe's.
template <typename GetV>
__device__ __forceinline__ void conv_store_tile(GetV getv, float bv, int relu_flags, int gy0, int gx0, int kh,
                                                int H, int W, int b, int cout, float* __restrict__ gout,
                                                int out_stride, float* __restrict__ gpool, int pool_stride) {
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

}  // namespace shf
