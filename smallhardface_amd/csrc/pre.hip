// Image pre-processing of one pyramid unit on the device: mean subtraction, bilinear resize,
// horizontal flip, zero padding to a multiple of MAX_RESOLUTION and the HWC -> CHW transpose
// (lib/utils/test_utils.py:29-46 _get_image_blob, lib/utils/blob.py:16-32 im_list_to_blob,
// lib/test.py:35-38 pad, :150 flip), one thread per output pixel.
//
// The arithmetic is the host mirror's (smallhardface_amd/test_utils.py resize_bilinear), which restates OpenCV's
// published INTER_LINEAR path for a CV_64F image (cv::resize / hal::resize / HResizeLinear / VResizeLinear in
// modules/imgproc/src/resize.cpp): scale = 1 / f once in double; the source coordinate is narrowed to FLOAT before its
// floor is subtracted (in float); the weights 1.f - fx and fx are floats widened to double; the mean-subtracted image
// is float64 (uint8 -> f32 minus f64 PIXEL_MEANS); products and sums are separate roundings (this file is built with
// -ffp-contract=off), and only the finished level is narrowed to f32.  HBM-bound and tiny next to the convolutions:
// 3 bytes in (x4 taps, L2-served), 12 bytes out per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shf_internal.h"

namespace {

struct AxisCoef {
  int i0, i1;
  double a0, a1;
};

__device__ inline AxisCoef axis_coef(int d, int n_src, double inv_f) {
  float fx = (float)(((double)d + 0.5) * inv_f - 0.5);   // (float)((dx + 0.5) * scale_x - 0.5)
  int sx = (int)floorf(fx);                              // cvFloor
  fx -= (float)sx;
  if (sx < 0) {
    sx = 0;
    fx = 0.f;
  }
  if (sx >= n_src - 1) {
    sx = n_src - 1;
    fx = 0.f;
  }
  AxisCoef c;
  c.i0 = sx;
  c.i1 = sx + 1 < n_src ? sx + 1 : n_src - 1;
  c.a0 = (double)(1.f - fx);
  c.a1 = (double)fx;
  return c;
}

__global__ void __launch_bounds__(256) pyramid_level_kernel(const uint8_t* __restrict__ im, int im_h, int im_w,
                                                            double scale, int flip, double m0, double m1, double m2,
                                                            float* __restrict__ out, int H, int W, int lvl_h,
                                                            int lvl_w) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W) return;
  const size_t plane = (size_t)H * W;
  float* o = out + (size_t)y * W + x;
  if (y >= lvl_h || x >= lvl_w) {
    o[0] = 0.f;
    o[plane] = 0.f;
    o[2 * plane] = 0.f;
    return;
  }
  const double inv_f = 1.0 / scale;
  const AxisCoef cy = axis_coef(y, im_h, inv_f);
  const AxisCoef cx = axis_coef(flip ? lvl_w - 1 - x : x, im_w, inv_f);
  const uint8_t* r0 = im + (size_t)cy.i0 * im_w * 3;
  const uint8_t* r1 = im + (size_t)cy.i1 * im_w * 3;
  const double mean[3] = {m0, m1, m2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double a00 = (double)(float)r0[cx.i0 * 3 + c] - mean[c];
    const double a01 = (double)(float)r0[cx.i1 * 3 + c] - mean[c];
    const double a10 = (double)(float)r1[cx.i0 * 3 + c] - mean[c];
    const double a11 = (double)(float)r1[cx.i1 * 3 + c] - mean[c];
    const double top = a00 * cx.a0 + a01 * cx.a1;
    const double bot = a10 * cx.a0 + a11 * cx.a1;
    o[c * plane] = (float)(top * cy.a0 + bot * cy.a1);
  }
}

}  // namespace

namespace shf {

int launch_pyramid_level(const uint8_t* im, int im_h, int im_w, double scale, int flip, const double* means,
                         float* out, int H, int W, int lvl_h, int lvl_w, hipStream_t s) {
  dim3 grid((W + 255) / 256, H);
  pyramid_level_kernel<<<grid, 256, 0, s>>>(im, im_h, im_w, scale, flip, means[0], means[1], means[2], out, H, W,
                                            lvl_h, lvl_w);
  return (int)hipGetLastError();
}

}  // namespace shf
