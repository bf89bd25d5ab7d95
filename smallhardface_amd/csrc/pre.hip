// Image pre-processing of one pyramid unit on the device: mean subtraction, bilinear resize,
// horizontal flip, zero padding to a multiple of MAX_RESOLUTION and the HWC -> CHW transpose
// (lib/utils/test_utils.py:29-46 _get_image_blob, lib/utils/blob.py:16-32 im_list_to_blob,
// lib/test.py:35-38 pad, :150 flip), one thread per output pixel.
//
// The arithmetic is the host mirror's (smallhardface_amd/test_utils.py resize_bilinear), which restates OpenCV's
// published INTER_LINEAR path for a CV_64F image (cv::resize / hal::resize / HResizeLinear / VResizeLinear in
// modules/imgproc/src/resize.cpp): scale = 1 / f once in double; the source coordinate is narrowed to FLOAT before its
// floor is subtracted (in float); the weights 1.f - fx and fx are floats widened to double (columns: weight zeroed at the
// border; rows: the two row indices clipped, weights kept -- row_coef); the mean-subtracted image
// is float64 (uint8 -> f32 minus f64 PIXEL_MEANS); products and sums are separate roundings (this file is built with
// -ffp-contract=off), and only the finished level is narrowed to f32.  When both scale factors are exactly 2x down,
// cv::resize replaces INTER_LINEAR by the INTER_AREA fast path (resizeAreaFast_Invoker<double, double, NoVec>): the four
// taps summed in one left-to-right chain times (double)0.25f, and (float)sum / count for windows that leave an odd-sized
// source -- restated here under `area2` (the launcher applies cv::resize's own test).  HBM-bound and tiny next to the convolutions:
// 3 bytes in (x4 taps, L2-served), 12 bytes out per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

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

// the vertical table: no border rule on the weight -- resizeGeneric_Invoker clips the two source ROW INDICES
// (clip(sy + k, 0, src_h), k = 0, 1) and keeps beta = {1.f - fy, fy} (test_utils.py _row_coeffs, oracle/resize.py y_table)
__device__ inline AxisCoef row_coef(int d, int n_src, double inv_f) {
  float fy = (float)(((double)d + 0.5) * inv_f - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  AxisCoef c;
  c.i0 = sy < 0 ? 0 : (sy < n_src ? sy : n_src - 1);
  c.i1 = sy + 1 < 0 ? 0 : (sy + 1 < n_src ? sy + 1 : n_src - 1);
  c.a0 = (double)(1.f - fy);
  c.a1 = (double)fy;
  return c;
}

__global__ void __launch_bounds__(256) pyramid_level_kernel(const uint8_t* __restrict__ im, int im_h, int im_w,
                                                            double scale, int flip, double m0, double m1, double m2,
                                                            float* __restrict__ out, int H, int W, int lvl_h,
                                                            int lvl_w, int area2) {
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
  const double mean[3] = {m0, m1, m2};
  if (area2) {
    const int sy0 = 2 * y, sx0 = 2 * (flip ? lvl_w - 1 - x : x);
    const bool full = sy0 + 2 <= im_h && sx0 + 2 <= im_w;   // dx < dwidth1 on a row with sy0 + scale_y <= height
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (sy0 >= im_h) {
        o[c * plane] = 0.f;
        continue;
      }
      if (full) {
        const uint8_t* p0 = im + ((size_t)sy0 * im_w + sx0) * 3 + c;
        const uint8_t* p1 = p0 + (size_t)im_w * 3;
        const double s00 = (double)(float)p0[0] - mean[c], s01 = (double)(float)p0[3] - mean[c];
        const double s10 = (double)(float)p1[0] - mean[c], s11 = (double)(float)p1[3] - mean[c];
        o[c * plane] = (float)((((s00 + s01) + s10) + s11) * (double)0.25f);
      } else {
        double sum = 0.0;
        int count = 0;
        for (int sy = 0; sy < 2 && sy0 + sy < im_h; ++sy)
          for (int sx = 0; sx < 2 && sx0 + sx < im_w; ++sx) {
            sum += (double)(float)im[((size_t)(sy0 + sy) * im_w + sx0 + sx) * 3 + c] - mean[c];
            ++count;
          }
        o[c * plane] = count ? (float)(double)((float)sum / (float)count) : 0.f;
      }
    }
    return;
  }
  const double inv_f = 1.0 / scale;
  const AxisCoef cy = row_coef(y, im_h, inv_f);
  const AxisCoef cx = axis_coef(flip ? lvl_w - 1 - x : x, im_w, inv_f);
  const uint8_t* r0 = im + (size_t)cy.i0 * im_w * 3;
  const uint8_t* r1 = im + (size_t)cy.i1 * im_w * 3;
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
  // cv::resize: iscale = cvRound(1 / f); INTER_LINEAR -> INTER_AREA when |1/f - iscale| < DBL_EPSILON and iscale == 2
  const double inv = 1.0 / scale;
  const int area2 = (std::fabs(inv - 2.0) < 2.220446049250313e-16 && (int)std::nearbyint(inv) == 2) ? 1 : 0;
  pyramid_level_kernel<<<grid, 256, 0, s>>>(im, im_h, im_w, scale, flip, means[0], means[1], means[2], out, H, W,
                                            lvl_h, lvl_w, area2);
  return (int)hipGetLastError();
}

}  // namespace shf
