/* Plain-C restatement of pieces of the reference's hot path -- TEST INFRASTRUCTURE ONLY
 * (an independent cross-check of oracle/oracle.py; never linked into the product).
 * Built by oracle/Makefile with -ffp-contract=off so fp32 expressions round like the
 * reference's unfused CUDA/numpy arithmetic.
 *
 *   oc_im2col / oc_conv      caffe/src/caffe/util/im2col.cpp:19-55, layers/base_conv_layer.cpp:256-279
 *   oc_maxpool               caffe/src/caffe/layers/pooling_layer.cpp:79-123,128-187 (MAX, ceil sizing)
 *   oc_iou                   lib/nms/nms_kernel.cu:24-32 (devIoU)
 *   oc_nms_bitmask           lib/nms/nms_kernel.cu:45-89 (mask kernel) + :138-150 (host reduce)
 *   oc_cpu_nms               lib/nms/cpu_nms.pyx:17-68 (the '>=' variant)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int is_a_ge_zero_and_a_lt_b(int a, int b) { return (unsigned)a < (unsigned)b; }

/* col: (C*kh*kw) x (Ho*Wo) */
void oc_im2col(const float* im, int C, int H, int W, int k, int pad, int stride, int dil, float* col) {
  const int Ho = (H + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int csize = H * W;
  for (int c = C; c--; im += csize)
    for (int kr = 0; kr < k; kr++)
      for (int kc = 0; kc < k; kc++) {
        int in_row = -pad + kr * dil;
        for (int orow = Ho; orow; orow--) {
          if (!is_a_ge_zero_and_a_lt_b(in_row, H)) {
            for (int oc = Wo; oc; oc--) *(col++) = 0;
          } else {
            int in_col = -pad + kc * dil;
            for (int oc = Wo; oc; oc--) {
              *(col++) = is_a_ge_zero_and_a_lt_b(in_col, W) ? im[in_row * W + in_col] : 0;
              in_col += stride;
            }
          }
          in_row += stride;
        }
      }
}

/* y (Co x Ho*Wo) = W (Co x C*k*k) * col + bias (rank-1 update), single image, group 1 */
int oc_conv(const float* x, int C, int H, int W, const float* w, const float* bias, int Co, int k, int pad,
            int stride, int dil, float* y) {
  const int Ho = (H + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
  const int K = C * k * k, N = Ho * Wo;
  float* col = (float*)malloc((size_t)K * N * sizeof(float));
  if (!col) return -1;
  oc_im2col(x, C, H, W, k, pad, stride, dil, col);
  for (int o = 0; o < Co; ++o) {
    float* yo = y + (size_t)o * N;
    for (int n = 0; n < N; ++n) yo[n] = 0.f;
    for (int kk = 0; kk < K; ++kk) {
      const float a = w[(size_t)o * K + kk];
      const float* cr = col + (size_t)kk * N;
      for (int n = 0; n < N; ++n) yo[n] += a * cr[n];
    }
    if (bias)
      for (int n = 0; n < N; ++n) yo[n] += bias[o];
  }
  free(col);
  return 0;
}

void oc_maxpool(const float* x, int C, int H, int W, int k, int stride, int pad, float* y, int* Ho_, int* Wo_) {
  int Ho = (int)ceil((double)(H + 2 * pad - k) / stride) + 1;
  int Wo = (int)ceil((double)(W + 2 * pad - k) / stride) + 1;
  if (pad) {
    if ((Ho - 1) * stride >= H + pad) --Ho;
    if ((Wo - 1) * stride >= W + pad) --Wo;
  }
  *Ho_ = Ho;
  *Wo_ = Wo;
  for (int c = 0; c < C; ++c)
    for (int ph = 0; ph < Ho; ++ph)
      for (int pw = 0; pw < Wo; ++pw) {
        int hs = ph * stride - pad, ws = pw * stride - pad;
        int he = hs + k < H ? hs + k : H, we = ws + k < W ? ws + k : W;
        if (hs < 0) hs = 0;
        if (ws < 0) ws = 0;
        float m = -3.402823466e+38f;
        for (int h = hs; h < he; ++h)
          for (int ww = ws; ww < we; ++ww) {
            const float v = x[((size_t)c * H + h) * W + ww];
            if (v > m) m = v;
          }
        y[((size_t)c * Ho + ph) * Wo + pw] = m;
      }
}

float oc_iou(const float* a, const float* b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

/* boxes: n x 5, ALREADY sorted by score descending (as _nms receives them); keep: indices into that order */
int oc_nms_bitmask(const float* boxes, int n, float thresh, int* keep) {
  const int TPB = 64;
  const int col_blocks = (n + TPB - 1) / TPB;
  unsigned long long* mask = (unsigned long long*)calloc((size_t)n * col_blocks, sizeof(unsigned long long));
  unsigned long long* remv = (unsigned long long*)calloc(col_blocks, sizeof(unsigned long long));
  if (!mask || !remv) return -1;
  for (int row_start = 0; row_start < col_blocks; ++row_start)
    for (int col_start = 0; col_start < col_blocks; ++col_start) {
      const int row_size = n - row_start * TPB < TPB ? n - row_start * TPB : TPB;
      const int col_size = n - col_start * TPB < TPB ? n - col_start * TPB : TPB;
      for (int t = 0; t < row_size; ++t) {
        const int cur = TPB * row_start + t;
        unsigned long long bits = 0;
        const int start = row_start == col_start ? t + 1 : 0;
        for (int i = start; i < col_size; ++i)
          if (oc_iou(boxes + (size_t)cur * 5, boxes + (size_t)(TPB * col_start + i) * 5) > thresh) bits |= 1ULL << i;
        mask[(size_t)cur * col_blocks + col_start] = bits;
      }
    }
  int num = 0;
  for (int i = 0; i < n; ++i) {
    const int nblock = i / TPB, inblock = i % TPB;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[num++] = i;
      const unsigned long long* p = mask + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
    }
  }
  free(mask);
  free(remv);
  return num;
}

/* dets: n x 5 unsorted, order: score-descending permutation; keep: ORIGINAL indices; predicate ovr >= thresh */
int oc_cpu_nms(const float* dets, const long long* order, int n, float thresh, int* keep) {
  int* suppressed = (int*)calloc(n, sizeof(int));
  float* areas = (float*)malloc(n * sizeof(float));
  if (!suppressed || !areas) return -1;
  for (int i = 0; i < n; ++i)
    areas[i] = (dets[i * 5 + 2] - dets[i * 5 + 0] + 1) * (dets[i * 5 + 3] - dets[i * 5 + 1] + 1);
  int num = 0;
  for (int _i = 0; _i < n; ++_i) {
    const int i = (int)order[_i];
    if (suppressed[i]) continue;
    keep[num++] = i;
    const float ix1 = dets[i * 5], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
    for (int _j = _i + 1; _j < n; ++_j) {
      const int j = (int)order[_j];
      if (suppressed[j]) continue;
      const float xx1 = ix1 >= dets[j * 5] ? ix1 : dets[j * 5];
      const float yy1 = iy1 >= dets[j * 5 + 1] ? iy1 : dets[j * 5 + 1];
      const float xx2 = ix2 <= dets[j * 5 + 2] ? ix2 : dets[j * 5 + 2];
      const float yy2 = iy2 <= dets[j * 5 + 3] ? iy2 : dets[j * 5 + 3];
      const float w = 0.0f >= xx2 - xx1 + 1 ? 0.0f : xx2 - xx1 + 1;
      const float h = 0.0f >= yy2 - yy1 + 1 ? 0.0f : yy2 - yy1 + 1;
      const float inter = w * h;
      const float ovr = inter / (areas[i] + areas[j] - inter);
      if (ovr >= thresh) suppressed[j] = 1;
    }
  }
  free(suppressed);
  free(areas);
  return num;
}
