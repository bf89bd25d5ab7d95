// Detection tail for gfx950: replaces, in a handful of wave64 kernels that never leave
// the device, what the reference does with six 1x1 cuDNN convs, two concats, a 5-kernel
// softmax, a reshape, a D2H copy and the numpy ProposalLayer:
//   cls_score_d / bbox_pred_d 1x1 convs   models/test_different_dilation_template.prototxt:555-644
//   concat / softmax / reshape            :646-683, caffe/src/caffe/layers/softmax_layer.cu:85-120
//   ProposalLayer.forward (TEST phase)    lib/layers/proposal_layer.py:60-220
//   bbox_transform_inv / clip_boxes       lib/utils/bbox_transform.py:33-93
// Compiled with -ffp-contract=off: the box arithmetic keeps numpy's unfused fp32 op order.
#include <math.h>

#include "shf_internal.h"

namespace shf {

static inline unsigned grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

struct TailK {
  const float* feat[8];
  int fstride[8];
  const float* Wt;  // [A][6][Cf]  rows: cls0, cls1, dx, dy, dw, dh
  const float* bt;  // [A][6]
  float* logits;    // [K][A][6]
  int K, A, Cf, w;
  float aw[8], ah[8];  // base anchor widths / heights (x2-x1+1)
  int* counters;       // [1] = overflow flag
};

// np.seterr(over='raise') around exp(dw) * widths (bbox_transform.py:9,52-56): does this delta overflow fp32?
__device__ __forceinline__ bool delta_overflows(float v, float anchor_extent) {
  if (!isfinite(v)) return false;
  const float e = expf(v);
  const float pw = e * anchor_extent;
  return isinf(e) || isinf(pw);
}

// ---- 1x1 cls/reg convs: wave handles two pixels, 32 lanes x float4 per pixel -------------
__global__ __launch_bounds__(256) void tail_logits_kernel(TailK p) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, q = lane & 31;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long npairs = ((long long)p.K + 1) / 2;
  for (int a = 0; a < p.A; ++a) {
    const float* f = p.feat[a];
    const int fs = p.fstride[a];
    for (long long pr = wave; pr < npairs; pr += nwaves) {
      const long long k = pr * 2 + sub;
      float part[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (k < p.K) {
        for (int c0 = 0; c0 < p.Cf; c0 += 128) {
          const float4 x = *(const float4*)(f + (size_t)k * fs + c0 + q * 4);
#pragma unroll
          for (int o = 0; o < 6; ++o) {
            const float4 wv = *(const float4*)(p.Wt + ((size_t)a * 6 + o) * p.Cf + c0 + q * 4);
            part[o] += x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
          }
        }
      }
#pragma unroll
      for (int o = 0; o < 6; ++o) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) part[o] += __shfl_xor(part[o], m, 64);
      }
      if (q == 0 && k < p.K) {
        float* L = p.logits + ((size_t)k * p.A + a) * 6;
        bool of = false;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
          const float v = part[o] + p.bt[a * 6 + o];
          L[o] = v;
          if (o >= 4) of |= delta_overflows(v, o == 4 ? p.aw[a] : p.ah[a]);
        }
        if (of) atomicOr(&p.counters[1], 1);
      }
    }
  }
}

struct DecodeK {
  const float* logits;  // [K][A][6]
  float* rec;           // [K*A][6] bg, fg, x1, y1, x2, y2
  unsigned long long* keys;
  int* counters;        // 0: n candidates, 1: overflow flag ; [4..5] as u64: best key
  int K, A, w;
  float anchors[32];
  int sub_stride[8];
  int feat_stride;
  float im_h, im_w, min_size_scaled, score_thresh;
  float* cls_nchw;   // optional (1,2A,h,w)
  float* bbox_nchw;  // optional (1,4A,h,w)
  int probs_given;   // diagnostics (shf_debug_proposal): L[0..1] already hold bg/fg probabilities
};

// diagnostics: (1,2A,h,w) probabilities + (1,4A,h,w) deltas in the blobs' NCHW layout -> the tail's
// [K][A][6] records, raising the same overflow flag the logits kernel raises
__global__ void tail_inject_kernel(const float* __restrict__ scores, const float* __restrict__ deltas,
                                   float* __restrict__ logits, int K, int A, TailK p) {
  const long long total = (long long)K * A;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < total;
       n += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(n % A);
    const long long k = n / A;
    float* L = logits + n * 6;
    L[0] = scores[(size_t)a * K + k];
    L[1] = scores[(size_t)(A + a) * K + k];
    bool of = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = deltas[(size_t)(a * 4 + j) * K + k];
      L[2 + j] = v;
      if (j >= 2) of |= delta_overflows(v, j == 2 ? p.aw[a] : p.ah[a]);
    }
    if (of) atomicOr(&p.counters[1], 1);
  }
}

__global__ __launch_bounds__(256) void tail_decode_kernel(DecodeK p) {
  const long long total = (long long)p.K * p.A;
  const bool clamp = p.counters[1] != 0;
  const int lane = threadIdx.x & 63;
  const long long step = (long long)gridDim.x * blockDim.x;
  unsigned long long best = 0;
  // every lane runs the same number of iterations so the ballots below are wave-complete
  const long long iters = (total + step - 1) / step;
  long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long it = 0; it < iters; ++it, n += step) {
    bool cand = false;
    unsigned long long key = 0;
    if (n < total) {
      const int a = (int)(n % p.A);
      const int k = (int)(n / p.A);
      const int y = k / p.w, x = k - y * p.w;
      const float* L = p.logits + n * 6;
      const float c0 = L[0], c1 = L[1];
      // SoftmaxLayer::Forward: max, subtract, exp, sum, divide (softmax_layer.cpp:27-60)
      const float m = fmaxf(c0, c1);
      const float e0 = expf(c0 - m), e1 = expf(c1 - m);
      const float sum = e0 + e1;
      const float bg = p.probs_given ? c0 : e0 / sum, fg = p.probs_given ? c1 : e1 / sum;
      float dx = L[2], dy = L[3], dw = L[4], dh = L[5];
      if (p.cls_nchw) {
        p.cls_nchw[(size_t)a * p.K + k] = bg;
        p.cls_nchw[(size_t)(p.A + a) * p.K + k] = fg;
      }
      if (p.bbox_nchw) {
        p.bbox_nchw[(size_t)(a * 4 + 0) * p.K + k] = dx;
        p.bbox_nchw[(size_t)(a * 4 + 1) * p.K + k] = dy;
        p.bbox_nchw[(size_t)(a * 4 + 2) * p.K + k] = dw;
        p.bbox_nchw[(size_t)(a * 4 + 3) * p.K + k] = dh;
      }
      if (clamp) {  // bbox_transform.py:57-63
        if (dw > 50.f) dw = 5.f;
        if (dh > 50.f) dh = 5.f;
      }
      const float sx = (float)(x * p.feat_stride), sy = (float)(y * p.feat_stride);
      const float ax1 = p.anchors[a * 4 + 0] + sx, ay1 = p.anchors[a * 4 + 1] + sy;
      const float ax2 = p.anchors[a * 4 + 2] + sx, ay2 = p.anchors[a * 4 + 3] + sy;
      // bbox_transform_inv (bbox_transform.py:39-75), fp32, unfused
      const float widths = ax2 - ax1 + 1.0f, heights = ay2 - ay1 + 1.0f;
      const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
      const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
      const float pw = expf(dw) * widths, ph = expf(dh) * heights;
      float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
      float x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
      // clip_boxes (bbox_transform.py:80-93)
      const float mw = p.im_w - 1.f, mh = p.im_h - 1.f;
      x1 = fmaxf(fminf(x1, mw), 0.f);
      y1 = fmaxf(fminf(y1, mh), 0.f);
      x2 = fmaxf(fminf(x2, mw), 0.f);
      y2 = fmaxf(fminf(y2, mh), 0.f);
      float* r = p.rec + n * 6;
      r[0] = bg; r[1] = fg; r[2] = x1; r[3] = y1; r[4] = x2; r[5] = y2;
      // anchor subsampling map + _filter_boxes (proposal_layer.py:160-175,231-236)
      const int ss = p.sub_stride[a];
      bool valid = (y % ss == 0) && (x % ss == 0);
      const float ws = x2 - x1 + 1.f, hs = y2 - y1 + 1.f;
      valid = valid && (ws >= p.min_size_scaled) && (hs >= p.min_size_scaled);
      if (valid) {
        key = ((unsigned long long)__float_as_uint(fg) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
        best = key > best ? key : best;
        cand = fg >= p.score_thresh;
      }
    }
    // wave-aggregated append of candidate keys
    const unsigned long long bal = __ballot(cand);
    if (bal) {
      int base = 0;
      const int leader = __ffsll((long long)bal) - 1;
      if (lane == leader) base = atomicAdd(&p.counters[0], __popcll(bal));
      base = __shfl(base, leader, 64);
      if (cand) p.keys[base + __popcll(bal & ((1ull << lane) - 1ull))] = key;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = __shfl_xor(best, m, 64);
    best = o > best ? o : best;
  }
  if (lane == 0 && best) atomicMax((unsigned long long*)(p.counters + 4), best);
}

// after the sort: R = min(C, topN), or the single best valid anchor when nothing reaches the
// threshold (proposal_layer.py:182-188)
__global__ void tail_finalize_kernel(unsigned long long* keys, int* counters, int topN) {
  const int C = counters[0];
  const unsigned long long best = *(unsigned long long*)(counters + 4);
  int R;
  if (C > 0) {
    R = (topN > 0 && C > topN) ? topN : C;
  } else if (best) {
    keys[0] = best;
    R = 1;
  } else {
    R = 0;
  }
  counters[2] = R;
}

__global__ void tail_gather_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ rec,
                                   const int* __restrict__ counters, float* __restrict__ boxes5,
                                   float* __restrict__ probs2) {
  const int R = counters[2];
  if (R == 0 && blockIdx.x == 0 && threadIdx.x == 0) {  // dummy roi, proposal_layer.py:207-208
    boxes5[0] = 0.f; boxes5[1] = 0.f; boxes5[2] = 0.f; boxes5[3] = 16.f; boxes5[4] = 16.f;
  }
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    const unsigned n = 0xFFFFFFFFu - (unsigned)(keys[r] & 0xFFFFFFFFull);
    const float* q = rec + (size_t)n * 6;
    boxes5[r * 5 + 0] = 0.f;
    boxes5[r * 5 + 1] = q[2];
    boxes5[r * 5 + 2] = q[3];
    boxes5[r * 5 + 3] = q[4];
    boxes5[r * 5 + 4] = q[5];
    probs2[r * 2 + 0] = q[0];
    probs2[r * 2 + 1] = q[1];
  }
}

// ---------------------------------------------------------------------------
// bitonic sort, descending, u64 keys, element count on the device.
// Chunks of 16384 keys are sorted / merged inside one CU's LDS (128 KiB of the 160).
// ---------------------------------------------------------------------------
constexpr int SORT_CH = 16384;

__device__ __forceinline__ unsigned next_pow2(unsigned v) {
  unsigned p = 1;
  while (p < v) p <<= 1;
  return p;
}

// all (k,j) stages with k <= chunk: full sort of each chunk in its network direction
__global__ __launch_bounds__(1024) void bitonic_local_sort_kernel(unsigned long long* keys, const int* n_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const unsigned n = (unsigned)max(*n_dev, 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  const unsigned m = npad < (unsigned)SORT_CH ? npad : (unsigned)SORT_CH;
  const unsigned start = blockIdx.x * (unsigned)SORT_CH;
  if (start >= npad) return;
  for (unsigned t = threadIdx.x; t < m; t += 1024) sk[t] = (start + t < n) ? keys[start + t] : 0ull;
  __syncthreads();
  for (unsigned k = 2; k <= m; k <<= 1) {
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned t = threadIdx.x; t < (m >> 1); t += 1024) {
        const unsigned i = ((t / j) * 2 * j) + (t % j);
        const unsigned l = i + j;
        const bool desc = (((start + i) & k) == 0);  // overall descending order
        const unsigned long long a = sk[i], b = sk[l];
        if ((a < b) == desc) { sk[i] = b; sk[l] = a; }
      }
      __syncthreads();
    }
  }
  for (unsigned t = threadIdx.x; t < m; t += 1024) keys[start + t] = sk[t];
}

// one global compare-exchange stage (k, j) with j >= chunk
__global__ void bitonic_global_step_kernel(unsigned long long* keys, const int* n_dev, unsigned k, unsigned j) {
  const unsigned n = (unsigned)max(*n_dev, 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  if (k > npad) return;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < (npad >> 1); t += gridDim.x * blockDim.x) {
    const unsigned i = ((t / j) * 2 * j) + (t % j);
    const unsigned l = i + j;
    const bool desc = ((i & k) == 0);
    const unsigned long long a = keys[i], b = keys[l];
    if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
  }
}

// remaining stages j = chunk/2 .. 1 of merge level k (> chunk), inside LDS
__global__ __launch_bounds__(1024) void bitonic_local_merge_kernel(unsigned long long* keys, const int* n_dev,
                                                                   unsigned k) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const unsigned n = (unsigned)max(*n_dev, 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  if (k > npad) return;
  const unsigned start = blockIdx.x * (unsigned)SORT_CH;
  if (start >= npad) return;
  for (unsigned t = threadIdx.x; t < (unsigned)SORT_CH; t += 1024) sk[t] = keys[start + t];
  __syncthreads();
  const bool desc = ((start & k) == 0);
  for (unsigned j = SORT_CH >> 1; j > 0; j >>= 1) {
    for (unsigned t = threadIdx.x; t < (unsigned)(SORT_CH >> 1); t += 1024) {
      const unsigned i = ((t / j) * 2 * j) + (t % j);
      const unsigned l = i + j;
      const unsigned long long a = sk[i], b = sk[l];
      if ((a < b) == desc) { sk[i] = b; sk[l] = a; }
    }
    __syncthreads();
  }
  for (unsigned t = threadIdx.x; t < (unsigned)SORT_CH; t += 1024) keys[start + t] = sk[t];
}

static bool g_sort_attr_done = false;

int launch_sort_desc_u64(unsigned long long* keys, const int* n_dev, size_t n_max, hipStream_t s) {
  if (!g_sort_attr_done) {
    SHF_HIP_OK(hipFuncSetAttribute((const void*)bitonic_local_sort_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8));
    SHF_HIP_OK(hipFuncSetAttribute((const void*)bitonic_local_merge_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8));
    g_sort_attr_done = true;
  }
  size_t npad = 1;
  while (npad < n_max) npad <<= 1;
  const unsigned nchunks = (unsigned)((npad + SORT_CH - 1) / SORT_CH);
  // keys beyond n inside [n, npad) are treated as 0 by the local sort; for npad > chunk the
  // caller's buffer must hold npad entries (launch_tail / merge allocate pow2 capacity).
  hipLaunchKernelGGL(bitonic_local_sort_kernel, dim3(nchunks), dim3(1024), SORT_CH * 8, s, keys, n_dev);
  for (size_t k = (size_t)SORT_CH * 2; k <= npad; k <<= 1) {
    for (size_t j = k >> 1; j >= (size_t)SORT_CH; j >>= 1)
      hipLaunchKernelGGL(bitonic_global_step_kernel, dim3(grid_for((long long)(npad >> 1))), dim3(256), 0, s, keys,
                         n_dev, (unsigned)k, (unsigned)j);
    hipLaunchKernelGGL(bitonic_local_merge_kernel, dim3(nchunks), dim3(1024), SORT_CH * 8, s, keys, n_dev,
                       (unsigned)k);
  }
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
int launch_tail(const TailArgs& a, TailWork& ws, float* out_boxes5, float* out_probs2, hipStream_t s,
                hipEvent_t after_logits, int phase) {
  const int K = a.h * a.w;
  const long long total = (long long)K * a.A;
  if ((size_t)total > ws.cap_anchors) { set_error("tail: workspace too small"); return -1; }
  if (a.Cf % 128) { set_error("tail: head feature width must be a multiple of 128"); return -1; }
  if (phase != 2) {
    SHF_HIP_OK(hipMemsetAsync(ws.counters, 0, 8 * sizeof(int), s));
    TailK lk;
    for (int i = 0; i < a.A; ++i) {
      const View& f = a.feat[a.heads == 1 ? 0 : i];
      lk.feat[i] = f.p + f.coff;
      lk.fstride[i] = f.cstride;
      lk.aw[i] = a.anchors[i * 4 + 2] - a.anchors[i * 4 + 0] + 1.0f;
      lk.ah[i] = a.anchors[i * 4 + 3] - a.anchors[i * 4 + 1] + 1.0f;
    }
    lk.Wt = a.wcls[0];  // combined [A][6][Cf] matrix prepared by the net (see net.cpp: build_tail_weights)
    lk.bt = a.bcls[0];
    lk.logits = ws.logits;
    lk.K = K; lk.A = a.A; lk.Cf = a.Cf; lk.w = a.w;
    lk.counters = ws.counters;
    hipLaunchKernelGGL(tail_logits_kernel, dim3(grid_for(((long long)K + 1) / 2 * 64)), dim3(256), 0, s, lk);
    // from here on the tail only touches its own workspace: the head feature maps may be overwritten
    if (after_logits) SHF_HIP_OK(hipEventRecord(after_logits, s));
  }
  if (phase == 1) return 0;

  DecodeK dk;
  dk.logits = ws.logits; dk.rec = ws.rec; dk.keys = ws.keys; dk.counters = ws.counters;
  dk.K = K; dk.A = a.A; dk.w = a.w;
  for (int i = 0; i < a.A * 4; ++i) dk.anchors[i] = a.anchors[i];
  for (int i = 0; i < a.A; ++i) dk.sub_stride[i] = a.sub_stride[i] < 1 ? 1 : a.sub_stride[i];
  dk.feat_stride = a.feat_stride;
  dk.im_h = a.im_h; dk.im_w = a.im_w;
  dk.min_size_scaled = a.min_size * a.im_scale;
  dk.score_thresh = a.score_thresh;
  dk.cls_nchw = a.cls_prob_reshape_nchw;
  dk.bbox_nchw = a.bbox_pred_nchw;
  dk.probs_given = a.probs_given;
  hipLaunchKernelGGL(tail_decode_kernel, dim3(grid_for(total)), dim3(256), 0, s, dk);
  if (launch_sort_desc_u64(ws.keys, ws.counters, (size_t)total, s)) return -1;
  hipLaunchKernelGGL(tail_finalize_kernel, dim3(1), dim3(1), 0, s, ws.keys, ws.counters, a.pre_nms_topN);
  const long long rmax = (a.pre_nms_topN > 0 && a.pre_nms_topN < total) ? a.pre_nms_topN : total;
  hipLaunchKernelGGL(tail_gather_kernel, dim3(grid_for(rmax)), dim3(256), 0, s, ws.keys, ws.rec, ws.counters,
                     out_boxes5, out_probs2);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_tail_inject(const TailArgs& a, TailWork& ws, const float* scores_nchw, const float* deltas_nchw,
                       hipStream_t s) {
  const int K = a.h * a.w;
  const long long total = (long long)K * a.A;
  if ((size_t)total > ws.cap_anchors) { set_error("tail: workspace too small"); return -1; }
  SHF_HIP_OK(hipMemsetAsync(ws.counters, 0, 8 * sizeof(int), s));
  TailK lk = {};
  for (int i = 0; i < a.A; ++i) {
    lk.aw[i] = a.anchors[i * 4 + 2] - a.anchors[i * 4 + 0] + 1.0f;
    lk.ah[i] = a.anchors[i * 4 + 3] - a.anchors[i * 4 + 1] + 1.0f;
  }
  lk.counters = ws.counters;
  hipLaunchKernelGGL(tail_inject_kernel, dim3(grid_for(total)), dim3(256), 0, s, scores_nchw, deltas_nchw, ws.logits,
                     K, a.A, lk);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// forward_net post-processing + the >thresh cut of detect() on the device
// (lib/test.py:52-54 flip fix, :59-66 unscale, :163-167 threshold).  Rows of a unit are
// score-descending, so the survivors are a prefix and land at base+r: the image list is
// in the reference's concatenation order.
// counters: [0] = base read by unit u when u is even / written when odd, [1] the other
// ---------------------------------------------------------------------------
__global__ void append_dets_kernel(const float* __restrict__ boxes5, const float* __restrict__ probs2,
                                   const int* __restrict__ R_dev, float im_w, float im_scale, int flip,
                                   float thresh, int unit, float* __restrict__ img_dets5,
                                   unsigned long long* __restrict__ img_keys, int* img_count, int img_cap) {
  const int R = *R_dev;
  const int base = img_count[unit & 1];
  int* next = &img_count[(unit + 1) & 1];
  if (blockIdx.x == 0 && threadIdx.x == 0 && (R == 0 || !(probs2[1] > thresh))) *next = base;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
    const float fg = probs2[r * 2 + 1];
    if (!(fg > thresh)) continue;
    if (r == R - 1 || !(probs2[(r + 1) * 2 + 1] > thresh)) *next = min(base + r + 1, img_cap);
    const int pos = base + r;
    if (pos >= img_cap) continue;
    float x1 = boxes5[r * 5 + 1], y1 = boxes5[r * 5 + 2], x2 = boxes5[r * 5 + 3], y2 = boxes5[r * 5 + 4];
    if (flip) {  // boxes[:, [1,3]] = w - boxes[:, [3,1]]
      const float nx1 = im_w - x2, nx2 = im_w - x1;
      x1 = nx1;
      x2 = nx2;
    }
    float* d = img_dets5 + (size_t)pos * 5;
    d[0] = x1 / im_scale; d[1] = y1 / im_scale; d[2] = x2 / im_scale; d[3] = y2 / im_scale; d[4] = fg;
    img_keys[pos] = ((unsigned long long)__float_as_uint(fg) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)pos);
  }
}

int launch_append_dets(const float* boxes5, const float* probs2, const int* R_dev, int r_max, float im_w,
                       float im_scale, int flip, float thresh, int unit, float* img_dets5,
                       unsigned long long* img_keys, int* img_count, int img_cap, hipStream_t s) {
  hipLaunchKernelGGL(append_dets_kernel, dim3(grid_for(r_max < 1 ? 1 : r_max)), dim3(256), 0, s, boxes5, probs2,
                     R_dev, im_w, im_scale, flip, thresh, unit, img_dets5, img_keys, img_count, img_cap);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
