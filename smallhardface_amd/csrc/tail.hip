// Detection tail for gfx950: replaces, in a handful of wave64 kernels that never leave
// the device, what the reference does with six 1x1 cuDNN convs, two concats, a 5-kernel
// softmax, a reshape, a D2H copy and the numpy ProposalLayer:
//   cls_score_d / bbox_pred_d 1x1 convs   models/test_different_dilation_template.prototxt:555-644
//   concat / softmax / reshape            :646-683, caffe/src/caffe/layers/softmax_layer.cu:85-120
//   ProposalLayer.forward (TEST phase)    lib/layers/proposal_layer.py:60-220
//   bbox_transform_inv / clip_boxes       lib/utils/bbox_transform.py:33-93
// Compiled with -ffp-contract=off: the box arithmetic keeps numpy's unfused fp32 op order.
#include <math.h>

#include <algorithm>

#include "shf_internal.h"

namespace shf {

static inline unsigned grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

constexpr int TG = 16;  // members per grouped launch (the units of an image's pyramid)

// which member does block `b` belong to?  blk_start[q] = first block of member q (unused entries INT_MAX)
__device__ __forceinline__ int tail_find_member(const int* blk_start, int b) {
  int mi = 0;
#pragma unroll
  for (int q = 1; q < TG; ++q) mi += (b >= blk_start[q]) ? 1 : 0;
  return mi;
}

// np.seterr(over='raise') around exp(dw) * widths (bbox_transform.py:9,52-56): does this delta overflow fp32?
__device__ __forceinline__ bool delta_overflows(float v, float anchor_extent) {
  if (!isfinite(v)) return false;
  const float e = expf(v);
  const float pw = e * anchor_extent;
  return isinf(e) || isinf(pw);
}

// ---- every kernel of the tail takes a GROUP of independent units (the 10 (level, flip) units of an image): one
//      launch per stage instead of one per unit and stage (~100 five-microsecond launches per image) --------------
struct TailResetK {
  int* counters[TG];
  unsigned* amax[TG];   // activation-exponent slots of the member lanes (null: none): consumed by this pass's convolutions,
  int n_amax[TG];       // zeroed here for the next pass
  int n;
};
__global__ void tail_reset_kernel(TailResetK p) {
  const int m = threadIdx.x >> 3, j = threadIdx.x & 7;
  if (m < p.n) {
    p.counters[m][j] = 0;
    if (p.amax[m])
      for (int i = j; i < p.n_amax[m]; i += 8) p.amax[m][i] = 0u;
  }
}

struct TailGM {  // per member
  const float* feat[8];
  int fstride[8];
  float* logits;    // [K][A][6]
  int* counters;    // [1] = overflow flag
  int K, w;
};
struct TailGK {
  const float* Wt;  // [A][6][Cf]  rows: cls0, cls1, dx, dy, dw, dh
  const float* bt;  // [A][6]
  int A, Cf;
  float aw[8], ah[8];  // base anchor widths / heights (x2-x1+1)
  int blk_start[TG + 1];
  TailGM m[TG];
};

// ---- 1x1 cls/reg convs: wave handles two pixels, 32 lanes x float4 per pixel -------------
__global__ __launch_bounds__(256) void tail_logits_kernel(TailGK g) {
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  const TailGM& p = g.m[mi];
  const int lb = blockIdx.x - g.blk_start[mi], lgrid = g.blk_start[mi + 1] - g.blk_start[mi];
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, q = lane & 31;
  const long long wave = ((long long)lb * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)lgrid * blockDim.x) >> 6;
  const long long npairs = ((long long)p.K + 1) / 2;
  for (int a = 0; a < g.A; ++a) {
    const float* f = p.feat[a];
    const int fs = p.fstride[a];
    for (long long pr = wave; pr < npairs; pr += nwaves) {
      const long long k = pr * 2 + sub;
      float part[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (k < p.K) {
        for (int c0 = 0; c0 < g.Cf; c0 += 128) {
          const float4 x = *(const float4*)(f + (size_t)k * fs + c0 + q * 4);
#pragma unroll
          for (int o = 0; o < 6; ++o) {
            const float4 wv = *(const float4*)(g.Wt + ((size_t)a * 6 + o) * g.Cf + c0 + q * 4);
            part[o] += x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
          }
        }
      }
      // the xor butterfly 16, 8, 4, 2, 1 over the 32 lanes of a pixel -- only the first step through the LDS crossbar: after
      // step m the lanes i and i ^ m hold the same bits (a + b == b + a), so the partner of step 8 / 4 may as well be the
      // lane 8 / 4 places round the 16-lane row (DPP row_ror), and steps 2 / 1 are quad permutes: the same tree, the same bits
#pragma unroll
      for (int o = 0; o < 6; ++o) {
        float v = part[o];
        v += __shfl_xor(v, 16, 64);
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
        part[o] = v;
      }
      if (q == 0 && k < p.K) {
        float* L = p.logits + ((size_t)k * g.A + a) * 6;
        bool of = false;
#pragma unroll
        for (int o = 0; o < 6; ++o) {
          const float v = part[o] + g.bt[a * 6 + o];
          L[o] = v;
          if (o >= 4) of |= delta_overflows(v, o == 4 ? g.aw[a] : g.ah[a]);
        }
        if (of) atomicOr(&p.counters[1], 1);
      }
    }
  }
}

// diagnostics: (1,2A,h,w) probabilities + (1,4A,h,w) deltas in the blobs' NCHW layout -> the tail's
// [K][A][6] records, raising the same overflow flag the logits kernel raises
__global__ void tail_inject_kernel(const float* __restrict__ scores, const float* __restrict__ deltas,
                                   float* __restrict__ logits, int K, int A, TailGK g, int* counters) {
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
      if (j >= 2) of |= delta_overflows(v, j == 2 ? g.aw[a] : g.ah[a]);
    }
    if (of) atomicOr(&counters[1], 1);
  }
}

struct DecodeGM {  // per member
  const float* logits;  // [K][A][6]
  float* rec;           // [K*A][6] bg, fg, x1, y1, x2, y2
  unsigned long long* keys;
  int* counters;        // 0: n candidates, 1: overflow flag ; [4..5] as u64: best key
  int K, w;
  float im_h, im_w, min_size_scaled;
  float* cls_nchw;   // optional (1,2A,h,w)
  float* bbox_nchw;  // optional (1,4A,h,w)
};
struct DecodeGK {
  int A;
  float anchors[32];
  int sub_stride[8];
  int feat_stride;
  float score_thresh;
  int probs_given;   // diagnostics (shf_debug_proposal): L[0..1] already hold bg/fg probabilities
  int blk_start[TG + 1];
  DecodeGM m[TG];
};

__global__ __launch_bounds__(256) void tail_decode_kernel(DecodeGK g) {
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  const DecodeGM& p = g.m[mi];
  const int lb = blockIdx.x - g.blk_start[mi], lgrid = g.blk_start[mi + 1] - g.blk_start[mi];
  const long long total = (long long)p.K * g.A;
  const bool clamp = p.counters[1] != 0;
  const int lane = threadIdx.x & 63;
  const long long step = (long long)lgrid * blockDim.x;
  unsigned long long best = 0;
  // every lane runs the same number of iterations so the ballots below are wave-complete
  const long long iters = (total + step - 1) / step;
  long long n = (long long)lb * blockDim.x + threadIdx.x;
  for (long long it = 0; it < iters; ++it, n += step) {
    bool cand = false;
    unsigned long long key = 0;
    if (n < total) {
      const int a = (int)(n % g.A);
      const int k = (int)(n / g.A);
      const int y = k / p.w, x = k - y * p.w;
      const float* L = p.logits + n * 6;
      const float c0 = L[0], c1 = L[1];
      // SoftmaxLayer::Forward: max, subtract, exp, sum, divide (softmax_layer.cpp:27-60)
      const float m = fmaxf(c0, c1);
      const float e0 = expf(c0 - m), e1 = expf(c1 - m);
      const float sum = e0 + e1;
      const float bg = g.probs_given ? c0 : e0 / sum, fg = g.probs_given ? c1 : e1 / sum;
      float dx = L[2], dy = L[3], dw = L[4], dh = L[5];
      if (p.cls_nchw) {
        p.cls_nchw[(size_t)a * p.K + k] = bg;
        p.cls_nchw[(size_t)(g.A + a) * p.K + k] = fg;
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
      const float sx = (float)(x * g.feat_stride), sy = (float)(y * g.feat_stride);
      const float ax1 = g.anchors[a * 4 + 0] + sx, ay1 = g.anchors[a * 4 + 1] + sy;
      const float ax2 = g.anchors[a * 4 + 2] + sx, ay2 = g.anchors[a * 4 + 3] + sy;
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
      const int ss = g.sub_stride[a];
      bool valid = (y % ss == 0) && (x % ss == 0);
      const float ws = x2 - x1 + 1.f, hs = y2 - y1 + 1.f;
      valid = valid && (ws >= p.min_size_scaled) && (hs >= p.min_size_scaled);
      if (valid) {
        key = ((unsigned long long)__float_as_uint(fg) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
        best = key > best ? key : best;
        cand = fg >= g.score_thresh;
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
__device__ __forceinline__ int tail_rows(const int* counters, int topN, unsigned long long* best_out) {
  const int C = counters[0];
  const unsigned long long best = *(const unsigned long long*)(counters + 4);
  if (best_out) *best_out = best;
  if (C > 0) return (topN > 0 && C > topN) ? topN : C;
  return best ? 1 : 0;
}

struct GatherGM {
  const unsigned long long* keys;
  const float* rec;
  int* counters;
  float* boxes5;
  float* probs2;
};
struct GatherGK {
  int topN;
  int blk_start[TG + 1];
  GatherGM m[TG];
};
// finalize + gather in one: every block derives R from the counters (block 0 of the member publishes it in
// counters[2] for the host / the append stage)
__global__ void tail_gather_kernel(GatherGK g) {
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  const GatherGM& p = g.m[mi];
  const int lb = blockIdx.x - g.blk_start[mi], lgrid = g.blk_start[mi + 1] - g.blk_start[mi];
  unsigned long long best;
  const int R = tail_rows(p.counters, g.topN, &best);
  const bool only_best = p.counters[0] <= 0;  // nothing reached the threshold: the row is the best valid anchor
  if (lb == 0 && threadIdx.x == 0) {
    p.counters[2] = R;
    if (R == 0) {  // dummy roi, proposal_layer.py:207-208
      p.boxes5[0] = 0.f; p.boxes5[1] = 0.f; p.boxes5[2] = 0.f; p.boxes5[3] = 16.f; p.boxes5[4] = 16.f;
    }
  }
  for (int r = lb * blockDim.x + threadIdx.x; r < R; r += lgrid * blockDim.x) {
    const unsigned long long key = only_best ? best : p.keys[r];
    const unsigned n = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const float* q = p.rec + (size_t)n * 6;
    p.boxes5[r * 5 + 0] = 0.f;
    p.boxes5[r * 5 + 1] = q[2];
    p.boxes5[r * 5 + 2] = q[3];
    p.boxes5[r * 5 + 3] = q[4];
    p.boxes5[r * 5 + 4] = q[5];
    p.probs2[r * 2 + 0] = q[0];
    p.probs2[r * 2 + 1] = q[1];
  }
}

// ---------------------------------------------------------------------------
// bitonic sort, descending, u64 keys, element count on the device; a GROUP of independent arrays per launch.
// Chunks of 16384 keys are sorted / merged inside one CU's LDS (128 KiB of the 160).
// ---------------------------------------------------------------------------
constexpr int SORT_CH = 16384;

__device__ __forceinline__ unsigned next_pow2(unsigned v) {
  unsigned p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct SortGK {
  unsigned long long* keys[TG];
  const int* n_dev[TG];
  int blk_start[TG + 1];  // in chunks (local kernels) or in 256-thread blocks (global step)
};

// all (k,j) stages with k <= chunk: full sort of each chunk in its network direction
__global__ __launch_bounds__(1024) void bitonic_local_sort_kernel(SortGK g) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  unsigned long long* keys = g.keys[mi];
  const unsigned n = (unsigned)max(*g.n_dev[mi], 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  const unsigned m = npad < (unsigned)SORT_CH ? npad : (unsigned)SORT_CH;
  const unsigned start = (unsigned)(blockIdx.x - g.blk_start[mi]) * (unsigned)SORT_CH;
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
__global__ void bitonic_global_step_kernel(SortGK g, unsigned k, unsigned j) {
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  unsigned long long* keys = g.keys[mi];
  const unsigned n = (unsigned)max(*g.n_dev[mi], 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  if (k > npad) return;
  const unsigned lb = blockIdx.x - g.blk_start[mi], lgrid = g.blk_start[mi + 1] - g.blk_start[mi];
  for (unsigned t = lb * blockDim.x + threadIdx.x; t < (npad >> 1); t += lgrid * blockDim.x) {
    const unsigned i = ((t / j) * 2 * j) + (t % j);
    const unsigned l = i + j;
    const bool desc = ((i & k) == 0);
    const unsigned long long a = keys[i], b = keys[l];
    if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
  }
}

// remaining stages j = chunk/2 .. 1 of merge level k (> chunk), inside LDS
__global__ __launch_bounds__(1024) void bitonic_local_merge_kernel(SortGK g, unsigned k) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  unsigned long long* keys = g.keys[mi];
  const unsigned n = (unsigned)max(*g.n_dev[mi], 0);
  const unsigned npad = next_pow2(n < 1 ? 1 : n);
  if (k > npad) return;
  const unsigned start = (unsigned)(blockIdx.x - g.blk_start[mi]) * (unsigned)SORT_CH;
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

// keys[m] beyond n inside [n, npad) are treated as 0 by the local sort; for npad > chunk the caller's
// buffers must hold npad entries (the tail workspace / merge context allocate pow2 capacity).
int launch_sort_desc_u64_group(unsigned long long* const* keys, const int* const* n_dev, const size_t* n_max, int n,
                               hipStream_t s) {
  if (n < 1 || n > TG) { set_error("sort group: 1..16 arrays"); return -1; }
  if (!g_sort_attr_done) {
    SHF_HIP_OK(hipFuncSetAttribute((const void*)bitonic_local_sort_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8));
    SHF_HIP_OK(hipFuncSetAttribute((const void*)bitonic_local_merge_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8));
    g_sort_attr_done = true;
  }
  SortGK loc, glo;
  size_t npad_max = 1;
  int nchunks = 0, gblocks = 0;
  for (int q = 0; q <= TG; ++q) loc.blk_start[q] = glo.blk_start[q] = 0x7fffffff;
  for (int m = 0; m < n; ++m) {
    size_t npad = 1;
    while (npad < n_max[m]) npad <<= 1;
    npad_max = std::max(npad_max, npad);
    loc.keys[m] = glo.keys[m] = keys[m];
    loc.n_dev[m] = glo.n_dev[m] = n_dev[m];
    loc.blk_start[m] = nchunks;
    glo.blk_start[m] = gblocks;
    nchunks += (int)((npad + SORT_CH - 1) / SORT_CH);
    gblocks += (int)grid_for((long long)(npad >> 1));
  }
  loc.blk_start[n] = nchunks;
  glo.blk_start[n] = gblocks;
  hipLaunchKernelGGL(bitonic_local_sort_kernel, dim3(nchunks), dim3(1024), SORT_CH * 8, s, loc);
  for (size_t k = (size_t)SORT_CH * 2; k <= npad_max; k <<= 1) {
    for (size_t j = k >> 1; j >= (size_t)SORT_CH; j >>= 1)
      hipLaunchKernelGGL(bitonic_global_step_kernel, dim3(gblocks), dim3(256), 0, s, glo, (unsigned)k, (unsigned)j);
    hipLaunchKernelGGL(bitonic_local_merge_kernel, dim3(nchunks), dim3(1024), SORT_CH * 8, s, loc, (unsigned)k);
  }
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_sort_desc_u64(unsigned long long* keys, const int* n_dev, size_t n_max, hipStream_t s) {
  return launch_sort_desc_u64_group(&keys, &n_dev, &n_max, 1, s);
}

// ---------------------------------------------------------------------------
static void fill_logits_shared(const TailArgs& a, TailGK& lk) {
  lk.Wt = a.wcls[0];  // combined [A][6][Cf] matrix prepared by the net (see net_graph.cpp: build_tail_weights)
  lk.bt = a.bcls[0];
  lk.A = a.A; lk.Cf = a.Cf;
  for (int i = 0; i < a.A; ++i) {
    lk.aw[i] = a.anchors[i * 4 + 2] - a.anchors[i * 4 + 0] + 1.0f;
    lk.ah[i] = a.anchors[i * 4 + 3] - a.anchors[i * 4 + 1] + 1.0f;
  }
}

// A group of units through the proposal stage.  phase 0: everything; 1: reset + logits (the kernels that read the
// head feature maps: `after_logits` is recorded behind them); 2: decode -> sort -> gather.
int launch_tail_group(const TailArgs* as, TailWork* const* wss, float* const* out_boxes5, float* const* out_probs2,
                      int n, hipStream_t s, hipEvent_t after_logits, int phase) {
  if (n < 1 || n > TG) { set_error("tail group: 1..16 units"); return -1; }
  const TailArgs& a0 = as[0];
  if (a0.A > 8) { set_error("tail: at most 8 anchors per cell"); return -1; }
  for (int m = 0; m < n; ++m) {
    const long long total = (long long)as[m].h * as[m].w * as[m].A;
    if ((size_t)total > wss[m]->cap_anchors) { set_error("tail: workspace too small"); return -1; }
    if (as[m].Cf % 128) { set_error("tail: head feature width must be a multiple of 128"); return -1; }
    if (as[m].A != a0.A || as[m].Cf != a0.Cf || as[m].wcls[0] != a0.wcls[0]) {
      set_error("tail group: members must share the proposal layer");
      return -1;
    }
  }
  if (phase != 2) {
    TailResetK rk;
    rk.n = n;
    for (int m = 0; m < n; ++m) {
      rk.counters[m] = wss[m]->counters;
      rk.amax[m] = wss[m]->amax;
      rk.n_amax[m] = wss[m]->n_amax;
    }
    hipLaunchKernelGGL(tail_reset_kernel, dim3(1), dim3(TG * 8), 0, s, rk);
    TailGK lk;
    fill_logits_shared(a0, lk);
    int blocks = 0;
    for (int q = 0; q <= TG; ++q) lk.blk_start[q] = 0x7fffffff;
    for (int m = 0; m < n; ++m) {
      const TailArgs& a = as[m];
      TailGM& g = lk.m[m];
      for (int i = 0; i < a.A; ++i) {
        const View& f = a.feat[a.heads == 1 ? 0 : i];
        g.feat[i] = f.p + f.coff;
        g.fstride[i] = f.cstride;
      }
      g.logits = wss[m]->logits;
      g.counters = wss[m]->counters;
      g.K = a.h * a.w;
      g.w = a.w;
      lk.blk_start[m] = blocks;
      blocks += (int)grid_for(((long long)g.K + 1) / 2 * 64);
    }
    lk.blk_start[n] = blocks;
    hipLaunchKernelGGL(tail_logits_kernel, dim3(blocks), dim3(256), 0, s, lk);
    // from here on the tails only touch their own workspaces: the head feature maps may be overwritten
    if (after_logits) SHF_HIP_OK(hipEventRecord(after_logits, s));
  }
  if (phase == 1) {
    SHF_HIP_OK(hipGetLastError());
    return 0;
  }
  DecodeGK dk;
  dk.A = a0.A;
  for (int i = 0; i < a0.A * 4; ++i) dk.anchors[i] = a0.anchors[i];
  for (int i = 0; i < a0.A; ++i) dk.sub_stride[i] = a0.sub_stride[i] < 1 ? 1 : a0.sub_stride[i];
  dk.feat_stride = a0.feat_stride;
  dk.score_thresh = a0.score_thresh;
  dk.probs_given = a0.probs_given;
  GatherGK gk;
  gk.topN = a0.pre_nms_topN;
  unsigned long long* keys[TG];
  const int* n_dev[TG];
  size_t n_max[TG];
  int dblocks = 0, gblocks = 0;
  for (int q = 0; q <= TG; ++q) dk.blk_start[q] = gk.blk_start[q] = 0x7fffffff;
  for (int m = 0; m < n; ++m) {
    const TailArgs& a = as[m];
    const long long total = (long long)a.h * a.w * a.A;
    DecodeGM& d = dk.m[m];
    d.logits = wss[m]->logits; d.rec = wss[m]->rec; d.keys = wss[m]->keys; d.counters = wss[m]->counters;
    d.K = a.h * a.w; d.w = a.w;
    d.im_h = a.im_h; d.im_w = a.im_w;
    d.min_size_scaled = a.min_size * a.im_scale;
    d.cls_nchw = a.cls_prob_reshape_nchw;
    d.bbox_nchw = a.bbox_pred_nchw;
    dk.blk_start[m] = dblocks;
    dblocks += (int)grid_for(total);
    GatherGM& q = gk.m[m];
    q.keys = wss[m]->keys; q.rec = wss[m]->rec; q.counters = wss[m]->counters;
    q.boxes5 = out_boxes5[m]; q.probs2 = out_probs2[m];
    const long long rmax = (a.pre_nms_topN > 0 && a.pre_nms_topN < total) ? a.pre_nms_topN : total;
    gk.blk_start[m] = gblocks;
    gblocks += (int)grid_for(rmax);
    keys[m] = wss[m]->keys;
    n_dev[m] = wss[m]->counters;
    n_max[m] = (size_t)total;
  }
  dk.blk_start[n] = dblocks;
  gk.blk_start[n] = gblocks;
  hipLaunchKernelGGL(tail_decode_kernel, dim3(dblocks), dim3(256), 0, s, dk);
  if (launch_sort_desc_u64_group(keys, n_dev, n_max, n, s)) return -1;
  hipLaunchKernelGGL(tail_gather_kernel, dim3(gblocks), dim3(256), 0, s, gk);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

int launch_tail(const TailArgs& a, TailWork& ws, float* out_boxes5, float* out_probs2, hipStream_t s,
                hipEvent_t after_logits, int phase) {
  TailWork* w = &ws;
  return launch_tail_group(&a, &w, &out_boxes5, &out_probs2, 1, s, after_logits, phase);
}

int launch_tail_inject(const TailArgs& a, TailWork& ws, const float* scores_nchw, const float* deltas_nchw,
                       hipStream_t s) {
  const int K = a.h * a.w;
  const long long total = (long long)K * a.A;
  if ((size_t)total > ws.cap_anchors) { set_error("tail: workspace too small"); return -1; }
  SHF_HIP_OK(hipMemsetAsync(ws.counters, 0, 8 * sizeof(int), s));
  TailGK lk = {};
  fill_logits_shared(a, lk);
  hipLaunchKernelGGL(tail_inject_kernel, dim3(grid_for(total)), dim3(256), 0, s, scores_nchw, deltas_nchw, ws.logits,
                     K, a.A, lk, ws.counters);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// forward_net post-processing + the >thresh cut of detect() on the device
// (lib/test.py:52-54 flip fix, :59-66 unscale, :163-167 threshold), for a GROUP of units in one launch.  Rows of a
// unit are score-descending, so its survivors are a prefix (found by bisection) and unit u lands behind the
// survivors of units 0..u-1: the image list is in the reference's concatenation order.
// dst->count: slot (pass & 1) holds the list length before this pass, slot ((pass + 1) & 1) receives the length
// after it (two slots: no block of this launch reads what another one writes).
// ---------------------------------------------------------------------------
struct AppendGM {
  const float* boxes5;
  const float* probs2;
  const int* counters;  // the unit's tail counters (rows R derived like the gather stage does)
  float im_w, im_scale;
  int flip;
  // destination list (per member in per-member-list mode, else the same for all)
  float* dets5;
  unsigned long long* keys;
  int* count;
  int cap;
};
struct AppendGK {
  int n, topN, pass, per_member;
  float thresh;
  int blk_start[TG + 1];
  AppendGM m[TG];
};

__device__ __forceinline__ int survivors(const float* probs2, int R, float thresh) {
  int lo = 0, hi = R;  // first r with !(fg > thresh)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (probs2[mid * 2 + 1] > thresh) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void append_dets_kernel(AppendGK g) {
  const int mi = tail_find_member(g.blk_start, blockIdx.x);
  const AppendGM& p = g.m[mi];
  const int lb = blockIdx.x - g.blk_start[mi], lgrid = g.blk_start[mi + 1] - g.blk_start[mi];
  int base, ns;
  if (g.per_member) {
    base = 0;
    ns = survivors(p.probs2, tail_rows(p.counters, g.topN, nullptr), g.thresh);
    if (lb == 0 && threadIdx.x == 0) { p.count[0] = 0; p.count[1] = min(ns, p.cap); }
  } else {
    base = p.count[g.pass & 1];
    for (int v = 0; v < mi; ++v) base += survivors(g.m[v].probs2, tail_rows(g.m[v].counters, g.topN, nullptr), g.thresh);
    ns = survivors(p.probs2, tail_rows(p.counters, g.topN, nullptr), g.thresh);
    if (mi == g.n - 1 && lb == 0 && threadIdx.x == 0) p.count[(g.pass + 1) & 1] = min(base + ns, p.cap);
  }
  for (int r = lb * blockDim.x + threadIdx.x; r < ns; r += lgrid * blockDim.x) {
    const int pos = base + r;
    if (pos >= p.cap) continue;
    const float fg = p.probs2[r * 2 + 1];
    float x1 = p.boxes5[r * 5 + 1], y1 = p.boxes5[r * 5 + 2], x2 = p.boxes5[r * 5 + 3], y2 = p.boxes5[r * 5 + 4];
    if (p.flip) {  // boxes[:, [1,3]] = w - boxes[:, [3,1]]
      const float nx1 = p.im_w - x2, nx2 = p.im_w - x1;
      x1 = nx1;
      x2 = nx2;
    }
    float* d = p.dets5 + (size_t)pos * 5;
    d[0] = x1 / p.im_scale; d[1] = y1 / p.im_scale; d[2] = x2 / p.im_scale; d[3] = y2 / p.im_scale; d[4] = fg;
    p.keys[pos] = ((unsigned long long)__float_as_uint(fg) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)pos);
  }
}

int launch_append_dets_group(const AppendUnit* us, int n, int topN, float thresh, int pass, int per_member,
                             hipStream_t s) {
  if (n < 1 || n > TG) { set_error("append group: 1..16 units"); return -1; }
  AppendGK g;
  g.n = n; g.topN = topN; g.pass = pass; g.per_member = per_member; g.thresh = thresh;
  int blocks = 0;
  for (int q = 0; q <= TG; ++q) g.blk_start[q] = 0x7fffffff;
  for (int m = 0; m < n; ++m) {
    AppendGM& d = g.m[m];
    d.boxes5 = us[m].boxes5; d.probs2 = us[m].probs2; d.counters = us[m].counters;
    d.im_w = us[m].im_w; d.im_scale = us[m].im_scale; d.flip = us[m].flip;
    d.dets5 = us[m].dets5; d.keys = us[m].keys; d.count = us[m].count; d.cap = us[m].cap;
    g.blk_start[m] = blocks;
    blocks += (int)grid_for(us[m].r_max < 1 ? 1 : us[m].r_max);
  }
  g.blk_start[n] = blocks;
  hipLaunchKernelGGL(append_dets_kernel, dim3(blocks), dim3(256), 0, s, g);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
