// Per-image box merging on the device: greedy NMS and the reference's default bbox_vote.
//   nms_kernel / devIoU / host reduce    lib/nms/nms_kernel.cu:24-32,45-89,138-150
//   gpu_nms (sort + order[keep])         lib/nms/gpu_nms.pyx:16-31
//   bbox_vote                            lib/test.py:181-217
// The pairwise IoU bit matrix keeps the reference's 64-column-per-word layout (one u64
// word per lane-row of a wave64); the serial host reduce of the reference becomes an
// on-device wavefront scan that resolves 64 boxes at a time out of registers.
// Compiled with -ffp-contract=off: IoU must round exactly like devIoU / numpy.
#include <algorithm>

#include "shf_internal.h"

namespace shf {

typedef unsigned long long u64;

static inline unsigned grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

__global__ void make_keys_kernel(const float* __restrict__ dets5, int n, u64* __restrict__ keys) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    keys[i] = ((u64)__float_as_uint(dets5[i * 5 + 4]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
}

int launch_make_keys(const float* dets5, int n, u64* keys, hipStream_t s) {
  hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(n)), dim3(256), 0, s, dets5, n, keys);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

__global__ void gather_sorted_kernel(const float* __restrict__ dets5, const u64* __restrict__ keys, int n,
                                     float* __restrict__ sorted5, int* __restrict__ perm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned src = 0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull);
    perm[i] = (int)src;
#pragma unroll
    for (int j = 0; j < 5; ++j) sorted5[i * 5 + j] = dets5[src * 5 + j];
  }
}

int launch_gather_sorted(const float* dets5, const u64* keys, int n, float* sorted5, int* perm, hipStream_t s) {
  hipLaunchKernelGGL(gather_sorted_kernel, dim3(grid_for(n)), dim3(256), 0, s, dets5, keys, n, sorted5, perm);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// devIoU, op for op (nms_kernel.cu:24-32)
__device__ __forceinline__ float dev_iou(const float* a, const float* b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

// one wave64 per (row block, col block >= row block); lane = row box, bit = col box
template <int GE>
__global__ __launch_bounds__(64) void iou_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                      u64* __restrict__ mask) {
  const int nb = (n + 63) >> 6;
  // upper-triangular block enumeration: blockIdx.x -> (rb, cb), cb >= rb
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
  __shared__ float bb[64 * 4];
  const int t = threadIdx.x;
  if (t < col_size) {
    const float* s = boxes + (size_t)(cb * 64 + t) * 5;
    bb[t * 4 + 0] = s[0]; bb[t * 4 + 1] = s[1]; bb[t * 4 + 2] = s[2]; bb[t * 4 + 3] = s[3];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = rb * 64 + t;
    const float* s = boxes + (size_t)cur * 5;
    const float me[4] = {s[0], s[1], s[2], s[3]};
    u64 bits = 0;
    const int start = (rb == cb) ? t + 1 : 0;
    for (int i = start; i < col_size; ++i) {
      const float o = dev_iou(me, bb + i * 4);
      if (GE ? (o >= thr) : (o > thr)) bits |= 1ull << i;
    }
    mask[(size_t)cur * nb + cb] = bits;
  }
}

int launch_iou_mask(const float* sorted5, int n, float thr, int ge_pred, u64* mask, hipStream_t s) {
  const int nb = (n + 63) / 64;
  if (nb == 0) return 0;
  dim3 g(nb, nb);
  if (ge_pred)
    hipLaunchKernelGGL(iou_mask_kernel<1>, g, dim3(64), 0, s, sorted5, n, thr, mask);
  else
    hipLaunchKernelGGL(iou_mask_kernel<0>, g, dim3(64), 0, s, sorted5, n, thr, mask);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Greedy scan, one block of 16 waves.  Boxes are resolved 64 at a time: the 64x64 diagonal tile of the bit matrix sits
// in the registers of wave 0 (lane j = row j) and the within-block dependency chain runs on scalar registers
// (readlane, find-first-set over the not-yet-removed candidates: one turn per KEPT box, not per box); then the rows of
// the block's kept heads are OR-ed into the removed bitmap of all later 64-box words, one word per wave and turn, lane j
// holding row j's word -- fetched for ALL 64 rows BEFORE the chain's result is known, so that the loads fly while wave 0
// runs the chain.  cluster[k] = index (sorted order) of the head that absorbed box k = the first kept head, in
// ascending order, whose row covers k (a ballot per newly removed box).
// (Round 2: one wave did both jobs, 2.4 us per block of 64 boxes = 0.26 ms for the bench image's 6 800 boxes.)
// ---------------------------------------------------------------------------
constexpr int SCAN_MAX_WORDS = 4096;  // 262144 boxes
constexpr int SCAN_WAVES = 16, SCAN_PRE = 4;   // words per wave whose rows are fetched ahead of the chain

__device__ __forceinline__ u64 wave_or_u64(u64 v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    lo |= (unsigned)__shfl_xor((int)lo, o, 64);
    hi |= (unsigned)__shfl_xor((int)hi, o, 64);
  }
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(64 * SCAN_WAVES) void greedy_scan_kernel(const u64* __restrict__ mask, int n,
                                                                      int* __restrict__ cluster, int* __restrict__ heads,
                                                                      int* __restrict__ counters) {
  __shared__ u64 removed[SCAN_MAX_WORDS];
  __shared__ u64 s_kept;
  const int nw = (n + 63) >> 6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = threadIdx.x; w < nw; w += 64 * SCAN_WAVES) removed[w] = 0;
  __syncthreads();
  int nheads = 0;   // (wave 0)
  for (int blk = 0; blk < nw; ++blk) {
    const int base = blk << 6;
    const int cnt = min(n - base, 64);
    // rows of this block at the first SCAN_PRE words this wave owns (w = blk + 1 + wave + 16 q): in flight during the chain
    u64 pre[SCAN_PRE];
#pragma unroll
    for (int q = 0; q < SCAN_PRE; ++q) {
      const int w = blk + 1 + wave + SCAN_WAVES * q;
      pre[q] = (w < nw && lane < cnt) ? mask[(size_t)(base + lane) * nw + w] : 0ull;
    }
    if (wave == 0) {
      const u64 diag = (lane < cnt) ? mask[(size_t)(base + lane) * nw + blk] : 0ull;
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      const u64 r0 = removed[blk];
      // wave-uniform state in scalar registers
      u64 rem = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r0 >> 32)) << 32) |
                (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r0);
      const u64 all = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
      u64 cand = ~rem & all, kept = 0;
      int cl = -1;
      while (cand) {
        const int j = __ffsll((long long)cand) - 1;
        kept |= 1ull << j;
        // readlane returns a signed int: go through unsigned or bit 31 smears over the high word
        const u64 row = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                        (u64)(unsigned)__builtin_amdgcn_readlane((int)dlo, j);
        const u64 nb = row & ~rem;
        if (((nb >> lane) & 1ull) || lane == j) cl = base + j;
        rem |= nb | (1ull << j);
        cand &= ~rem;
      }
      if (cl >= 0) cluster[base + lane] = cl;
      if ((kept >> lane) & 1ull) heads[nheads + __popcll(kept & ((1ull << lane) - 1ull))] = base + lane;
      nheads += __popcll(kept);
      if (lane == 0) s_kept = kept;
    }
    __syncthreads();
    const u64 kept = s_kept;
    const bool mine = (kept >> lane) & 1ull;   // lane j speaks for row j of the block, if it was kept
    auto propagate = [&](int w, u64 val) {
      val = mine ? val : 0ull;
      const u64 orv = wave_or_u64(val);
      const u64 r = removed[w];
      u64 nb = orv & ~r;
      if (nb) {
        if (lane == 0) removed[w] = r | orv;
        while (nb) {   // (wave-uniform)
          const int b = __ffsll((long long)nb) - 1;
          nb &= nb - 1;
          const u64 col = __ballot((val >> b) & 1ull);          // the kept heads of this block that cover box (w, b)
          if (lane == 0) cluster[(w << 6) + b] = base + (__ffsll((long long)col) - 1);
        }
      }
    };
#pragma unroll
    for (int q = 0; q < SCAN_PRE; ++q) {
      const int w = blk + 1 + wave + SCAN_WAVES * q;
      if (w < nw) propagate(w, pre[q]);
    }
    for (int w = blk + 1 + wave + SCAN_WAVES * SCAN_PRE; w < nw; w += SCAN_WAVES)
      propagate(w, (lane < cnt) ? mask[(size_t)(base + lane) * nw + w] : 0ull);
    __syncthreads();
  }
  if (threadIdx.x == 0) counters[0] = nheads;
}

int launch_greedy_scan(const u64* mask, int n, int* cluster, int* heads, int* counters, hipStream_t s) {
  if ((n + 63) / 64 > SCAN_MAX_WORDS) { set_error("merge: too many boxes for the scan bitmap"); return -1; }
  hipLaunchKernelGGL(greedy_scan_kernel, dim3(1), dim3(64 * SCAN_WAVES), 0, s, mask, n, cluster, heads, counters);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// bbox_vote accumulation, one thread per cluster head (test.py:199-214).
// numpy semantics reproduced: fp32 products, row-order (sequential) fp32 sum of the
// weighted boxes (np.sum(axis=0) on an (m,4) view), numpy's pairwise summation for the
// strided 1-D score sum, fp32 divide, result widened to fp64.
// ---------------------------------------------------------------------------
struct MemberIter {
  const u64* row;  // mask row of the head
  const int* cluster;
  const float* dets;
  int head, nw, w;
  u64 cur;
  bool first;
  __device__ void init(const u64* mask, const int* cl, const float* d, int h, int nw_) {
    row = mask + (size_t)h * nw_;
    cluster = cl; dets = d; head = h; nw = nw_;
    w = h >> 6;
    cur = row[w];
    first = true;
  }
  // next member index in ascending order, or -1
  __device__ int next() {
    if (first) { first = false; return head; }
    for (;;) {
      while (cur) {
        const int b = __ffsll((long long)cur) - 1;
        cur &= cur - 1;
        const int k = (w << 6) + b;
        if (cluster[k] == head) return k;
      }
      if (++w >= nw) return -1;
      cur = row[w];
    }
  }
};

// numpy pairwise_sum over the next `n` member scores of the iterator
// (numpy/core/src/umath/loops_utils.h.src, PW_BLOCKSIZE 128)
__device__ float pairwise_scores(MemberIter& it, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res += it.dets[it.next() * 5 + 4];
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = it.dets[it.next() * 5 + 4];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += it.dets[it.next() * 5 + 4];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += it.dets[it.next() * 5 + 4];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const float a = pairwise_scores(it, n2);
  const float b = pairwise_scores(it, n - n2);
  return a + b;
}

// one cluster, one thread: the round-1/2 form, kept for clusters beyond the wave kernel's LDS list
__device__ void vote_one_serial(const float* __restrict__ dets, const u64* __restrict__ mask, const int* __restrict__ cluster,
                                int nw, int h, int t, int nheads, double* __restrict__ rows, int* __restrict__ emit) {
  MemberIter it;
  it.init(mask, cluster, dets, h, nw);
  int m = 0;
  while (it.next() >= 0) ++m;
  double* o = rows + (size_t)t * 5;
  if (m <= 1) {
    // a lone box is dropped unless nothing is left after it (test.py:200-206)
    if (t == nheads - 1) {
      for (int j = 0; j < 5; ++j) o[j] = (double)dets[h * 5 + j];
      emit[t] = 1;
    } else {
      emit[t] = 0;
    }
    return;
  }
  float sx1 = 0.f, sy1 = 0.f, sx2 = 0.f, sy2 = 0.f, mx = 0.f;
  it.init(mask, cluster, dets, h, nw);
  for (int i = 0; i < m; ++i) {
    const float* d = dets + (size_t)it.next() * 5;
    const float sc = d[4];
    const float px1 = d[0] * sc, py1 = d[1] * sc, px2 = d[2] * sc, py2 = d[3] * sc;
    if (i == 0) { sx1 = px1; sy1 = py1; sx2 = px2; sy2 = py2; mx = sc; }
    else { sx1 += px1; sy1 += py1; sx2 += px2; sy2 += py2; mx = fmaxf(mx, sc); }
  }
  it.init(mask, cluster, dets, h, nw);
  const float ssum = pairwise_scores(it, m);
  o[0] = (double)(sx1 / ssum); o[1] = (double)(sy1 / ssum);
  o[2] = (double)(sx2 / ssum); o[3] = (double)(sy2 / ssum);
  o[4] = (double)mx;
  emit[t] = 1;
}

// numpy pairwise_sum over a[0..n) (the same recursion as pairwise_scores, on an array)
__device__ float pairwise_array(const float* a, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_array(a, n2) + pairwise_array(a + n2, n - n2);
}

// One WAVE per cluster head.  The members (the head, then the boxes of its mask row that the scan gave to it, ascending)
// are listed in LDS by all 64 lanes -- a lane filters one 64-box word of the row, a wave prefix sum places its hits --,
// the fp32 products are formed in parallel, and the order-sensitive parts run on the list: lane c < 4 adds coordinate c
// in member order (np.sum(axis=0) of the (m, 4) products), every lane forms numpy's pairwise score sum.  Bit-for-bit
// what the one-thread form computes (round 2: 0.21 ms for the bench image, latency-bound on dependent row loads).
constexpr int VOTE_CAP = 1024;
__global__ __launch_bounds__(64) void vote_accumulate_kernel(const float* __restrict__ dets, const u64* __restrict__ mask,
                                                             const int* __restrict__ cluster, int n,
                                                             const int* __restrict__ heads,
                                                             const int* __restrict__ counters, double* __restrict__ rows,
                                                             int* __restrict__ emit) {
  __shared__ int idx[VOTE_CAP];
  __shared__ float prod[VOTE_CAP * 4];
  __shared__ float scs[VOTE_CAP];
  const int nheads = counters[0];
  const int nw = (n + 63) >> 6;
  const int lane = threadIdx.x;
  for (int t = blockIdx.x; t < nheads; t += gridDim.x) {
    const int h = heads[t];
    const u64* row = mask + (size_t)h * nw;
    __syncthreads();   // (the previous cluster's list is dead)
    if (lane == 0) idx[0] = h;
    int m = 1;
    for (int wb = h >> 6; wb < nw; wb += 64) {
      const int w = wb + lane;
      u64 cur = w < nw ? row[w] : 0ull, fm = 0;
      while (cur) {
        const int b = __ffsll((long long)cur) - 1;
        cur &= cur - 1;
        if (cluster[(w << 6) + b] == h) fm |= 1ull << b;
      }
      const int cntl = __popcll(fm);
      int incl = cntl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      int pos = m + incl - cntl;
      while (fm) {
        const int b = __ffsll((long long)fm) - 1;
        fm &= fm - 1;
        if (pos < VOTE_CAP) idx[pos] = (w << 6) + b;
        ++pos;
      }
      m += __shfl(incl, 63, 64);
    }
    if (m > VOTE_CAP) {   // (wave-uniform) a cluster beyond the list: the serial form
      if (lane == 0) vote_one_serial(dets, mask, cluster, nw, h, t, nheads, rows, emit);
      continue;
    }
    double* o = rows + (size_t)t * 5;
    if (m <= 1) {
      // a lone box is dropped unless nothing is left after it (test.py:200-206)
      if (t == nheads - 1) {
        if (lane < 5) o[lane] = (double)dets[h * 5 + lane];
        if (lane == 0) emit[t] = 1;
      } else if (lane == 0) {
        emit[t] = 0;
      }
      continue;
    }
    __syncthreads();
    float mx = 0.f;
    for (int i = lane; i < m; i += 64) {
      const float* d = dets + (size_t)idx[i] * 5;
      const float sc = d[4];
      prod[i * 4 + 0] = d[0] * sc; prod[i * 4 + 1] = d[1] * sc; prod[i * 4 + 2] = d[2] * sc; prod[i * 4 + 3] = d[3] * sc;
      scs[i] = sc;
      mx = fmaxf(mx, sc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    __syncthreads();
    const float ssum = pairwise_array(scs, m);
    if (lane < 4) {
      float sacc = prod[lane];
      for (int i = 1; i < m; ++i) sacc += prod[i * 4 + lane];
      o[lane] = (double)(sacc / ssum);
    }
    if (lane == 0) {
      o[4] = (double)mx;
      emit[t] = 1;
    }
  }
}

// order-preserving compaction of the emitted rows (single block)
__global__ __launch_bounds__(1024) void compact_rows_kernel(const double* __restrict__ rows,
                                                            const int* __restrict__ emit,
                                                            const int* __restrict__ counters,
                                                            double* __restrict__ out, int* __restrict__ n_out) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int nheads = counters[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < nheads; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    const int e = (t < nheads) ? emit[t] : 0;
    const u64 bal = __ballot(e != 0);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (e) {
      double* o = out + (size_t)(before + within) * 5;
      const double* r = rows + (size_t)t * 5;
      for (int j = 0; j < 5; ++j) o[j] = r[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wsum[w];
      carry += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = carry;
}

int launch_vote_accumulate(const float* sorted5, const u64* mask, const int* cluster, int n, const int* heads,
                           const int* counters, double* out5, int* n_out, hipStream_t s) {
  // rows/emit scratch live behind the compacted output: out5 has room for 2n rows (see net_internal.h: MergeCtx)
  double* rows = out5 + (size_t)n * 5;
  int* emit = (int*)(rows + (size_t)n * 5);
  // (one wave per cluster head, grid-stride: the head count is only known on the device)
  hipLaunchKernelGGL(vote_accumulate_kernel, dim3((unsigned)std::min<long long>(std::max(n, 1), 4096)), dim3(64), 0, s, sorted5, mask,
                     cluster, n, heads, counters, rows, emit);
  hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, s, rows, emit, counters, out5, n_out);
  SHF_HIP_OK(hipGetLastError());
  return 0;
}

}  // namespace shf
