// Net runtime + C ABI (include/shf_hip.h).
//
// A static-graph executor for the detector's TEST-phase prototxt: it replaces
// caffe::Net (caffe/src/caffe/net.cpp:28-257 Init, :421-513 AppendParam sharing,
// :516-532 ForwardFromTo, :733-768 CopyTrainedLayersFrom), Blob/SyncedMemory
// (blob.cpp:23-51, syncedmem.cpp:39-91) and the in-graph Python ProposalLayer
// trampoline (include/caffe/layers/python_layer.hpp:14-51) for the layer types that
// graph instantiates.  Differences by design (MI355X-first):
//   * activations stay NHWC on the device; only Blob.data read-back transposes;
//   * conv + bias + in-place ReLU are one kernel; channel concat is zero-copy
//     (producers write channel slices of the concat buffer);
//   * the 1x1 cls/reg convs, concats, softmax, reshape and the proposal layer are
//     one fused device-side tail (no D2H, no Python re-entry);
//   * buffers are grow-only and shape changes re-plan nothing but pointers/sizes.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>

#include "../../include/shf_hip.h"
#include "proto_text.h"
#include "shf_internal.h"

namespace shf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

#define HIP_THROW(expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define CHECK_RC(expr)                                     \
  do {                                                     \
    if ((expr) != 0) throw std::runtime_error(g_err);      \
  } while (0)
// ... naming the layer (a launcher's message speaks of shapes and kernels, not of the graph)
#define CHECK_RC_LAYER(expr, lname)                                                           \
  do {                                                                                        \
    if ((expr) != 0) throw std::runtime_error("layer '" + std::string(lname) + "': " + g_err); \
  } while (0)

// hipMemset on the null stream may return before the fill has run, and the null stream is not ordered with the runtime's
// non-blocking streams: a fill that must be in place before the first kernel touches the buffer is waited for here.  (A
// lane cloned inside FusedDetector.submit launches its first convolutions microseconds after its slots are cleared: a
// late fill zeroed slots the first kernels had already published into -- the first image of a process then ran a unit with
// e = 0 and differed from the same image processed later in the last bit.)
static void fill_now(void* p, int byte, size_t n) {
  HIP_THROW(hipMemsetAsync(p, byte, n, nullptr));
  HIP_THROW(hipStreamSynchronize(nullptr));
}

// (re)allocations made by the grow-only buffers of this process: a new level shape re-plans sizes and pointers and only
// allocates where a buffer has to grow (shf_alloc_counts; bench.py's mixed-shape leg reports them after its first pass)
static std::atomic<long long> g_dev_allocs{0}, g_host_allocs{0};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    ++g_dev_allocs;
    if (p) HIP_THROW(hipFree(p));
    p = nullptr;
    size_t want = bytes + bytes / 8;  // grow-only with slack (Blob::Reshape never shrinks, blob.cpp:46-50)
    want = (want + 255) & ~(size_t)255;
    HIP_THROW(hipMalloc(&p, want));
    cap = want;
    // SHF_POISON_ALLOC=<byte> (tests): fresh device buffers start filled with that byte (0xff: NaNs) instead of whatever the
    // allocator hands out -- a kernel whose result depends on memory it never wrote shows up as a changed detection
    static const char* poison = getenv("SHF_POISON_ALLOC");
    if (poison) fill_now(p, (int)strtol(poison, nullptr, 0), want);
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

struct HostBuf {
  float* p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    float* np = nullptr;
    ++g_host_allocs;
    size_t want = std::max<size_t>(bytes + bytes / 8, 64);
    HIP_THROW(hipHostMalloc((void**)&np, want, hipHostMallocDefault));
    if (p) {
      memcpy(np, p, cap);
      (void)hipHostFree(p);
    }
    memset((char*)np + cap, 0, want - cap);
    p = np;
    cap = want;
  }
  ~HostBuf() {
    if (p) (void)hipHostFree(p);
  }
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
};

struct ParamBlob {
  std::vector<int> shape;
  std::vector<float> host;
  DevBuf raw, packed, packed16, packed16h, packed16r, first_t, first_frag;   // (packed16r: the fused first pair's rotated-row pack)
  DevBuf packed16b, packed16hb, packed16rb, first_frag_b;   // the same three packs for conv mode "bf16" (built on first use of the mode)
  bool bf_stale = true;                         // ... and whether they hold the current weights
  float wscale_inv = 1.f;  // packed16h: the power of two its weights were scaled by, inverted
  bool dirty = true;
  bool split_stale = false;  // committed while the net was in fp32 mode: packed16 / packed16h hold OLDER weights
  size_t count() const {
    size_t c = 1;
    for (int d : shape) c *= (size_t)d;
    return c;
  }
};

enum BlobKind { BK_INPUT_NCHW, BK_NHWC, BK_FLAT, BK_NCHW_MAT, BK_FUSED };

struct Blob {
  std::string name;
  std::vector<int> shape;  // logical Caffe shape
  BlobKind kind = BK_NHWC;
  int owner = -1;  // blob owning the device buffer (channel-concat views)
  int coff = 0;
  DevBuf dev, stage;
  HostBuf host;
  bool host_newer = false, dev_newer = false;
  const float* ext_dev = nullptr;  // externally bound device input (fused path)
  bool split_fused = false;        // fused split-fp16 path: stored pre-split ([chunk][hi|lo] fp16), see ConvArgs::in_split
  size_t count() const {
    size_t c = 1;
    for (int d : shape) c *= (size_t)d;
    return c;
  }
};

enum OpType { OP_SKIP, OP_CONV, OP_POOL, OP_DECONV, OP_TAIL };

struct Layer {
  std::string name, type;
  const PMsg* msg = nullptr;
  std::vector<int> bottoms, tops;
  std::vector<std::shared_ptr<ParamBlob>> params;
  OpType op = OP_SKIP;
  // conv / deconv / pool hyper-parameters
  int k = 1, pad = 0, stride = 1, dil = 1, group = 1, nout = 0, relu = 0, bias_term = 1;
  int kclass = 0;
  int fuse_pool = -1;      // conv: index of the 2x2/2 MAX pool folded into its epilogue (fused path only)
  bool pool_only = false;  // conv: its un-pooled top has no other reader
  int fused_into = -1;     // pool: index of the conv that produces it in the fused path
  int first_src = -1;      // conv: index of the first-layer conv computed inside this conv's halo staging (f16x3)
  int first_dst = -1;      // first-layer conv: index of the conv that absorbs it
  // the three shared-weight dilated heads (prototxt :480-552): on the dilation-1 layer, the indices of its dilation-2 / -4
  // siblings (same bottom, same parameter blobs); on those, the index of the dilation-1 layer.  One launch covers the three
  // when the shapes and the mode allow (conv_f16x3_group_is_heads3).
  int heads3_d2 = -1, heads3_d4 = -1, heads3_lead = -1;
};

// the first 8 classes are the instantiations of conv_mfma_f32_kernel, named like rocprofv3 prints them
enum ProfClass { PC_CONV_MFMA, PC_CONV_MFMA_1 = 1, PC_CONV_MFMA_7 = 7, PC_CONV_F16X3_128, PC_CONV_F16X3_W4, PC_CONV_F16X3_W4_SPLIT, PC_CONV_F16X3_W4_MT2, PC_CONV_F16X3_W4_SPLIT_MT2, PC_CONV_F16X3_64, PC_CONV_F16X3_64_FUSE1, PC_CONV_F16X3_64_D2, PC_CONV_F16X3_64_D4, PC_CONV_F16X3_128_K1, PC_CONV_F16X3_64_K1, PC_CONV_F16X3_PC, PC_CONV_F16X3_W4D_0, PC_CONV_F16X3_W4D_7 = PC_CONV_F16X3_W4D_0 + 7, PC_CONV_F16X3_PCP, PC_CONV_F16X3_K1G, PC_CONV_F16X3_W4D_D2, PC_CONV_F16X3_W4D_D4, PC_CONV_F16X3_H3, PC_CONV_FIRST, PC_CONV_DIRECT, PC_POOL, PC_DECONV, PC_TAIL, PC_MERGE, PC_LAYOUT, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"conv_mfma_f32_kernel<3, 1, 128, 8, 16>", "conv_mfma_f32_kernel<3, 2, 128, 8, 16>",
                                           "conv_mfma_f32_kernel<3, 4, 128, 8, 16>", "conv_mfma_f32_kernel<3, 1, 64, 16, 16>",
                                           "conv_mfma_f32_kernel<3, 2, 64, 16, 16>", "conv_mfma_f32_kernel<3, 4, 64, 16, 16>",
                                           "conv_mfma_f32_kernel<1, 0, 128, 8, 16>", "conv_mfma_f32_kernel<1, 0, 64, 16, 16>",
                                           "conv_mfma_f16x3_kernel<128, false, 1, 3, 3, false>", "(retired: single-tile 4-wave kernel)",
                                           "(retired)", "(retired)", "(retired)",
                                           "conv_mfma_f16x3_kernel<64, false, 1, 3, 3, false>",
                                           "conv_mfma_f16x3_kernel<64, true, 1, 3, 3, false>", "conv_mfma_f16x3_kernel<64, false, 2, 3, 3, false>",
                                           "conv_mfma_f16x3_kernel<64, false, 4, 3, 3, false>", "conv_mfma_f16x3_kernel<128, false, 1, 1, 3, false>",
                                           "conv_mfma_f16x3_kernel<64, false, 1, 1, 3, false>", "conv_mfma_f16x3_pc_kernel<3, false, false>",
                                           // dual-tile family <IN_SPLIT, rows / 4, tiles per block, products, bf16, dilation>: index = in_split * 4 + (rows == 8) * 2 + (tiles == 1)
                                           "conv_mfma_f16x3_w4d_kernel<false, 4, 2, 3, false, 1>", "conv_mfma_f16x3_w4d_kernel<false, 4, 1, 3, false, 1>",
                                           "conv_mfma_f16x3_w4d_kernel<false, 2, 2, 3, false, 1>", "conv_mfma_f16x3_w4d_kernel<false, 2, 1, 3, false, 1>",
                                           "conv_mfma_f16x3_w4d_kernel<true, 4, 2, 3, false, 1>", "conv_mfma_f16x3_w4d_kernel<true, 4, 1, 3, false, 1>",
                                           "conv_mfma_f16x3_w4d_kernel<true, 2, 2, 3, false, 1>", "conv_mfma_f16x3_w4d_kernel<true, 2, 1, 3, false, 1>",
                                           // kernels of their own (names: the headline mode's instantiation -- split-format input, three
                                           // products; the reduced modes run the same templates with other NP / BF arguments): the persistent
                                           // first pair, the 1x1 GEMM, the family's dilated forms, the three heads in one launch
                                           "conv_mfma_f16x3_pc_kernel<3, false, true>", "conv_mfma_f16x3_k1_kernel<true, 3>",
                                           "conv_mfma_f16x3_w4d_kernel<true, 4, 1, 3, false, 2>", "conv_mfma_f16x3_w4d_kernel<true, 4, 1, 3, false, 4>",
                                           "conv_mfma_f16x3_heads3_kernel<true, 3>",
                                           "conv_first_kernel", "conv_direct_kernel", "maxpool_kernel",
                                           "deconv_depthwise", "detect_tail", "box_merge", "layout"};

struct Prof {
  bool on = false;
  int only = -1;   // >= 0: only launches of this class are bracketed (shf_prof_only)
  bool wants(int cls) const { return on && (only < 0 || only == cls); }
  struct Rec { int cls; hipEvent_t a, b; double flops, bytes; };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;
  int64_t launches[PC_COUNT] = {0};
  double ms[PC_COUNT] = {0}, flops[PC_COUNT] = {0}, bytes[PC_COUNT] = {0};
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e;
    HIP_THROW(hipEventCreate(&e));
    return e;
  }
  void drain() {
    for (auto& r : pending) {
      HIP_THROW(hipEventSynchronize(r.b));
      float t = 0;
      HIP_THROW(hipEventElapsedTime(&t, r.a, r.b));
      launches[r.cls]++;
      ms[r.cls] += t;
      flops[r.cls] += r.flops;
      bytes[r.cls] += r.bytes;
      pool.push_back(r.a);
      pool.push_back(r.b);
    }
    pending.clear();
  }
  ~Prof() {
    for (auto& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
  }
};

// which split-fp16 kernel launch_conv_f16x3_group picks for these arguments -- decided by the launcher's OWN predicates on the
// actual arguments, so that the 8-wave fallbacks (unaligned views, Cout % 256, bf16 1x1s ...) are not booked under the name of
// the kernel the knobs would normally select
static int f16x3_prof_class(const ConvArgs& a, int nout, const ConvArgs* group = nullptr, int n = 1) {
  const ConvArgs* as = group ? group : &a;
  if (a.img) {
    if (!(conv_f16x3_uses_pc() && a.in.C == 64 && nout == 64)) return PC_CONV_F16X3_64_FUSE1;
    return conv_f16x3_pc_persistent() ? PC_CONV_F16X3_PCP : PC_CONV_F16X3_PC;
  }
  if (a.k == 1) return conv_f16x3_group_is_k1_gemm(as, n) ? PC_CONV_F16X3_K1G : (nout % 128 ? PC_CONV_F16X3_64_K1 : PC_CONV_F16X3_128_K1);
  if (a.dil == 2) return conv_f16x3_group_is_dilated_w4(as, n) ? PC_CONV_F16X3_W4D_D2 : PC_CONV_F16X3_64_D2;
  if (a.dil == 4) return conv_f16x3_group_is_dilated_w4(as, n) ? PC_CONV_F16X3_W4D_D4 : PC_CONV_F16X3_64_D4;
  if (nout % 128) return PC_CONV_F16X3_64;
  return PC_CONV_F16X3_128;   // (the dual-tile family reports through SubProf, one record per kernel of the layer)
}

static int conv_prof_class(int k, int dil, int nout) {
  const int bn64 = (nout % 128 == 0) ? 0 : 1;
  if (k == 1) return 6 + bn64;
  const int d = dil == 1 ? 0 : dil == 2 ? 1 : 2;
  return bn64 * 3 + d;
}

struct ProfScope {
  Prof& p;
  hipStream_t s;
  Prof::Rec r;
  bool on;
  ProfScope(Prof& p_, hipStream_t s_, int cls, double flops, double bytes) : p(p_), s(s_), on(p_.wants(cls)) {
    if (!on) return;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    r.a = p.get(); r.b = p.get();
    HIP_THROW(hipEventRecord(r.a, s));
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    p.pending.push_back(r);
  }
};

// dual-tile conv family: one profiler record per kernel of a (possibly two-launch) layer, through ConvArgs::sub_hook
struct SubProf {
  Prof* p;
  hipStream_t s;
  double flops, bytes;
  Prof::Rec r;
  static void hook(void* ctx, int after, int variant, double share) {
    SubProf* sp = (SubProf*)ctx;
    if (!sp->p->wants(PC_CONV_F16X3_W4D_0 + variant)) return;
    if (!after) {
      sp->r.cls = PC_CONV_F16X3_W4D_0 + variant;
      sp->r.flops = sp->flops * share;
      sp->r.bytes = sp->bytes * share;
      sp->r.a = sp->p->get();
      sp->r.b = sp->p->get();
      (void)hipEventRecord(sp->r.a, sp->s);
    } else {
      (void)hipEventRecord(sp->r.b, sp->s);
      sp->p->pending.push_back(sp->r);
    }
  }
};

// ---------------------------------------------------------------------------
// box merging context (also used stand-alone by shf_nms / shf_bbox_vote)
// ---------------------------------------------------------------------------
struct MergeCtx {
  DevBuf dets, keys, sorted, perm, mask, cluster, heads, counters, out;
  std::vector<double> hout;
  std::vector<int> hidx;

  // dets_dev: (n,5) fp32 on the device.  method 0 = vote (>=), 1 = nms (>).
  // vote: rows -> out5 (cap rows) ; nms: kept ORIGINAL indices -> keep
  int run(const float* dets_dev, int n, int method, float thr, double* out5, int cap, int* n_out, int32_t* keep,
          hipStream_t s) {
    *n_out = 0;
    if (n <= 0) return 0;
    size_t npad = 1;
    while (npad < (size_t)n) npad <<= 1;
    const size_t nw = ((size_t)n + 63) / 64;
    keys.ensure(npad * 8);
    sorted.ensure((size_t)n * 5 * 4);
    perm.ensure((size_t)n * 4);
    mask.ensure((size_t)n * nw * 8);
    cluster.ensure((size_t)n * 4);
    heads.ensure((size_t)n * 4);
    counters.ensure(64);
    out.ensure((size_t)n * 5 * 8 * 2 + (size_t)n * 4 + 64);
    int* cnt = (int*)counters.p;
    HIP_THROW(hipMemsetAsync(cnt, 0, 64, s));
    HIP_THROW(hipMemcpyAsync(cnt + 3, &n, sizeof(int), hipMemcpyHostToDevice, s));
    CHECK_RC(launch_make_keys(dets_dev, n, (unsigned long long*)keys.p, s));
    CHECK_RC(launch_sort_desc_u64((unsigned long long*)keys.p, cnt + 3, (size_t)n, s));
    CHECK_RC(launch_gather_sorted(dets_dev, (unsigned long long*)keys.p, n, (float*)sorted.p, (int*)perm.p, s));
    CHECK_RC(launch_iou_mask((float*)sorted.p, n, thr, method == 0 ? 1 : 0, (unsigned long long*)mask.p, s));
    CHECK_RC(launch_greedy_scan((unsigned long long*)mask.p, n, (int*)cluster.p, (int*)heads.p, cnt, s));
    if (method == 0) {
      CHECK_RC(launch_vote_accumulate((float*)sorted.p, (unsigned long long*)mask.p, (int*)cluster.p, n,
                                      (int*)heads.p, cnt, (double*)out.p, cnt + 1, s));
      int h[2];
      HIP_THROW(hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, s));
      HIP_THROW(hipStreamSynchronize(s));
      const int m = h[1];
      *n_out = m;
      const int w = std::min(m, cap);
      if (w > 0) HIP_THROW(hipMemcpy(out5, out.p, (size_t)w * 5 * 8, hipMemcpyDeviceToHost));
    } else {
      int nh = 0;
      HIP_THROW(hipMemcpyAsync(&nh, cnt, 4, hipMemcpyDeviceToHost, s));
      HIP_THROW(hipStreamSynchronize(s));
      hidx.resize((size_t)n * 2);
      HIP_THROW(hipMemcpy(hidx.data(), heads.p, (size_t)nh * 4, hipMemcpyDeviceToHost));
      HIP_THROW(hipMemcpy(hidx.data() + n, perm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
      *n_out = nh;
      if (keep)
        for (int i = 0; i < nh; ++i) keep[i] = hidx[n + hidx[i]];
      if (out5) {
        std::vector<float> hs((size_t)n * 5);
        HIP_THROW(hipMemcpy(hs.data(), sorted.p, (size_t)n * 5 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < std::min(nh, cap); ++i)
          for (int j = 0; j < 5; ++j) out5[i * 5 + j] = (double)hs[(size_t)hidx[i] * 5 + j];
      }
    }
    return 0;
  }
};

// generate_anchors.py:11-86 in double precision
static void gen_anchors(int base_size, const std::vector<double>& ratios, const std::vector<double>& scales,
                        const std::vector<double>& shifts, const std::vector<double>& strides,
                        std::vector<double>& out) {
  out.clear();
  const double bw = base_size, bh = base_size;  // base anchor (0,0,base-1,base-1)
  const double bxc = 0 + 0.5 * (bw - 1), byc = 0 + 0.5 * (bh - 1);
  const double size = bw * bh;
  for (double r : ratios) {
    const double ws = std::nearbyint(std::sqrt(size / r));
    const double hs = std::nearbyint(ws * r);
    // ratio anchor
    const double rx1 = bxc - 0.5 * (ws - 1), ry1 = byc - 0.5 * (hs - 1);
    const double rx2 = bxc + 0.5 * (ws - 1), ry2 = byc + 0.5 * (hs - 1);
    const double w = rx2 - rx1 + 1, h = ry2 - ry1 + 1;
    const double xc = rx1 + 0.5 * (w - 1), yc = ry1 + 0.5 * (h - 1);
    const size_t ns = std::min(scales.size(), strides.size());  // zip(scales, strides)
    for (size_t j = 0; j < ns; ++j) {
      const double sw = w * scales[j], sh = h * scales[j];
      const double a[4] = {xc - 0.5 * (sw - 1), yc - 0.5 * (sh - 1), xc + 0.5 * (sw - 1), yc + 0.5 * (sh - 1)};
      for (double sy : shifts)
        for (double sx : shifts) {
          out.push_back(a[0] + sx * strides[j]);
          out.push_back(a[1] + sy * strides[j]);
          out.push_back(a[2] + sx * strides[j]);
          out.push_back(a[3] + sy * strides[j]);
        }
    }
  }
}

// "{'feat_stride': [8,8,8],'scales': [1,2,4], 'ratios':[1,]}" -> key -> numbers
static std::map<std::string, std::vector<double>> parse_param_str(const std::string& s) {
  std::map<std::string, std::vector<double>> out;
  size_t i = 0;
  while (i < s.size()) {
    const size_t q = s.find_first_of("'\"", i);
    if (q == std::string::npos) break;
    const size_t q2 = s.find(s[q], q + 1);
    if (q2 == std::string::npos) break;
    const std::string key = s.substr(q + 1, q2 - q - 1);
    size_t c = s.find(':', q2);
    if (c == std::string::npos) break;
    ++c;
    while (c < s.size() && isspace((unsigned char)s[c])) ++c;
    std::vector<double> vals;
    size_t end = c;
    if (c < s.size() && (s[c] == '[' || s[c] == '(')) {
      end = s.find_first_of("])", c);
      if (end == std::string::npos) end = s.size();
      std::string body = s.substr(c + 1, end - c - 1);
      for (auto& ch : body)
        if (ch == ',') ch = ' ';
      std::stringstream ss(body);
      std::string tok;
      while (ss >> tok) {
        if (tok == "True" || tok == "true") vals.push_back(1);
        else if (tok == "False" || tok == "false") vals.push_back(0);
        else vals.push_back(std::strtod(tok.c_str(), nullptr));
      }
      ++end;
    } else {
      end = s.find_first_of(",}", c);
      if (end == std::string::npos) end = s.size();
      std::string tok = s.substr(c, end - c);
      while (!tok.empty() && isspace((unsigned char)tok.back())) tok.pop_back();
      if (tok == "True" || tok == "true") vals.push_back(1);
      else if (tok == "False" || tok == "false") vals.push_back(0);
      else vals.push_back(std::strtod(tok.c_str(), nullptr));
    }
    out[key] = vals;
    i = end;
  }
  return out;
}

static int conv_out(int n, int k, int pad, int stride, int dil) {
  const int kext = dil * (k - 1) + 1;
  return (n + 2 * pad - kext) / stride + 1;
}

}  // namespace shf

namespace shf {
int calib_matrix_pipe(int bf16, int zero_eighths, int constant, int iters, int reps, double* tflops);   // calib.hip
}
using namespace shf;

// configuration shared by a net and every lane cloned from it: the arithmetic mode and what the reference's Python
// layer reads from the global cfg at every forward (lib/layers/proposal_layer.py:88-92)
struct NetShared {
  int conv_mode = 0;  // 0: exact fp32 MFMA everywhere; 1: split-fp16, 3 products (fp32-class accuracy); 2 / 3: the
                      // reduced ladder -- 2 products (activations rounded to fp16) / 1 product (plain fp16 operands)
  std::map<std::string, int> layer_products;  // per-layer override of the number of fp16 products (shf_net_set_layer_products)
  int pre_nms_topN = 10000;
  float score_thresh = 0.002f, min_size = 0.f;
  bool weights_exceed_f16 = false;  // some conv weight is outside the fp16 range: split-fp16 mode refuses to run
  long long range_fallbacks = 0;    // forwards re-run on the exact fp32 kernels after a split-fp16 range overflow
  // image pipeline (shf_net_set_pipeline): ONE in-order stream carries the convolutions + logits kernels of every
  // image; each head's own (high-priority) stream carries the rest of its image's tails, appends and the merge
  hipStream_t conv_stream = nullptr;
  ~NetShared() {
    if (conv_stream) {
      (void)hipStreamSynchronize(conv_stream);
      (void)hipStreamDestroy(conv_stream);
    }
  }
};

struct shf_net {
  std::shared_ptr<NetShared> sh;
  int& conv_mode;
  int& pre_nms_topN;
  float& score_thresh;
  float& min_size;
  explicit shf_net(std::shared_ptr<NetShared> shared = nullptr)
      : sh(shared ? shared : std::make_shared<NetShared>()), conv_mode(sh->conv_mode), pre_nms_topN(sh->pre_nms_topN),
        score_thresh(sh->score_thresh), min_size(sh->min_size) {}
  std::shared_ptr<PMsg> root;
  std::deque<Blob> blobs;
  std::vector<Layer> layers;
  std::map<std::string, int> blob_index;
  std::vector<int> inputs, outputs;
  std::map<std::string, std::shared_ptr<ParamBlob>> shared_params;
  hipStream_t stream = nullptr;
  Prof prof;
  int phase = 1;
  // tail
  int tail_layer = -1;
  std::vector<int> tail_cls_layers, tail_box_layers;  // per head (or single)
  std::vector<int> tail_feat_blobs;
  int tail_A = 0, tail_heads = 0, tail_Cf = 0;
  int tail_cls_blob = -1, tail_box_blob = -1, im_info_blob = -1, boxes_blob = -1, prob_blob = -1, data_blob = -1;
  std::vector<double> anchors;
  std::vector<int> sub_stride;
  int feat_stride = 8;
  DevBuf tail_W, tail_b;
  bool tail_w_dirty = true;
  std::string proto_text;
  shf_net* clone_src = nullptr;     // lanes share the parameter tensors of the net they were cloned from
  std::shared_ptr<int> wgen = std::make_shared<int>(0);  // bumped by every param commit
  int tail_gen = -1;
  hipEvent_t ev_mark = nullptr;
  hipEvent_t ev_logits = nullptr;  // recorded by every fused tail pass right after its logits kernel
  hipEvent_t logits_done = nullptr;  // (not owned) what a later pass over this member waits for: its own ev_logits, or --
                                     // after a pipelined grouped pass -- the head's ev_convs (ONE record for the group:
                                     // ten event records in a row were ~90 us of idle conv stream per image)
  hipEvent_t ev_convs = nullptr;   // group pass: recorded on the head's stream after the last layer before the tails
  shf_net* pred = nullptr;         // shf_net_set_predecessor: the head lane whose image precedes this one's
  bool pipelined = false;   // shf_net_set_pipeline: convolutions go to sh->conv_stream, the rest stays on `stream`
  hipStream_t cstream() { return pipelined && sh->conv_stream ? sh->conv_stream : stream; }
  int* flag_ptr = nullptr;  // the flag this net's kernels raise: its own, or the head's during a grouped pass
  // activation-exponent slots (conv_common.h): one u32 per blob of THIS lane = bit pattern of max |value| of the unit
  // it currently holds; zeroed at the start of every forward / unit, raised by the producers' epilogues, read by the
  // single-accumulator split-fp16 kernels.  Concat members share their owner's slot.
  DevBuf amax_slots;
  unsigned* amax_slot(int bi) {
    if (!amax_slots.p || conv_mode < 1 || conv_mode == 4) return nullptr;
    const int o = blobs[bi].owner >= 0 ? blobs[bi].owner : bi;
    return (unsigned*)amax_slots.p + o;
  }
  void reset_amax(hipStream_t st) {
    if (conv_mode >= 1 && conv_mode != 4 && amax_slots.p) HIP_THROW(hipMemsetAsync(amax_slots.p, 0, blobs.size() * 4, st));
  }
  DevBuf range_flag;  // device int: raised by a split-fp16 conv epilogue that produced |x| > 65504 (fp16 hi overflows)
  TailWork tw;
  DevBuf tw_logits, tw_rec, tw_keys, tw_counters;
  bool materialize_tail = true;
  std::vector<int> last_data_shape;
  // fused per-image path
  DevBuf img_dets, img_keys, img_count;
  int img_cap = 0, img_units = 0;
  int img_pass = 0;  // append passes since detect_begin: img_count[img_pass & 1] is the current list length
  MergeCtx merge;
  float cur_im_info[3] = {0, 0, 1};
  bool use_blob_im_info = true;

  ~shf_net() {
    if (ev_logits) (void)hipEventDestroy(ev_logits);
    if (ev_convs) (void)hipEventDestroy(ev_convs);
    if (ev_mark) (void)hipEventDestroy(ev_mark);
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
  }

  int add_blob(const std::string& name) {
    auto it = blob_index.find(name);
    if (it != blob_index.end()) return it->second;
    Blob b;
    b.name = name;
    blobs.emplace_back();
    blobs.back().name = name;
    blob_index[name] = (int)blobs.size() - 1;
    return (int)blobs.size() - 1;
  }

  View view_of(int bi) {
    Blob& b = blobs[bi];
    const int o = b.owner >= 0 ? b.owner : bi;
    Blob& ob = blobs[o];
    View v;
    v.p = (float*)ob.dev.p;
    v.B = b.shape[0]; v.C = b.shape[1]; v.H = b.shape[2]; v.W = b.shape[3];
    v.cstride = ob.shape[1];
    v.coff = b.coff;
    return v;
  }

  void build(const std::string& text, const char* caffemodel);
  void infer_shapes();
  void alloc_buffers();
  void ensure_tail_workspace(size_t total_anchors);
  void commit_params(int li);
  void build_tail_weights();
  TailArgs tail_args(float im_h, float im_w, float im_scale, bool fused_path);
  float* probs_out() { return prob_blob >= 0 ? (float*)blobs[prob_blob].dev.p : (float*)tw_rec.p; }
  void forward_ops(bool fused_path, float im_h, float im_w, float im_scale, hipStream_t s_override = nullptr,
                   Prof* prof_override = nullptr, int only_layer = -1, ConvArgs* collect = nullptr);
  void prepare_unit(const float* data, int data_on_device, int H, int W, hipStream_t st);
  void ensure_img_cap(int units_after);
  void forward();
  float* host_data(int bi);
  void load_caffemodel(const std::string& path);
};

static int geti(const PMsg* m, const char* n, int d) { return m ? (int)m->num(n, d) : d; }

void shf_net::build(const std::string& text, const char* caffemodel) {
  proto_text = text;
  if (!clone_src && getenv("SHF_CONV_MODE")) conv_mode = atoi(getenv("SHF_CONV_MODE"));
  range_flag.ensure(64);
  fill_now(range_flag.p, 0, 64);
  TextParser tp(proto_text);
  root = tp.parse();
  HIP_THROW(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  CHECK_RC(conv_init_attributes());
  CHECK_RC(conv_f16x3_init_attributes());

  // ---- inputs: legacy `input:` + input_shape / input_dim (upgrade_proto.cpp:966-1000)
  auto in_names = root->all("input");
  auto in_shapes = root->all("input_shape");
  auto in_dims = root->all("input_dim");
  for (size_t i = 0; i < in_names.size(); ++i) {
    const int bi = add_blob(in_names[i]->scalar);
    inputs.push_back(bi);
    std::vector<int> shp;
    if (i < in_shapes.size() && in_shapes[i]->msg)
      for (auto d : in_shapes[i]->msg->all("dim")) shp.push_back(atoi(d->scalar.c_str()));
    else
      for (size_t j = 4 * i; j < 4 * i + 4 && j < in_dims.size(); ++j) shp.push_back(atoi(in_dims[j]->scalar.c_str()));
    if (shp.empty()) shp = {1};
    blobs[bi].shape = shp;
  }
  for (auto lf : root->all("layer")) {
    const PMsg* lm = lf->msg.get();
    if (!lm) continue;
    Layer L;
    L.msg = lm;
    L.name = lm->str("name");
    L.type = lm->str("type");
    if (L.type == "Input") {
      auto tops = lm->all("top");
      const PMsg* ip = lm->sub("input_param");
      auto shapes = ip ? ip->all("shape") : std::vector<const PField*>();
      for (size_t i = 0; i < tops.size(); ++i) {
        const int bi = add_blob(tops[i]->scalar);
        inputs.push_back(bi);
        std::vector<int> shp;
        if (i < shapes.size() && shapes[i]->msg)
          for (auto d : shapes[i]->msg->all("dim")) shp.push_back(atoi(d->scalar.c_str()));
        if (shp.empty()) shp = {1};
        blobs[bi].shape = shp;
        L.tops.push_back(bi);
      }
      layers.push_back(L);
      continue;
    }
    for (auto b : lm->all("bottom")) {
      auto it = blob_index.find(b->scalar);
      if (it == blob_index.end())
        throw std::runtime_error("Unknown bottom blob '" + b->scalar + "' (layer '" + L.name + "')");
      L.bottoms.push_back(it->second);
    }
    for (auto t : lm->all("top")) L.tops.push_back(add_blob(t->scalar));
    layers.push_back(L);
  }
  for (int bi : inputs) {
    Blob& b = blobs[bi];
    b.kind = (b.shape.size() == 4) ? BK_INPUT_NCHW : BK_FLAT;
    if (b.name == "data") data_blob = bi;
    if (b.name == "im_info") im_info_blob = bi;
  }
  if (data_blob < 0) {
    for (int bi : inputs)
      if (blobs[bi].shape.size() == 4) { data_blob = bi; break; }
  }
  // outputs = blobs still "available" after the last layer (net.cpp:95-110,240-246): a bottom
  // takes a blob off the set, a top (also an in-place one) puts it back; the set is ordered
  // by NAME (std::set<string>), and an input nobody reads is an output too.
  {
    std::set<std::string> avail;
    for (int bi : inputs) avail.insert(blobs[bi].name);
    for (auto& L : layers) {
      if (L.type == "Input") continue;
      for (int b : L.bottoms) avail.erase(blobs[b].name);
      for (int t : L.tops) avail.insert(blobs[t].name);
    }
    for (auto& n : avail) outputs.push_back(blob_index[n]);
  }

  // ---- layer hyper-parameters, params, op assignment
  std::map<int, int> producer;  // blob -> last producing layer
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& L = layers[li];
    if (L.type == "Convolution" || L.type == "Deconvolution") {
      const PMsg* cp = L.msg->sub("convolution_param");
      if (!cp) throw std::runtime_error("layer '" + L.name + "': missing convolution_param");
      L.nout = geti(cp, "num_output", 0);
      L.k = geti(cp, "kernel_size", 1);
      L.pad = geti(cp, "pad", 0);
      L.stride = geti(cp, "stride", 1);
      L.dil = geti(cp, "dilation", 1);
      L.group = geti(cp, "group", 1);
      L.bias_term = cp->str("bias_term", "true") != "false";
      L.op = (L.type == "Convolution") ? OP_CONV : OP_DECONV;
    } else if (L.type == "ReLU") {
      if (L.bottoms.size() != 1 || L.tops.size() != 1 || L.bottoms[0] != L.tops[0])
        throw std::runtime_error("ReLU '" + L.name + "': only in-place ReLU after a convolution is supported");
      auto pit = producer.find(L.bottoms[0]);
      if (pit == producer.end() || layers[pit->second].type != "Convolution")
        throw std::runtime_error("ReLU '" + L.name + "': producer is not a Convolution");
      if (L.msg->sub("relu_param") && L.msg->sub("relu_param")->real("negative_slope", 0) != 0)
        throw std::runtime_error("ReLU negative_slope != 0 unsupported");
      // nothing may read the pre-activation value between the conv and this ReLU
      for (size_t lj = pit->second + 1; lj < li; ++lj)
        for (int b : layers[lj].bottoms)
          if (b == L.bottoms[0]) throw std::runtime_error("ReLU '" + L.name + "': blob read before activation");
      layers[pit->second].relu = 1;
      L.op = OP_SKIP;
    } else if (L.type == "Pooling") {
      const PMsg* pp = L.msg->sub("pooling_param");
      if (pp && pp->str("pool", "MAX") != "MAX") throw std::runtime_error("only MAX pooling is supported");
      L.k = geti(pp, "kernel_size", 2);
      L.stride = geti(pp, "stride", 1);
      L.pad = geti(pp, "pad", 0);
      L.op = OP_POOL;
    } else if (L.type == "Python") {
      const PMsg* py = L.msg->sub("python_param");
      if (!py || py->str("layer") != "ProposalLayer")
        throw std::runtime_error("Python layer '" + L.name + "': only ProposalLayer has a native implementation");
      L.op = OP_TAIL;
      tail_layer = (int)li;
    } else if (L.type == "Concat" || L.type == "Softmax" || L.type == "Reshape" || L.type == "Split" ||
               L.type == "Input") {
      L.op = OP_SKIP;
    } else {
      throw std::runtime_error("Unsupported layer type '" + L.type + "' (layer '" + L.name + "')");
    }
    for (int t : L.tops) producer[t] = (int)li;
  }

  // ---- the fused tail: walk back from the proposal layer
  std::set<int> fused_layers;
  if (tail_layer >= 0) {
    Layer& T = layers[tail_layer];
    if (T.bottoms.size() != 3 || T.tops.empty())
      throw std::runtime_error("ProposalLayer: expected bottoms (cls_prob, bbox_pred, im_info)");
    tail_cls_blob = T.bottoms[0];
    tail_box_blob = T.bottoms[1];
    boxes_blob = T.tops[0];
    prob_blob = T.tops.size() > 1 ? T.tops[1] : -1;
    auto prod = [&](int blob, const char* want) -> int {
      auto it = producer.find(blob);
      if (it == producer.end() || layers[it->second].type != want)
        throw std::runtime_error(std::string("tail: expected a ") + want + " producing '" + blobs[blob].name + "'");
      return it->second;
    };
    // cls branch: Reshape <- Softmax <- (Concat axis2 of 1x1 convs | Reshape <- 1x1 conv)
    const int l_rs = prod(tail_cls_blob, "Reshape");
    const int l_sm = prod(layers[l_rs].bottoms[0], "Softmax");
    fused_layers.insert(l_rs);
    fused_layers.insert(l_sm);
    int pre = layers[l_sm].bottoms[0];
    auto pit = producer.find(pre);
    if (pit == producer.end()) throw std::runtime_error("tail: dangling softmax input");
    if (layers[pit->second].type == "Concat") {
      const int l_cc = pit->second;
      if (geti(layers[l_cc].msg->sub("concat_param"), "axis", 1) != 2)
        throw std::runtime_error("tail: class-score concat must be on axis 2");
      fused_layers.insert(l_cc);
      for (int b : layers[l_cc].bottoms) tail_cls_layers.push_back(prod(b, "Convolution"));
    } else if (layers[pit->second].type == "Reshape") {
      fused_layers.insert(pit->second);
      tail_cls_layers.push_back(prod(layers[pit->second].bottoms[0], "Convolution"));
    } else {
      throw std::runtime_error("tail: unsupported class-score branch");
    }
    // box branch: Concat axis1 of 1x1 convs | single 1x1 conv
    auto bit = producer.find(tail_box_blob);
    if (bit == producer.end()) throw std::runtime_error("tail: dangling bbox input");
    if (layers[bit->second].type == "Concat") {
      if (geti(layers[bit->second].msg->sub("concat_param"), "axis", 1) != 1)
        throw std::runtime_error("tail: bbox concat must be on axis 1");
      fused_layers.insert(bit->second);
      for (int b : layers[bit->second].bottoms) tail_box_layers.push_back(prod(b, "Convolution"));
    } else if (layers[bit->second].type == "Convolution") {
      tail_box_layers.push_back(bit->second);
    } else {
      throw std::runtime_error("tail: unsupported bbox branch");
    }
    if (tail_cls_layers.size() != tail_box_layers.size())
      throw std::runtime_error("tail: class / bbox branches disagree");
    tail_heads = (int)tail_cls_layers.size();
    for (int i = 0; i < tail_heads; ++i) {
      Layer& c = layers[tail_cls_layers[i]];
      Layer& b = layers[tail_box_layers[i]];
      if (c.k != 1 || b.k != 1 || c.bottoms[0] != b.bottoms[0])
        throw std::runtime_error("tail: cls/bbox predictors must be 1x1 convs on the same head blob");
      tail_feat_blobs.push_back(c.bottoms[0]);
      fused_layers.insert(tail_cls_layers[i]);
      fused_layers.insert(tail_box_layers[i]);
    }
    const PMsg* py = T.msg->sub("python_param");
    auto ps = parse_param_str(py->str("param_str"));
    std::vector<double> fs = ps.count("feat_stride") ? ps["feat_stride"] : std::vector<double>{16};
    std::vector<double> scales = ps.count("scales") ? ps["scales"] : std::vector<double>{8, 16, 32};
    std::vector<double> ratios = ps.count("ratios") ? ps["ratios"] : std::vector<double>{0.5, 1, 2};
    std::vector<double> shifts = ps.count("shifts") ? ps["shifts"] : std::vector<double>{0};
    const int base_size = ps.count("base_size") ? (int)ps["base_size"][0] : 16;
    const bool subsampled = ps.count("subsampled") ? ps["subsampled"][0] != 0 : true;
    if (ps.count("num_feats") && ps["num_feats"][0] != 1) throw std::runtime_error("tail: num_feats != 1 unsupported");
    gen_anchors(base_size, ratios, scales, shifts, fs, anchors);
    tail_A = (int)anchors.size() / 4;
    feat_stride = (int)fs[0];
    sub_stride.assign(tail_A, 1);
    if (subsampled)
      for (int i = 0; i < tail_A; ++i) {
        const size_t idx = (size_t)i / (shifts.size() * shifts.size());
        sub_stride[i] = (int)fs[std::min(idx, fs.size() - 1)] / (int)fs[0];
      }
    if (tail_A > 8) throw std::runtime_error("tail: more than 8 anchors per cell unsupported");
    if (tail_heads != 1 && tail_heads != tail_A) throw std::runtime_error("tail: heads must be 1 or == anchors");
    const int ncls = tail_heads == 1 ? 2 * tail_A : 2, nbox = tail_heads == 1 ? 4 * tail_A : 4;
    for (int i = 0; i < tail_heads; ++i)
      if (layers[tail_cls_layers[i]].nout != ncls || layers[tail_box_layers[i]].nout != nbox)
        throw std::runtime_error("tail: predictor channel counts do not match the anchors");
    for (int li2 : fused_layers) {
      layers[li2].op = OP_SKIP;
      for (int t : layers[li2].tops) blobs[t].kind = BK_FUSED;
    }
    blobs[tail_cls_blob].kind = BK_NCHW_MAT;
    blobs[tail_box_blob].kind = BK_NCHW_MAT;
    blobs[boxes_blob].kind = BK_FLAT;
    if (prob_blob >= 0) blobs[prob_blob].kind = BK_FLAT;
  }

  // ---- channel-concat views (zero-copy): bottoms of a non-fused axis-1 Concat live inside the top
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& L = layers[li];
    if (L.type != "Concat" || fused_layers.count((int)li)) continue;
    if (geti(L.msg->sub("concat_param"), "axis", 1) != 1)
      throw std::runtime_error("Concat '" + L.name + "': only channel concat is supported outside the tail");
    for (int b : L.bottoms) {
      if (blobs[b].owner >= 0 || std::count(inputs.begin(), inputs.end(), b))
        throw std::runtime_error("Concat '" + L.name + "': bottom already aliased");
      blobs[b].owner = L.tops[0];
    }
  }

  // ---- params (shapes need channel counts: run shape inference once)
  infer_shapes();
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& L = layers[li];
    if (L.type != "Convolution" && L.type != "Deconvolution") continue;
    const int cin = blobs[L.bottoms[0]].shape[1];
    std::vector<std::vector<int>> shapes;
    if (L.type == "Convolution") {
      if (L.group != 1) throw std::runtime_error("Convolution '" + L.name + "': group != 1 unsupported");
      if (L.stride != 1) throw std::runtime_error("Convolution '" + L.name + "': stride != 1 unsupported");
      if (!((L.k == 3 && L.pad == L.dil) || (L.k == 1 && L.pad == 0)))
        throw std::runtime_error("Convolution '" + L.name + "': only 3x3 pad==dilation and 1x1 pad 0 are supported");
      shapes.push_back({L.nout, cin, L.k, L.k});
    } else {
      if (L.group != cin || L.nout != cin)
        throw std::runtime_error("Deconvolution '" + L.name + "': only depthwise (group == channels) is supported");
      shapes.push_back({cin, 1, L.k, L.k});
    }
    if (L.bias_term) shapes.push_back({L.nout});
    auto pspecs = L.msg->all("param");
    for (size_t pi = 0; pi < shapes.size(); ++pi) {
      std::string pname = (pi < pspecs.size() && pspecs[pi]->msg) ? pspecs[pi]->msg->str("name") : "";
      std::shared_ptr<ParamBlob> pb;
      if (clone_src) {
        pb = clone_src->layers[li].params[pi];
      } else if (!pname.empty() && shared_params.count(pname)) {
        pb = shared_params[pname];
        if (pb->shape != shapes[pi]) throw std::runtime_error("Shared parameter '" + pname + "' shape mismatch");
      } else {
        pb = std::make_shared<ParamBlob>();
        pb->shape = shapes[pi];
        pb->host.assign(pb->count(), 0.f);
        if (!pname.empty()) shared_params[pname] = pb;
      }
      L.params.push_back(pb);
    }
    if (L.type == "Convolution")
      L.kclass = conv_kernel_class(cin, L.nout, L.k, L.pad, L.dil, blobs[L.bottoms[0]].kind == BK_INPUT_NCHW);
  }
  // ---- conv -> MAX 2x2/2 pool pairs that the fused (detect) path runs as one kernel
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& P = layers[li];
    if (P.op != OP_POOL || P.k != 2 || P.stride != 2 || P.pad != 0) continue;
    const int x = P.bottoms[0];
    int prod = -1, others = 0;
    for (size_t lj = 0; lj < layers.size(); ++lj) {
      if (lj == li) continue;
      Layer& Q = layers[lj];
      if (Q.type == "Convolution" && !Q.tops.empty() && Q.tops[0] == x) prod = (int)lj;
      if (Q.op == OP_SKIP && Q.type == "ReLU") continue;  // the in-place ReLU is part of the conv
      for (int bb : Q.bottoms)
        if (bb == x) ++others;
    }
    if (prod < 0 || layers[prod].op != OP_CONV || layers[prod].kclass != 0 || !layers[prod].relu) continue;
    if (blobs[x].owner >= 0 || blobs[P.tops[0]].owner >= 0) continue;
    layers[prod].fuse_pool = (int)li;
    layers[prod].pool_only = (others == 0);
    P.fused_into = prod;
  }
  // ---- first-layer conv (on the raw image) that the split-fp16 kernel of the NEXT conv can compute in place
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& F = layers[li];
    if (F.op != OP_CONV || F.kclass != 1 || !F.relu || F.k != 3 || F.pad != 1 || F.dil != 1 || F.nout != 64) continue;
    if (blobs[F.bottoms[0]].shape.size() != 4 || blobs[F.bottoms[0]].shape[1] != 3) continue;
    const int x = F.tops[0];
    int next = -1, readers = 0;
    for (size_t lj = 0; lj < layers.size(); ++lj) {
      Layer& Q = layers[lj];
      if (lj == li || (Q.op == OP_SKIP && Q.type == "ReLU")) continue;
      for (int bb : Q.bottoms)
        if (bb == x) { ++readers; next = (int)lj; }
    }
    if (readers != 1 || layers[next].op != OP_CONV || layers[next].kclass != 0) continue;
    Layer& N = layers[next];
    if (N.k != 3 || N.dil != 1 || !conv_f16x3_eligible(64, N.nout, N.k, N.pad, N.dil) || blobs[x].owner >= 0) continue;
    N.first_src = (int)li;
    F.first_dst = next;
  }
  // ---- dilation-1 / 2 / 4 convolutions over one bottom with shared parameter blobs: the shared-weight heads
  for (size_t li = 0; li < layers.size(); ++li) {
    Layer& A = layers[li];
    if (A.op != OP_CONV || A.kclass != 0 || A.k != 3 || A.dil != 1 || A.pad != 1 || A.nout != 128 || A.params.empty() ||
        A.fuse_pool >= 0 || A.first_src >= 0)
      continue;
    int d2 = -1, d4 = -1;
    for (size_t lj = li + 1; lj < layers.size(); ++lj) {
      Layer& Q = layers[lj];
      if (Q.op != OP_CONV || Q.kclass != 0 || Q.k != 3 || Q.pad != Q.dil || Q.nout != A.nout || Q.bottoms[0] != A.bottoms[0] ||
          Q.params.size() != A.params.size() || Q.relu != A.relu || Q.fuse_pool >= 0)
        continue;
      bool same = true;
      for (size_t pi = 0; pi < A.params.size(); ++pi) same = same && Q.params[pi] == A.params[pi];
      if (!same) continue;
      if (Q.dil == 2 && d2 < 0) d2 = (int)lj;
      if (Q.dil == 4 && d4 < 0) d4 = (int)lj;
    }
    if (d2 < 0 || d4 < 0) continue;
    A.heads3_d2 = d2;
    A.heads3_d4 = d4;
    layers[d2].heads3_lead = layers[d4].heads3_lead = (int)li;
  }
  // ---- blobs the fused split-fp16 path keeps in the pre-split activation format: produced by a split-fp16 conv
  //      (or by the pool fused into its epilogue) and read ONLY by convs that run on the 4-wave kernel
  {
    static const bool split_act = !(getenv("SHF_F16X3_SPLIT_ACT") && atoi(getenv("SHF_F16X3_SPLIT_ACT")) == 0);
    // (the static half of the launch-time predicates conv_f16x3_group_is_dual / _dilated_w4 / _k1_gemm: what a reader writes
    // -- its top and the top of a pool fused into it -- must be a 16-byte aligned channel view, and the GEMM kernel takes no
    // fused pool; the dynamic half -- an input of 4 GiB or more -- fails the launch with the layer's name)
    auto aligned_view = [&](int bi_) {
      const Blob& b_ = blobs[bi_];
      const Blob& ob_ = blobs[b_.owner >= 0 ? b_.owner : bi_];
      return ob_.shape.size() == 4 && ob_.shape[1] % 4 == 0 && b_.coff % 4 == 0;
    };
    auto w4_reader = [&](const Layer& Q, int cin) {
      const bool dil_ok = Q.dil == 1 || Q.dil == 2 || Q.dil == 4;   // (the heads: family's DIL form / the three-heads kernel)
      if (Q.op != OP_CONV || Q.tops.empty() || !aligned_view(Q.tops[0])) return false;
      if (Q.fuse_pool >= 0 && !aligned_view(layers[Q.fuse_pool].tops[0])) return false;
      if (Q.op == OP_CONV && Q.kclass == 0 && Q.k == 1 && Q.pad == 0 && Q.first_src < 0)   // 1x1 layers on the GEMM kernel
        return Q.fuse_pool < 0 && conv_f16x3_k1_gemm_shape(cin, Q.nout) && conv_f16x3_eligible(cin, Q.nout, Q.k, Q.pad, Q.dil);
      return Q.op == OP_CONV && Q.kclass == 0 && Q.k == 3 && dil_ok && Q.pad == Q.dil && cin % 32 == 0 && Q.nout % 128 == 0 &&
             Q.first_src < 0 && conv_f16x3_uses_w4(cin) && conv_f16x3_eligible(cin, Q.nout, Q.k, Q.pad, Q.dil);
    };
    for (size_t bi = 0; bi < blobs.size() && split_act; ++bi) {
      Blob& B = blobs[bi];
      if (B.owner >= 0 || B.kind != BK_NHWC || B.shape.size() != 4) continue;
      if (std::count(tail_feat_blobs.begin(), tail_feat_blobs.end(), (int)bi)) continue;
      bool concat_member = false;
      for (auto& O : blobs) concat_member = concat_member || O.owner == (int)bi;
      if (concat_member) continue;
      // producer: a kclass-0 conv eligible for a split-fp16 kernel, directly or through its fused pool
      int prod = -1;
      for (size_t lj = 0; lj < layers.size(); ++lj) {
        Layer& Q = layers[lj];
        if (Q.op == OP_CONV && Q.kclass == 0 && !Q.tops.empty() && Q.tops[0] == (int)bi) prod = (int)lj;
        if (Q.op == OP_POOL && Q.fused_into >= 0 && !Q.tops.empty() && Q.tops[0] == (int)bi) prod = Q.fused_into;
      }
      if (prod < 0) continue;
      const Layer& Pq = layers[prod];
      const int pcin = blobs[Pq.bottoms[0]].shape.size() == 4 ? blobs[Pq.bottoms[0]].shape[1] : 0;
      if (!conv_f16x3_eligible(Pq.first_src >= 0 ? 64 : pcin, Pq.nout, Pq.k, Pq.pad, Pq.dil)) continue;
      int readers = 0;
      bool all_w4 = true;
      for (size_t lj = 0; lj < layers.size(); ++lj) {
        Layer& Q = layers[lj];
        if (Q.op == OP_SKIP && Q.type == "ReLU") continue;                       // in-place, part of the conv
        if (Q.op == OP_POOL && Q.fused_into >= 0 && Q.bottoms[0] == (int)bi) continue;  // folded into the producer
        for (int bb : Q.bottoms)
          if (bb == (int)bi) {
            ++readers;
            all_w4 = all_w4 && w4_reader(Q, B.shape[1]);
          }
      }
      B.split_fused = readers > 0 && all_w4;
    }
  }
  alloc_buffers();
  amax_slots.ensure(std::max<size_t>(blobs.size(), 1) * 4);
  fill_now(amax_slots.p, 0, std::max<size_t>(blobs.size(), 1) * 4);
  if (clone_src) {
    wgen = clone_src->wgen;
    return;
  }
  if (caffemodel && caffemodel[0]) load_caffemodel(caffemodel);
  for (size_t li = 0; li < layers.size(); ++li) commit_params((int)li);
}

void shf_net::infer_shapes() {
  for (auto& L : layers) {
    if (L.type == "Input") continue;
    auto& bs = blobs[L.bottoms.empty() ? 0 : L.bottoms[0]].shape;
    if (L.type == "Convolution") {
      if (bs.size() != 4) throw std::runtime_error("Convolution '" + L.name + "': 4-D bottom expected");
      blobs[L.tops[0]].shape = {bs[0], L.nout, conv_out(bs[2], L.k, L.pad, L.stride, L.dil),
                                conv_out(bs[3], L.k, L.pad, L.stride, L.dil)};
    } else if (L.type == "Deconvolution") {
      blobs[L.tops[0]].shape = {bs[0], L.nout, L.stride * (bs[2] - 1) + L.k - 2 * L.pad,
                                L.stride * (bs[3] - 1) + L.k - 2 * L.pad};
    } else if (L.type == "ReLU" || L.type == "Softmax" || L.type == "Split") {
      for (int t : L.tops) blobs[t].shape = bs;
    } else if (L.type == "Pooling") {
      int ho = (int)std::ceil((bs[2] + 2 * L.pad - L.k) / (double)L.stride) + 1;
      int wo = (int)std::ceil((bs[3] + 2 * L.pad - L.k) / (double)L.stride) + 1;
      if (L.pad) {
        if ((ho - 1) * L.stride >= bs[2] + L.pad) --ho;
        if ((wo - 1) * L.stride >= bs[3] + L.pad) --wo;
      }
      blobs[L.tops[0]].shape = {bs[0], bs[1], ho, wo};
    } else if (L.type == "Concat") {
      const int axis = geti(L.msg->sub("concat_param"), "axis", 1);
      std::vector<int> s = bs;
      int sum = 0;
      int off = 0;
      for (int b : L.bottoms) {
        if (blobs[b].owner == L.tops[0]) blobs[b].coff = off;
        off += blobs[b].shape[axis];
        sum += blobs[b].shape[axis];
      }
      s[axis] = sum;
      blobs[L.tops[0]].shape = s;
    } else if (L.type == "Reshape") {
      const PMsg* rp = L.msg->sub("reshape_param");
      std::vector<int> dims;
      if (rp && rp->sub("shape"))
        for (auto d : rp->sub("shape")->all("dim")) dims.push_back(atoi(d->scalar.c_str()));
      std::vector<int> out;
      int infer = -1;
      long total = 1, known = 1;
      for (int d : bs) total *= d;
      for (size_t i = 0; i < dims.size(); ++i) {
        if (dims[i] == 0) out.push_back(bs[i]);
        else if (dims[i] == -1) { infer = (int)i; out.push_back(1); }
        else out.push_back(dims[i]);
      }
      for (int d : out) known *= d;
      if (infer >= 0) out[infer] = (int)(total / std::max<long>(known, 1));
      blobs[L.tops[0]].shape = out;
    } else if (L.type == "Python") {
      if (blobs[L.tops[0]].shape.size() != 2) blobs[L.tops[0]].shape = {1, 5};
      if (L.tops.size() > 1 && blobs[L.tops[1]].shape.size() != 2) blobs[L.tops[1]].shape = {1, 2};
    }
  }
  if (data_blob >= 0) last_data_shape = blobs[data_blob].shape;
}

void shf_net::alloc_buffers() {
  for (size_t i = 0; i < blobs.size(); ++i) {
    Blob& b = blobs[i];
    if (b.kind == BK_FUSED) continue;
    if (b.owner >= 0) continue;  // view into a concat buffer
    if (b.kind == BK_FLAT && ((int)i == boxes_blob || (int)i == prob_blob)) continue;  // sized by the tail
    b.dev.ensure(std::max<size_t>(b.count(), 1) * sizeof(float));
  }
  if (tail_layer >= 0) {
    Blob& f = blobs[tail_feat_blobs[0]];
    ensure_tail_workspace((size_t)f.shape[2] * f.shape[3] * tail_A);
  }
}

// tail workspace + proposal output blobs for `total` anchors (grow-only)
void shf_net::ensure_tail_workspace(size_t total) {
  size_t npad = 1;
  while (npad < total) npad <<= 1;
  tw_logits.ensure(total * 6 * 4);
  tw_rec.ensure(total * 6 * 4);
  tw_keys.ensure(std::max<size_t>(npad, 16384) * 8);
  tw_counters.ensure(64);
  tw.logits = (float*)tw_logits.p;
  tw.rec = (float*)tw_rec.p;
  tw.keys = (unsigned long long*)tw_keys.p;
  tw.counters = (int*)tw_counters.p;
  tw.amax = conv_mode >= 1 && conv_mode != 4 ? (unsigned*)amax_slots.p : nullptr;   // (the tail's reset kernel zeroes the slots for the next pass)
  tw.n_amax = (int)blobs.size();
  tw.cap_anchors = total;
  tw.cap_keys = npad;
  const size_t rmax = (pre_nms_topN > 0) ? std::min<size_t>(total, (size_t)pre_nms_topN) : total;
  blobs[boxes_blob].dev.ensure(std::max<size_t>(rmax, 1) * 5 * 4);
  if (prob_blob >= 0) blobs[prob_blob].dev.ensure(std::max<size_t>(rmax, 1) * 2 * 4);
}

void shf_net::build_tail_weights() {
  if (tail_layer < 0) return;
  tail_Cf = blobs[tail_feat_blobs[0]].shape[1];
  const int A = tail_A, Cf = tail_Cf;
  std::vector<float> W((size_t)A * 6 * Cf, 0.f), B((size_t)A * 6, 0.f);
  for (int a = 0; a < A; ++a) {
    const int h = tail_heads == 1 ? 0 : a;
    Layer& c = layers[tail_cls_layers[h]];
    Layer& b = layers[tail_box_layers[h]];
    const float* cw = c.params[0]->host.data();
    const float* bw = b.params[0]->host.data();
    const float* cb = c.params.size() > 1 ? c.params[1]->host.data() : nullptr;
    const float* bb = b.params.size() > 1 ? b.params[1]->host.data() : nullptr;
    for (int cls = 0; cls < 2; ++cls) {
      // plain template: cls_score channel = cls*A + a (Reshape (0,2,-1,0)); dilation template: channel = cls
      const int row = tail_heads == 1 ? cls * A + a : cls;
      memcpy(&W[((size_t)a * 6 + cls) * Cf], cw + (size_t)row * Cf, Cf * sizeof(float));
      B[a * 6 + cls] = cb ? cb[row] : 0.f;
    }
    for (int j = 0; j < 4; ++j) {
      const int row = tail_heads == 1 ? a * 4 + j : j;
      memcpy(&W[((size_t)a * 6 + 2 + j) * Cf], bw + (size_t)row * Cf, Cf * sizeof(float));
      B[a * 6 + 2 + j] = bb ? bb[row] : 0.f;
    }
  }
  tail_W.ensure(W.size() * 4);
  tail_b.ensure(B.size() * 4);
  HIP_THROW(hipMemcpy(tail_W.p, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  HIP_THROW(hipMemcpy(tail_b.p, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  tail_w_dirty = false;
  tail_gen = *wgen;
}

void shf_net::commit_params(int li) {
  Layer& L = layers[li];
  if (L.params.empty()) return;
  // the dual-tile family's weight pack (16-channel slabs, unscaled low parts): its 3x3 layers, and the 1x1 GEMM kernel's
  auto wants_family_pack = [](const Layer& Q, const ParamBlob& w) {
    if (Q.k == 1) return Q.pad == 0 && conv_f16x3_k1_gemm_shape(w.shape[1], w.shape[0]);
    return Q.k == 3 && (Q.dil == 1 || Q.dil == 2 || Q.dil == 4) && conv_f16x3_uses_w4(w.shape[1]) &&
           w.shape[0] % 128 == 0 && w.shape[1] % 32 == 0;
  };
  // the raw / packed tensors are shared by every lane cloned from this net: nothing may be in flight on any stream
  HIP_THROW(hipDeviceSynchronize());
  const bool in_tail = std::count(tail_cls_layers.begin(), tail_cls_layers.end(), li) ||
                       std::count(tail_box_layers.begin(), tail_box_layers.end(), li);
  for (size_t pi = 0; pi < L.params.size(); ++pi) {
    ParamBlob& p = *L.params[pi];
    p.raw.ensure(p.count() * 4);
    HIP_THROW(hipMemcpy(p.raw.p, p.host.data(), p.count() * 4, hipMemcpyHostToDevice));
    if (pi == 0 && L.type == "Convolution" && L.kclass == 1) {
      // first layer: (Cout, Cin*k*k) -> (Cin*k*k, Cout) so a wave's 16 output channels are one uniform run
      const int co = p.shape[0], K = (int)(p.count() / p.shape[0]);
      std::vector<float> t(p.count());
      for (int o = 0; o < co; ++o)
        for (int r = 0; r < K; ++r) t[(size_t)r * co + o] = p.host[(size_t)o * K + r];
      p.first_t.ensure(t.size() * 4);
      HIP_THROW(hipMemcpy(p.first_t.p, t.data(), t.size() * 4, hipMemcpyHostToDevice));
      if (co == 64 && K == 27) {  // the shape the fused producer/consumer kernel computes on the matrix cores
        std::vector<uint16_t> fr(kFirstConvFragHalfs);
        pack_first_conv_frags(p.host.data(), fr.data());
        p.first_frag.ensure(fr.size() * 2);
        HIP_THROW(hipMemcpy(p.first_frag.p, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        p.bf_stale = true;
        if (conv_mode == 4) {
          pack_first_conv_frags(p.host.data(), fr.data(), true);
          p.first_frag_b.ensure(fr.size() * 2);
          HIP_THROW(hipMemcpy(p.first_frag_b.p, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
          p.bf_stale = false;
        }
      }
    }
    if (pi == 0 && L.type == "Convolution" && L.kclass == 0 && !in_tail) {
      std::vector<float> packed(p.count());
      pack_conv_weights(p.host.data(), p.shape[0], p.shape[1], p.shape[2], packed.data());
      p.packed.ensure(packed.size() * 4);
      HIP_THROW(hipMemcpy(p.packed.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      // a commit in fp32 mode leaves the split-fp16 packs behind: shf_net_set_conv_mode re-packs them on the way back
      p.split_stale = p.packed16.p != nullptr;
      p.bf_stale = true;
      if (conv_mode == 4 && conv_f16x3_eligible(p.shape[1], p.shape[0], L.k, L.pad, L.dil)) {
        // bf16 mode: hi = bf16(w) bit patterns in the same layouts (no range check: bf16 has fp32's exponent range)
        std::vector<uint16_t> sp(split16_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
        pack_conv_weights_split16(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sp.data(), true);
        p.packed16b.ensure(sp.size() * 2);
        HIP_THROW(hipMemcpy(p.packed16b.p, sp.data(), sp.size() * 2, hipMemcpyHostToDevice));
        if (L.first_src >= 0 && p.shape[0] == 64 && p.shape[1] == 64) {   // the fused first pair's own pack
          std::vector<uint16_t> sr(split16r_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
          pack_conv_weights_split16r(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sr.data(), true);
          p.packed16rb.ensure(sr.size() * 2);
          HIP_THROW(hipMemcpy(p.packed16rb.p, sr.data(), sr.size() * 2, hipMemcpyHostToDevice));
        }
        if (wants_family_pack(L, p)) {
          std::vector<uint16_t> sh(split16h_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
          pack_conv_weights_split16h(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sh.data(), true);
          p.packed16hb.ensure(sh.size() * 2);
          HIP_THROW(hipMemcpy(p.packed16hb.p, sh.data(), sh.size() * 2, hipMemcpyHostToDevice));
        }
        p.bf_stale = false;
      }
      if (conv_mode >= 1 && conv_mode <= 3 && conv_f16x3_eligible(p.shape[1], p.shape[0], L.k, L.pad, L.dil)) {
        // split-fp16 keeps hi = fp16(w): a weight beyond the fp16 range would become inf (the reference is fp32
        // everywhere, caffe/python/caffe/_caffe.cpp:46-48) -- refuse the mode instead of computing garbage
        for (float w : p.host)
          if (!(std::fabs(w) <= 65504.f))
            throw std::runtime_error("layer '" + L.name + "': a weight is outside the fp16 range (|w| > 65504 or not "
                                     "finite); the split-fp16 conv mode cannot represent it -- use conv mode fp32");
        std::vector<uint16_t> sp(split16_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
        pack_conv_weights_split16(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sp.data());
        p.packed16.ensure(sp.size() * 2);
        HIP_THROW(hipMemcpy(p.packed16.p, sp.data(), sp.size() * 2, hipMemcpyHostToDevice));
        if (L.first_src >= 0 && p.shape[0] == 64 && p.shape[1] == 64) {   // the fused first pair's own pack
          std::vector<uint16_t> sr(split16r_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
          pack_conv_weights_split16r(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sr.data());
          p.packed16r.ensure(sr.size() * 2);
          HIP_THROW(hipMemcpy(p.packed16r.p, sr.data(), sr.size() * 2, hipMemcpyHostToDevice));
        }
        p.split_stale = false;
        if (wants_family_pack(L, p)) {
          std::vector<uint16_t> sh(split16h_conv_weight_halfs(p.shape[0], p.shape[1], p.shape[2]));
          p.wscale_inv = pack_conv_weights_split16h(p.host.data(), p.shape[0], p.shape[1], p.shape[2], sh.data());
          p.packed16h.ensure(sh.size() * 2);
          HIP_THROW(hipMemcpy(p.packed16h.p, sh.data(), sh.size() * 2, hipMemcpyHostToDevice));
        }
      }
    }
    p.dirty = false;
  }
  if (in_tail) tail_w_dirty = true;
  ++*wgen;
}

void shf_net::load_caffemodel(const std::string& path) {
  // CopyTrainedLayersFrom (net.cpp:733-768): match by layer NAME, check shapes, copy blobs
  auto src = read_caffemodel(path);
  for (auto& sl : src) {
    for (auto& L : layers) {
      if (L.name != sl.name || L.params.empty()) continue;
      if (sl.blobs.size() != L.params.size())
        throw std::runtime_error("Incompatible number of blobs for layer " + L.name);
      for (size_t i = 0; i < L.params.size(); ++i) {
        ParamBlob& p = *L.params[i];
        if (sl.blobs[i].data.size() != p.count())
          throw std::runtime_error("Cannot copy param " + std::to_string(i) + " weights from layer '" + L.name +
                                   "'; shape mismatch.");
        std::copy(sl.blobs[i].data.begin(), sl.blobs[i].data.end(), p.host.begin());
        p.dirty = true;
      }
    }
  }
}

static double conv_flops(const Layer& L, const std::vector<int>& in, const std::vector<int>& out) {
  return 2.0 * out[0] * out[2] * out[3] * (double)L.nout * in[1] * L.k * L.k;
}

// the proposal stage's arguments for the current shapes (also sizes the workspace: pre_nms_topN is shared with the
// other lanes and may have grown)
TailArgs shf_net::tail_args(float im_h, float im_w, float im_scale, bool fused_path) {
  TailArgs t;
  t.A = tail_A; t.heads = tail_heads; t.Cf = tail_Cf;
  for (int i = 0; i < tail_heads; ++i) t.feat[i] = view_of(tail_feat_blobs[i]);
  t.wcls[0] = (const float*)tail_W.p;
  t.bcls[0] = (const float*)tail_b.p;
  t.h = blobs[tail_feat_blobs[0]].shape[2];
  t.w = blobs[tail_feat_blobs[0]].shape[3];
  for (int i = 0; i < tail_A * 4; ++i) t.anchors[i] = (float)anchors[i];
  for (int i = 0; i < tail_A; ++i) t.sub_stride[i] = sub_stride[i];
  t.feat_stride = feat_stride;
  t.im_h = im_h; t.im_w = im_w; t.im_scale = im_scale;
  t.min_size = min_size; t.score_thresh = score_thresh; t.pre_nms_topN = pre_nms_topN;
  if (materialize_tail && !fused_path) {
    t.cls_prob_reshape_nchw = (float*)blobs[tail_cls_blob].dev.p;
    t.bbox_pred_nchw = (float*)blobs[tail_box_blob].dev.p;
  }
  ensure_tail_workspace((size_t)t.h * t.w * tail_A);
  return t;
}

void shf_net::forward_ops(bool fused_path, float im_h, float im_w, float im_scale, hipStream_t s_override,
                          Prof* prof_override, int only_layer, ConvArgs* collect) {
  if (tail_w_dirty || tail_gen != *wgen) build_tail_weights();
  hipStream_t st = s_override ? s_override : stream;
  Prof& pf = prof_override ? *prof_override : prof;
  int heads3_done = -1;   // index of a dilation-1 head whose launch also wrote its dilation-2 / -4 siblings
  // the three shared-weight heads of this unit in one launch (conv_f16x3_h3.h): `a` = the dilation-1 layer's arguments
  auto try_heads3 = [&](int li, const ConvArgs& a, hipStream_t st_, Prof& pf_) {
    const Layer& L1 = layers[li];
    ConvArgs a2, a4;
    forward_ops(fused_path, im_h, im_w, im_scale, st_, &pf_, L1.heads3_d2, &a2);
    forward_ops(fused_path, im_h, im_w, im_scale, st_, &pf_, L1.heads3_d4, &a4);
    if (!conv_f16x3_group_is_heads3(&a, &a2, &a4, 1)) return false;
    const double fl = 3.0 * conv_flops(L1, blobs[L1.bottoms[0]].shape, blobs[L1.tops[0]].shape);
    const double by = 4.0 * (blobs[L1.bottoms[0]].count() + 3.0 * blobs[L1.tops[0]].count() + L1.params[0]->count());
    ProfScope ps(pf_, st_, PC_CONV_F16X3_H3, fl, by);
    CHECK_RC(launch_conv_f16x3_heads3(&a, &a2, &a4, 1, st_));
    return true;
  };
  for (size_t li = 0; li < layers.size(); ++li) {
    if (only_layer >= 0 && (int)li != only_layer) continue;
    Layer& L = layers[li];
    switch (L.op) {
      case OP_SKIP: break;
      case OP_CONV: {
        ConvArgs a;
        Blob& ib = blobs[L.bottoms[0]];
        a.out = view_of(L.tops[0]);
        a.k = L.k; a.dil = L.dil; a.pad = L.pad; a.relu = L.relu;
        a.bias = L.params.size() > 1 ? (const float*)L.params[1]->raw.p : nullptr;
        a.wraw = (const float*)L.params[0]->raw.p;
        a.wpacked = (const float*)L.params[0]->packed.p;
        a.wfirst = (const float*)L.params[0]->first_t.p;
        const bool bf = conv_mode == 4;
        const bool split16 = conv_mode >= 1 && L.kclass == 0 && (bf ? L.params[0]->packed16b.p : L.params[0]->packed16.p) &&
                             conv_f16x3_eligible(ib.shape[1], L.nout, L.k, L.pad, L.dil);
        a.wsplit16 = split16 ? (bf ? L.params[0]->packed16b.p : L.params[0]->packed16.p) : nullptr;
        a.wsplit16h = split16 ? (bf ? L.params[0]->packed16hb.p : L.params[0]->packed16h.p) : nullptr;
        a.wsplit16r = split16 ? (bf ? L.params[0]->packed16rb.p : L.params[0]->packed16r.p) : nullptr;
        a.wscale_inv = bf ? 1.f : L.params[0]->wscale_inv;
        a.bf16 = bf && split16 ? 1 : 0;
        if (fused_path && L.fuse_pool >= 0) {
          a.pool = view_of(layers[L.fuse_pool].tops[0]);
          a.write_main = L.pool_only ? 0 : 1;
          a.pool_split = split16 && !bf && blobs[layers[L.fuse_pool].tops[0]].split_fused;
        }
        // split-fp16 mode: every producer of a map that a split-fp16 conv may read guards the fp16 range
        // (bf16 has fp32's exponent range: no fp16 range guard; amax_slot() is null in that mode)
        a.range_flag = conv_mode >= 1 && !bf ? (flag_ptr ? flag_ptr : (int*)range_flag.p) : nullptr;
        a.in_amax = amax_slot(L.bottoms[0]);
        a.out_amax = amax_slot(L.tops[0]);
        if (a.pool.p) a.pool_amax = amax_slot(layers[L.fuse_pool].tops[0]);
        if (split16) {  // how many of the three fp16 products this layer forms
          a.nprod = conv_mode == 1 ? 3 : conv_mode == 2 ? 2 : 1;   // (modes 3 "f16" and 4 "bf16": one product)
          auto it = sh->layer_products.find(L.name);
          if (it != sh->layer_products.end()) a.nprod = it->second;
        }
        if (fused_path && split16 && !bf) {   // (bf16 mode keeps fp32 activations in HBM)
          a.in_split = ib.split_fused;
          a.out_split = blobs[L.tops[0]].split_fused;
        }
        // bf16 mode has the fused first pair on the producer/consumer kernel only: without its preconditions conv1_1
        // runs on its own kernel and this layer as a plain bf16 convolution (the fp16 modes fall back to the 8-wave
        // FUSE1 form instead)
        auto pair_fused = [&](const Layer& F1, const Layer& F2) {
          if (!bf) return true;
          return conv_f16x3_uses_pc() && F1.params[0]->first_frag_b.p != nullptr && F1.nout == 64 && F2.nout == 64;
        };
        if (fused_path && split16 && L.first_src >= 0 && pair_fused(layers[L.first_src], L)) {
          Layer& F = layers[L.first_src];
          Blob& db = blobs[F.bottoms[0]];
          a.img = db.ext_dev ? db.ext_dev : (const float*)db.dev.p;
          a.w1t = (const float*)F.params[0]->first_t.p;
          a.w1f = bf ? F.params[0]->first_frag_b.p : F.params[0]->first_frag.p;
          a.b1 = F.params.size() > 1 ? (const float*)F.params[1]->raw.p : nullptr;
        }
        if (fused_path && conv_mode >= 1 && L.first_dst >= 0 &&
            (bf ? layers[L.first_dst].params[0]->packed16b.p : layers[L.first_dst].params[0]->packed16.p) &&
            pair_fused(L, layers[L.first_dst]))
          break;  // computed inside the next conv's halo staging
        const double fl = conv_flops(L, ib.shape, blobs[L.tops[0]].shape);
        const double by = 4.0 * (ib.count() + blobs[L.tops[0]].count() + L.params[0]->count());
        if (L.kclass == 1) {
          a.in.B = ib.shape[0]; a.in.C = ib.shape[1]; a.in.H = ib.shape[2]; a.in.W = ib.shape[3];
          const float* src = ib.ext_dev ? ib.ext_dev : (const float*)ib.dev.p;
          ProfScope ps(pf, st, PC_CONV_FIRST, fl, by);
          CHECK_RC(launch_conv_first(src, a, st));
        } else {
          a.in = view_of(L.bottoms[0]);
          if (L.kclass == 0 && collect) {
            *collect = a;  // grouped launch: the caller batches this layer over several units
          } else if (L.kclass == 0 && split16 && L.heads3_lead >= 0 && heads3_done == L.heads3_lead) {
            // written by the dilation-1 sibling's launch (the three shared-weight heads in one kernel)
          } else if (L.kclass == 0 && split16 && L.heads3_d2 >= 0 && only_layer < 0 && try_heads3((int)li, a, st, pf)) {
            heads3_done = (int)li;
          } else if (L.kclass == 0 && split16) {
            if (conv_f16x3_group_is_dual(&a, 1)) {
              SubProf sp{&pf, st, fl, by, {}};
              a.sub_hook = &SubProf::hook;
              a.sub_ctx = &sp;
              CHECK_RC_LAYER(launch_conv_f16x3_group(&a, 1, st), L.name);
            } else {
              ProfScope ps(pf, st, f16x3_prof_class(a, L.nout), fl, by);
              CHECK_RC_LAYER(launch_conv_f16x3_group(&a, 1, st), L.name);
            }
          } else if (L.kclass == 0) {
            const int pc = conv_prof_class(L.k, L.dil, L.nout);
            ProfScope ps(pf, st, pc, fl, by);
            CHECK_RC(launch_conv_mfma(a, st));
          } else {
            ProfScope ps(pf, st, PC_CONV_DIRECT, fl, by);
            CHECK_RC(launch_conv_direct(a, st));
          }
        }
        break;
      }
      case OP_POOL: {
        if (fused_path && L.fused_into >= 0) break;  // done by the producing conv's epilogue
        ProfScope ps(pf, st, PC_POOL, 0, 4.0 * (blobs[L.bottoms[0]].count() + blobs[L.tops[0]].count()));
        CHECK_RC(launch_maxpool(view_of(L.bottoms[0]), view_of(L.tops[0]), L.k, L.stride, L.pad, st));
        if (amax_slot(L.tops[0]))  // max |pooled| <= max |input|: the bound serves as the pooled blob's activation exponent
          CHECK_RC(launch_amax_raise(amax_slot(L.tops[0]), amax_slot(L.bottoms[0]), st));
        break;
      }
      case OP_DECONV: {
        ProfScope ps(pf, st, PC_DECONV, 2.0 * blobs[L.tops[0]].count() * 4,
                     4.0 * (blobs[L.bottoms[0]].count() + blobs[L.tops[0]].count()));
        CHECK_RC(launch_deconv_depthwise(view_of(L.bottoms[0]), view_of(L.tops[0]), (const float*)L.params[0]->raw.p,
                                         L.params.size() > 1 ? (const float*)L.params[1]->raw.p : nullptr, L.k,
                                         L.stride, L.pad, st,
                                         conv_mode >= 1 && conv_mode != 4 ? (flag_ptr ? flag_ptr : (int*)range_flag.p) : nullptr,
                                         amax_slot(L.tops[0])));
        break;
      }
      case OP_TAIL: {
        TailArgs t = tail_args(im_h, im_w, im_scale, fused_path);
        const double K = (double)t.h * t.w;
        ProfScope ps(pf, st, PC_TAIL, 2.0 * K * tail_A * 6 * tail_Cf,
                     4.0 * K * (tail_heads * tail_Cf + tail_A * 18));
        if (fused_path && !ev_logits) HIP_THROW(hipEventCreateWithFlags(&ev_logits, hipEventDisableTiming));
        CHECK_RC(launch_tail(t, tw, (float*)blobs[boxes_blob].dev.p,
                             prob_blob >= 0 ? (float*)blobs[prob_blob].dev.p : (float*)tw_rec.p, st,
                             fused_path ? ev_logits : nullptr, 0));
        if (fused_path) logits_done = ev_logits;
        break;
      }
    }
  }
}

void shf_net::forward() {
  if (data_blob >= 0 && blobs[data_blob].shape != last_data_shape) {
    infer_shapes();
    alloc_buffers();
  }
  for (int bi : inputs) {
    Blob& b = blobs[bi];
    b.ext_dev = nullptr;
    if (b.host_newer && b.host.p) {
      b.dev.ensure(b.count() * 4);
      HIP_THROW(hipMemcpyAsync(b.dev.p, b.host.p, b.count() * 4, hipMemcpyHostToDevice, stream));
      b.host_newer = false;
    }
  }
  float ii[3] = {0, 0, 1};
  if (im_info_blob >= 0 && blobs[im_info_blob].host.p && blobs[im_info_blob].count() >= 3)
    memcpy(ii, blobs[im_info_blob].host.p, 12);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (conv_mode >= 1) HIP_THROW(hipMemsetAsync(range_flag.p, 0, 4, stream));
    reset_amax(stream);
    forward_ops(false, ii[0], ii[1], ii[2]);
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flag = 0;
    if (tail_layer >= 0) HIP_THROW(hipMemcpyAsync(cnt, tw.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    if (conv_mode >= 1) HIP_THROW(hipMemcpyAsync(&flag, range_flag.p, 4, hipMemcpyDeviceToHost, stream));
    HIP_THROW(hipStreamSynchronize(stream));
    if (flag && conv_mode >= 1) {
      const int mode_was = conv_mode;
      // a convolution produced |x| > 65504: fp16(hi) of the split overflowed somewhere downstream.  The reference
      // computes in fp32 (_caffe.cpp:46-48): redo THIS forward on the exact fp32 matrix-core kernels.
      ++sh->range_fallbacks;
      conv_mode = 0;
      try {
        forward_ops(false, ii[0], ii[1], ii[2]);
        if (tail_layer >= 0) HIP_THROW(hipMemcpyAsync(cnt, tw.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
        HIP_THROW(hipStreamSynchronize(stream));
      } catch (...) {
        conv_mode = mode_was;
        throw;
      }
      conv_mode = mode_was;
    }
    if (tail_layer >= 0) {
      const int R = cnt[2];
      blobs[boxes_blob].shape = {std::max(R, 1), 5};
      if (prob_blob >= 0) blobs[prob_blob].shape = {R, 2};
    }
    break;
  }
  for (size_t i = 0; i < blobs.size(); ++i)
    if (!std::count(inputs.begin(), inputs.end(), (int)i) && blobs[i].kind != BK_FUSED) blobs[i].dev_newer = true;
}

float* shf_net::host_data(int bi) {
  Blob& b = blobs[bi];
  if (b.kind == BK_FUSED)
    throw std::runtime_error("blob '" + b.name + "' is fused into the detection tail and not materialised");
  const size_t n = b.count();
  b.host.ensure(std::max<size_t>(n, 1) * 4);
  const bool is_input = std::count(inputs.begin(), inputs.end(), bi) > 0;
  if (b.dev_newer && n > 0) {
    if (b.kind == BK_NHWC) {
      b.stage.ensure(n * 4);
      {
        ProfScope ps(prof, stream, PC_LAYOUT, 0, 8.0 * n);
        CHECK_RC(launch_nhwc_to_nchw(view_of(bi), (float*)b.stage.p, stream));
      }
      HIP_THROW(hipMemcpyAsync(b.host.p, b.stage.p, n * 4, hipMemcpyDeviceToHost, stream));
    } else {
      HIP_THROW(hipMemcpyAsync(b.host.p, b.dev.p, n * 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_THROW(hipStreamSynchronize(stream));
    b.dev_newer = false;
  }
  if (is_input) b.host_newer = true;
  return b.host.p;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
#define API_BEGIN try {
#define API_END(failval)                  \
  }                                       \
  catch (const std::exception& e) {       \
    set_error(e.what());                  \
    return failval;                       \
  }                                       \
  catch (...) {                           \
    set_error("unknown error");           \
    return failval;                       \
  }

static std::mutex g_box_mu;
static MergeCtx* g_box_ctx = nullptr;
static hipStream_t g_box_stream = nullptr;
static DevBuf* g_box_in = nullptr;

extern "C" {

const char* shf_last_error(void) { return g_err.c_str(); }
const char* shf_version(void) { return "smallhardface_amd 0.1 (gfx950)"; }
int shf_set_mode_gpu(void) { return 0; }

int shf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int shf_set_device(int device_id) {
  API_BEGIN
  HIP_THROW(hipSetDevice(device_id));
  return 0;
  API_END(-1)
}

shf_net* shf_net_create(const char* prototxt_path, const char* prototxt_text, const char* caffemodel_path,
                        int phase) {
  API_BEGIN
  std::string text;
  if (prototxt_text && prototxt_text[0]) {
    text = prototxt_text;
  } else {
    if (!prototxt_path) throw std::runtime_error("no prototxt given");
    std::ifstream f(prototxt_path);
    if (!f) throw std::runtime_error(std::string("Could not open file ") + prototxt_path);
    std::stringstream ss;
    ss << f.rdbuf();
    text = ss.str();
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw std::runtime_error("no HIP device available: the detection runtime has no CPU fallback");
  std::unique_ptr<shf_net> net(new shf_net());
  net->phase = phase;
  net->build(text, caffemodel_path);
  return net.release();
  API_END(nullptr)
}

shf_net* shf_net_clone(shf_net* src) {
  API_BEGIN
  std::unique_ptr<shf_net> net(new shf_net(src->sh));
  net->phase = src->phase;
  net->clone_src = src;
  net->build(src->proto_text, nullptr);
  net->clone_src = nullptr;
  return net.release();
  API_END(nullptr)
}

void shf_net_destroy(shf_net* net) { delete net; }

int shf_net_num_blobs(shf_net* net) { return (int)net->blobs.size(); }
const char* shf_net_blob_name(shf_net* net, int i) {
  if (i < 0 || i >= (int)net->blobs.size()) return nullptr;
  return net->blobs[i].name.c_str();
}
int shf_net_num_inputs(shf_net* net) { return (int)net->inputs.size(); }
int shf_net_input_blob(shf_net* net, int i) { return net->inputs[i]; }
int shf_net_num_outputs(shf_net* net) { return (int)net->outputs.size(); }
int shf_net_output_blob(shf_net* net, int i) { return net->outputs[i]; }
int shf_net_num_layers(shf_net* net) { return (int)net->layers.size(); }
const char* shf_net_layer_name(shf_net* net, int i) { return net->layers[i].name.c_str(); }
const char* shf_net_layer_type(shf_net* net, int i) { return net->layers[i].type.c_str(); }
int shf_net_layer_num_params(shf_net* net, int layer) { return (int)net->layers[layer].params.size(); }

int shf_net_param_shape(shf_net* net, int layer, int idx, int* dims) {
  auto& p = *net->layers[layer].params[idx];
  for (size_t i = 0; i < p.shape.size(); ++i) dims[i] = p.shape[i];
  return (int)p.shape.size();
}

float* shf_net_param_data(shf_net* net, int layer, int idx) {
  auto& p = *net->layers[layer].params[idx];
  p.dirty = true;
  return p.host.data();
}

int shf_net_param_commit(shf_net* net, int layer) {
  API_BEGIN
  // a shared tensor is committed for every layer that holds it
  for (size_t li = 0; li < net->layers.size(); ++li) {
    bool share = (int)li == layer;
    for (auto& p : net->layers[li].params)
      for (auto& q : net->layers[layer].params)
        if (p == q) share = true;
    if (share) net->commit_params((int)li);
  }
  return 0;
  API_END(-1)
}

int shf_blob_reshape(shf_net* net, int blob, const int* dims, int ndim) {
  API_BEGIN
  if (blob < 0 || blob >= (int)net->blobs.size()) throw std::runtime_error("bad blob index");
  Blob& b = net->blobs[blob];
  std::vector<int> s(dims, dims + ndim);
  for (int d : s)
    if (d < 0) throw std::runtime_error("negative blob dimension");
  if (s != b.shape) {
    b.shape = s;
    if (std::count(net->inputs.begin(), net->inputs.end(), blob)) {
      b.dev.ensure(std::max<size_t>(b.count(), 1) * 4);
      b.host.ensure(std::max<size_t>(b.count(), 1) * 4);
      b.host_newer = true;
    }
  }
  return 0;
  API_END(-1)
}

int shf_blob_shape(shf_net* net, int blob, int* dims) {
  Blob& b = net->blobs[blob];
  for (size_t i = 0; i < b.shape.size(); ++i) dims[i] = b.shape[i];
  return (int)b.shape.size();
}

float* shf_blob_mutable_host_data(shf_net* net, int blob) {
  API_BEGIN
  if (blob < 0 || blob >= (int)net->blobs.size()) throw std::runtime_error("bad blob index");
  return net->host_data(blob);
  API_END(nullptr)
}

int shf_net_forward(shf_net* net) {
  API_BEGIN
  net->forward();
  return 0;
  API_END(-1)
}

int shf_net_set_proposal_cfg(shf_net* net, int pre_nms_topN, float score_thresh, float min_size) {
  API_BEGIN
  net->pre_nms_topN = pre_nms_topN;
  net->score_thresh = score_thresh;
  net->min_size = min_size;
  net->alloc_buffers();
  return 0;
  API_END(-1)
}

int shf_net_set_conv_mode(shf_net* net, int mode) {
  API_BEGIN
  if (mode < 0 || mode > 4)
    throw std::runtime_error("conv mode must be 0 (fp32), 1 (split-fp16 x3), 2 (x2), 3 (plain fp16) or 4 (bf16)");
  if (net->conv_mode == mode) return 0;
  HIP_THROW(hipDeviceSynchronize());  // the mode is shared with every lane: nothing may be in flight while it flips
  net->conv_mode = mode;
  if (mode >= 1) {
    // the fp32 packs always exist; the split-fp16 packs are made on first use, re-made by every commit in a split
    // mode, and re-made here when a commit in fp32 mode left them stale (also re-runs the |w| <= 65504 check)
    try {
      for (size_t li = 0; li < net->layers.size(); ++li) {
        Layer& L = net->layers[li];
        if (L.type != "Convolution" || L.params.empty()) continue;
        ParamBlob& w = *L.params[0];
        if (L.kclass == 1 && mode == 4 && w.first_frag.p && (!w.first_frag_b.p || w.bf_stale)) net->commit_params((int)li);
        if (L.kclass != 0 || !conv_f16x3_eligible(w.shape[1], w.shape[0], L.k, L.pad, L.dil)) continue;
        if (mode == 4 ? (!w.packed16b.p || w.bf_stale) : (!w.packed16.p || w.split_stale)) net->commit_params((int)li);
      }
    } catch (...) {
      net->conv_mode = 0;  // e.g. a weight outside the fp16 range: stay on the exact kernels
      throw;
    }
  }
  return 0;
  API_END(-1)
}

int shf_net_get_conv_mode(shf_net* net) { return net->conv_mode; }

int shf_net_set_layer_products(shf_net* net, const char* layer, int nprod) {
  API_BEGIN
  if (!layer) throw std::runtime_error("set_layer_products: null layer name");
  if (nprod == 0) {
    net->sh->layer_products.erase(layer);
    return 0;
  }
  if (nprod < 1 || nprod > 3) throw std::runtime_error("set_layer_products: 1, 2 or 3 products (0 clears the override)");
  bool found = false;
  for (auto& L : net->layers) found = found || L.name == layer;
  if (!found) throw std::runtime_error(std::string("set_layer_products: no layer named '") + layer + "'");
  HIP_THROW(hipDeviceSynchronize());
  net->sh->layer_products[layer] = nprod;
  return 0;
  API_END(-1)
}

long long shf_net_range_fallbacks(shf_net* net) { return net->sh->range_fallbacks; }

void shf_alloc_counts(long long* device_allocs, long long* pinned_host_allocs) {
  if (device_allocs) *device_allocs = g_dev_allocs.load();
  if (pinned_host_allocs) *pinned_host_allocs = g_host_allocs.load();
}

int shf_net_record_event(shf_net* net) {
  API_BEGIN
  if (!net->ev_mark) HIP_THROW(hipEventCreateWithFlags(&net->ev_mark, hipEventDisableTiming));
  HIP_THROW(hipEventRecord(net->ev_mark, net->stream));
  return 0;
  API_END(-1)
}

int shf_net_wait_event(shf_net* net, shf_net* other) {
  API_BEGIN
  if (other->ev_mark) HIP_THROW(hipStreamWaitEvent(net->stream, other->ev_mark, 0));
  return 0;
  API_END(-1)
}

int shf_net_set_pipeline(shf_net* net, int enable) {
  API_BEGIN
  if (!enable) {
    net->pipelined = false;
    return 0;
  }
  HIP_THROW(hipDeviceSynchronize());
  if (!net->sh->conv_stream) HIP_THROW(hipStreamCreateWithFlags(&net->sh->conv_stream, hipStreamNonBlocking));
  // this head's own stream carries ~100 tiny kernels per image beside the convolutions of the next image:
  // highest priority, so the dispatcher never parks them behind a grid of thousands of workgroups
  int least = 0, greatest = 0;
  HIP_THROW(hipDeviceGetStreamPriorityRange(&least, &greatest));
  if (greatest != least) {
    hipStream_t hs = nullptr;
    HIP_THROW(hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, greatest));
    (void)hipStreamDestroy(net->stream);
    net->stream = hs;
  }
  net->pipelined = true;
  return 0;
  API_END(-1)
}

int shf_net_set_predecessor(shf_net* net, shf_net* prev) {
  API_BEGIN
  net->pred = prev;
  return 0;
  API_END(-1)
}

// fused path: has a split-fp16 convolution enqueued on `net` (as head of a pass) left the fp16 range?  Synchronises.
static const char* kRangeMsg =
    "split-fp16 range exceeded: a convolution output has |x| > 65504 (fp16 hi overflows); this image must be "
    "re-run with conv mode fp32";
static void throw_if_out_of_range(shf_net* net) {
  if (net->conv_mode < 1) return;
  int flag = 0;
  HIP_THROW(hipMemcpyAsync(&flag, net->range_flag.p, 4, hipMemcpyDeviceToHost, net->stream));
  HIP_THROW(hipStreamSynchronize(net->stream));
  if (flag) throw std::runtime_error(kRangeMsg);
}

int shf_detect_begin(shf_net* net) {
  API_BEGIN
  if (net->tail_layer < 0) throw std::runtime_error("net has no proposal layer");
  if (!net->pipelined) HIP_THROW(hipMemsetAsync(net->range_flag.p, 0, 4, net->stream));
  net->img_count.ensure(64);
  HIP_THROW(hipMemsetAsync(net->img_count.p, 0, 64, net->stream));
  net->img_units = 0;
  net->img_pass = 0;
  return 0;
  API_END(-1)
}

void shf_net::prepare_unit(const float* data, int data_on_device, int H, int W, hipStream_t st) {
  Blob& d = blobs[data_blob];
  std::vector<int> shp = {1, d.shape.size() == 4 ? d.shape[1] : 3, H, W};
  if (shp != d.shape) d.shape = shp;
  if (d.shape != last_data_shape) {
    infer_shapes();
    alloc_buffers();
  }
  if (data_on_device) {
    d.ext_dev = data;
  } else {
    d.ext_dev = nullptr;
    d.dev.ensure(d.count() * 4);
    HIP_THROW(hipMemcpyAsync(d.dev.p, data, d.count() * 4, hipMemcpyHostToDevice, st));
  }
  // (the activation-exponent slots are zero here: zeroed at build, by Net.forward(), and by every pass's tail reset)
}

void shf_net::ensure_img_cap(int units_after) {
  const int rmax = pre_nms_topN > 0 ? pre_nms_topN : (int)tw.cap_anchors;
  const int need = units_after * rmax;
  if (need <= img_cap) return;
  // grow geometrically; keep what is already gathered
  const int ncap = std::max(need, std::max(img_cap * 2, 16 * rmax));
  DevBuf nd, nk;
  nd.ensure((size_t)ncap * 5 * 4);
  size_t npad = 1;
  while (npad < (size_t)ncap) npad <<= 1;
  nk.ensure(npad * 8);
  if (img_dets.p) {
    HIP_THROW(hipMemcpyAsync(nd.p, img_dets.p, (size_t)img_cap * 5 * 4, hipMemcpyDeviceToDevice, stream));
    HIP_THROW(hipMemcpyAsync(nk.p, img_keys.p, (size_t)img_cap * 8, hipMemcpyDeviceToDevice, stream));
    HIP_THROW(hipStreamSynchronize(stream));
  }
  std::swap(img_dets.p, nd.p); std::swap(img_dets.cap, nd.cap);
  std::swap(img_keys.p, nk.p); std::swap(img_keys.cap, nk.cap);
  img_cap = ncap;
}

constexpr int kMaxGroup = 16;  // units per grouped pass (conv_common.h MAX_GROUP, tail.hip TG)

// append a group of finished units (their proposals sit in the members' output blobs) to `net`'s image list -- or,
// per_member, to each member's own (reset) list -- in ONE launch
static void append_units(shf_net* net, shf_net* const* srcs, int n, const int* im_w, const float* im_scale,
                         const int* flip, float thresh, bool per_member, hipStream_t st = nullptr, Prof* pf = nullptr) {
  if (!st) st = net->stream;
  if (!per_member) net->ensure_img_cap(net->img_units + n);
  AppendUnit us[kMaxGroup];
  for (int m = 0; m < n; ++m) {
    shf_net* src = srcs[m];
    shf_net* dst = per_member ? src : net;
    if (per_member) {
      dst->img_count.ensure(64);
      dst->ensure_img_cap(1);
      dst->img_units = 1;
      dst->img_pass = 1;  // the kernel writes count[0] = 0, count[1] = rows
    }
    AppendUnit& u = us[m];
    u.boxes5 = (const float*)src->blobs[src->boxes_blob].dev.p;
    u.probs2 = src->probs_out();
    u.counters = src->tw.counters;
    u.r_max = src->pre_nms_topN > 0 ? src->pre_nms_topN : (int)src->tw.cap_anchors;
    u.im_w = (float)im_w[m]; u.im_scale = im_scale[m]; u.flip = flip[m];
    u.dets5 = (float*)dst->img_dets.p;
    u.keys = (unsigned long long*)dst->img_keys.p;
    u.count = (int*)dst->img_count.p;
    u.cap = dst->img_cap;
  }
  ProfScope ps(pf ? *pf : net->prof, st, PC_TAIL, 0, 0);
  CHECK_RC(launch_append_dets_group(us, n, srcs[0]->pre_nms_topN, thresh, net->img_pass, per_member ? 1 : 0, st));
  if (!per_member) {
    net->img_units += n;
    net->img_pass++;
  }
}

int shf_pyramid_level_shape(int im_h, int im_w, double scale, int max_resolution, int* lvl_h, int* lvl_w, int* H,
                            int* W) {
  if (im_h < 1 || im_w < 1 || !(scale > 0) || max_resolution < 1) return -1;
  // np.round / cvRound: round half to even (the default FP environment of nearbyint)
  const int lh = scale == 1.0 ? im_h : (int)std::nearbyint((double)im_h * scale);
  const int lw = scale == 1.0 ? im_w : (int)std::nearbyint((double)im_w * scale);
  if (lh < 1 || lw < 1) return -1;
  if (lvl_h) *lvl_h = lh;
  if (lvl_w) *lvl_w = lw;
  if (H) *H = (lh + max_resolution - 1) / max_resolution * max_resolution;
  if (W) *W = (lw + max_resolution - 1) / max_resolution * max_resolution;
  return 0;
}

int shf_make_pyramid_level(shf_net* net, const uint8_t* im_bgr_dev, int im_h, int im_w, double scale, int flip,
                           const double* pixel_means, float* out_dev, int H, int W, int lvl_h, int lvl_w) {
  API_BEGIN
  if (!im_bgr_dev || !out_dev || !pixel_means) throw std::runtime_error("make_pyramid_level: null pointer");
  if (lvl_h > H || lvl_w > W || lvl_h < 1 || lvl_w < 1) throw std::runtime_error("make_pyramid_level: bad geometry");
  ProfScope ps(net->prof, net->cstream(), PC_LAYOUT, 0, 15.0 * H * W);
  CHECK_RC(launch_pyramid_level(im_bgr_dev, im_h, im_w, scale, flip, pixel_means, out_dev, H, W, lvl_h, lvl_w,
                                net->cstream()));
  return 0;
  API_END(-1)
}

int shf_detect_add_level(shf_net* net, const float* data, int data_on_device, int H, int W, int im_h, int im_w,
                         float im_scale, int flip, float thresh) {
  API_BEGIN
  net->prepare_unit(data, data_on_device, H, W, net->stream);
  net->forward_ops(true, (float)im_h, (float)im_w, im_scale);
  net->blobs[net->data_blob].ext_dev = nullptr;
  append_units(net, &net, 1, &im_w, &im_scale, &flip, thresh, false);
  return 0;
  API_END(-1)
}

int shf_detect_add_levels(shf_net* net, int n, shf_net** members, const float* const* data, int data_on_device,
                          const int* H, const int* W, const int* im_h, const int* im_w, const float* im_scale,
                          const int* flip, float thresh, int per_member_lists) {
  API_BEGIN
  if (n < 1 || n > kMaxGroup) throw std::runtime_error("detect_add_levels: 1..16 units per group");
  for (int m = 0; m < n; ++m) {
    for (int q = 0; q < m; ++q)
      if (members[q] == members[m]) throw std::runtime_error("detect_add_levels: members must be distinct nets");
    if (members[m]->layers.size() != net->layers.size())
      throw std::runtime_error("detect_add_levels: members must be lanes of the same net");
  }
  // The member lanes' activations are free as soon as the previous pass over them has run its logits
  // kernels (the rest of a tail works on its own buffers), so with a predecessor head set this pass's
  // convolutions overlap the predecessor's sorts / gathers / appends; the full hand-over is only
  // awaited before this pass's own tail (below).
  // (With a predecessor head the start only awaits its last convolution; the logits events are awaited
  // right before the first layer that writes a feature map the tails read: every blob owns its buffer.)
  int first_feat_writer = (int)net->layers.size();
  for (size_t li = 0; li < net->layers.size(); ++li)
    for (int t : net->layers[li].tops)
      for (int f : net->tail_feat_blobs)
        if (t == f && (int)li < first_feat_writer) first_feat_writer = (int)li;
  // Pipelined heads (shf_net_set_pipeline): the convolutions and logits kernels of consecutive images share ONE
  // in-order stream, so no cross-stream hand-over is needed for the activation buffers; only the rest of the tails,
  // the appends and the merge run on this head's own stream, beside the next image's convolutions.
  const bool shared = net->pipelined && net->sh->conv_stream;
  hipStream_t cs = shared ? net->sh->conv_stream : net->stream;
  const bool early_start = !shared && net->pred && net->pred->ev_convs && first_feat_writer < (int)net->layers.size();
  if (early_start) HIP_THROW(hipStreamWaitEvent(net->stream, net->pred->ev_convs, 0));
  for (int m = 0; m < n; ++m) {
    if (!shared && !early_start && members[m]->logits_done)
      HIP_THROW(hipStreamWaitEvent(net->stream, members[m]->logits_done, 0));
    members[m]->prepare_unit(data[m], data_on_device, H[m], W[m], cs);
  }
  // (detect_begin zeroes it on the head's stream; a pipelined head: on the conv stream, by the image's FIRST pass -- a
  // longer unit list comes as several passes into the same list and a later one must not clear an earlier one's flag)
  if (shared && !per_member_lists && net->img_units == 0) HIP_THROW(hipMemsetAsync(net->range_flag.p, 0, 4, cs));
  if (per_member_lists) HIP_THROW(hipMemsetAsync(net->range_flag.p, 0, 4, cs));  // no detect_begin on this path
  struct FlagScope {  // one range flag per pass: the head's
    shf_net** mb; int n;
    FlagScope(shf_net** m, int n_, int* f) : mb(m), n(n_) { for (int i = 0; i < n; ++i) mb[i]->flag_ptr = f; }
    ~FlagScope() { for (int i = 0; i < n; ++i) mb[i]->flag_ptr = nullptr; }
  } flag_scope(members, n, (int*)net->range_flag.p);
  std::vector<ConvArgs> group(n);
  auto launch_group_conv = [&](size_t li, hipStream_t st) {
    Layer& L = net->layers[li];
    double fl = 0, by = 4.0 * L.params[0]->count();
    for (int m = 0; m < n; ++m) {
      shf_net* mb = members[m];
      mb->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], st, &net->prof, (int)li, &group[m]);
      fl += conv_flops(mb->layers[li], mb->blobs[mb->layers[li].bottoms[0]].shape,
                       mb->blobs[mb->layers[li].tops[0]].shape);
      by += 4.0 * (mb->blobs[mb->layers[li].bottoms[0]].count() + mb->blobs[mb->layers[li].tops[0]].count());
    }
    if (group[0].wsplit16) {
      if (group[0].img && L.first_src >= 0) {  // conv1_1's work rides in this launch
        for (int m = 0; m < n; ++m) {
          shf_net* mb = members[m];
          const Layer& F = mb->layers[L.first_src];
          fl += conv_flops(F, mb->blobs[F.bottoms[0]].shape, mb->blobs[F.tops[0]].shape);
        }
      }
      if (conv_f16x3_group_is_dual(group.data(), n)) {
        SubProf sp{&net->prof, st, fl, by, {}};
        group[0].sub_hook = &SubProf::hook;
        group[0].sub_ctx = &sp;
        CHECK_RC_LAYER(launch_conv_f16x3_group(group.data(), n, st), L.name);
      } else {
        ProfScope ps(net->prof, st, f16x3_prof_class(group[0], L.nout, group.data(), n), fl, by);
        CHECK_RC_LAYER(launch_conv_f16x3_group(group.data(), n, st), L.name);
      }
    } else {
      const int pc = conv_prof_class(L.k, L.dil, L.nout);
      ProfScope ps(net->prof, st, pc, fl, by);
      CHECK_RC(launch_conv_mfma_group(group.data(), n, st));
    }
  };
  // the three shared-weight dilated heads of every unit as ONE launch (conv_f16x3_h3.h); false: not that shape / mode
  int heads3_done = -1;
  std::vector<ConvArgs> g2(n), g4(n);
  auto launch_group_heads3 = [&](size_t li, hipStream_t st) {
    Layer& L = net->layers[li];
    double fl = 0, by = 4.0 * L.params[0]->count();
    for (int m = 0; m < n; ++m) {
      shf_net* mb = members[m];
      mb->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], st, &net->prof, (int)li, &group[m]);
      mb->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], st, &net->prof, L.heads3_d2, &g2[m]);
      mb->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], st, &net->prof, L.heads3_d4, &g4[m]);
      fl += 3.0 * conv_flops(mb->layers[li], mb->blobs[mb->layers[li].bottoms[0]].shape, mb->blobs[mb->layers[li].tops[0]].shape);
      by += 4.0 * (mb->blobs[mb->layers[li].bottoms[0]].count() + 3.0 * mb->blobs[mb->layers[li].tops[0]].count());
    }
    if (!group[0].wsplit16h || !conv_f16x3_group_is_heads3(group.data(), g2.data(), g4.data(), n)) return false;
    ProfScope ps(net->prof, st, PC_CONV_F16X3_H3, fl, by);
    CHECK_RC(launch_conv_f16x3_heads3(group.data(), g2.data(), g4.data(), n, st));
    return true;
  };
  for (size_t li = 0; li < net->layers.size(); ++li) {
    Layer& L = net->layers[li];
    if (early_start && (int)li == first_feat_writer)
      for (int m = 0; m < n; ++m)
        if (members[m]->logits_done) HIP_THROW(hipStreamWaitEvent(net->stream, members[m]->logits_done, 0));
    if (L.op == OP_SKIP) continue;
    if (L.op == OP_CONV && L.kclass == 0 && L.heads3_lead >= 0 && heads3_done == L.heads3_lead) {
      continue;   // written by the dilation-1 sibling's launch
    } else if (L.op == OP_CONV && L.kclass == 0 && L.heads3_d2 >= 0 && launch_group_heads3(li, cs)) {
      heads3_done = (int)li;
    } else if (L.op == OP_CONV && L.kclass == 0) {
      launch_group_conv(li, cs);
    } else if (L.op == OP_DECONV && n > 1) {
      // the units' depthwise up-samplings as one launch (ten serial 5..60-us launches otherwise)
      View dins[kMaxGroup], douts[kMaxGroup];
      unsigned* dslots[kMaxGroup];
      double fl = 0, by = 0;
      bool ok = true;
      for (int m = 0; m < n; ++m) {
        shf_net* mb = members[m];
        dins[m] = mb->view_of(L.bottoms[0]);
        douts[m] = mb->view_of(L.tops[0]);
        dslots[m] = mb->amax_slot(L.tops[0]);
        ok = ok && dins[m].B == 1;
        fl += 2.0 * mb->blobs[L.tops[0]].count() * 4;
        by += 4.0 * (mb->blobs[L.bottoms[0]].count() + mb->blobs[L.tops[0]].count());
      }
      if (ok) {
        ProfScope ps(net->prof, cs, PC_DECONV, fl, by);
        CHECK_RC(launch_deconv_depthwise_group(dins, douts, n, (const float*)L.params[0]->raw.p,
                                               L.params.size() > 1 ? (const float*)L.params[1]->raw.p : nullptr, L.k,
                                               L.stride, L.pad, cs,
                                               net->conv_mode >= 1 && net->conv_mode != 4 ? (int*)net->range_flag.p : nullptr, dslots));
      } else {
        for (int m = 0; m < n; ++m)
          members[m]->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], cs, &net->prof, (int)li, nullptr);
      }
    } else if (L.op == OP_TAIL) {
      // The detection tails of all units as ONE launch per stage (counters reset, logits, decode, sort stages,
      // gather): ~15 launches per image instead of ~100.  Phase 1 (reset + logits) is what reads the head feature
      // maps; phase 2 works on the members' tail workspaces only.
      TailArgs targs[kMaxGroup];
      TailWork* tws[kMaxGroup];
      float* tb[kMaxGroup];
      float* tp[kMaxGroup];
      double tfl = 0, tby = 0;
      for (int m = 0; m < n; ++m) {
        shf_net* mb = members[m];
        if (mb->tail_w_dirty || mb->tail_gen != *mb->wgen) mb->build_tail_weights();
        targs[m] = mb->tail_args((float)im_h[m], (float)im_w[m], im_scale[m], true);
        tws[m] = &mb->tw;
        tb[m] = (float*)mb->blobs[mb->boxes_blob].dev.p;
        tp[m] = mb->probs_out();
        const double K = (double)targs[m].h * targs[m].w;
        tfl += 2.0 * K * mb->tail_A * 6 * mb->tail_Cf;
        tby += 4.0 * K * (mb->tail_heads * mb->tail_Cf + mb->tail_A * 18);
      }
      for (int m = 1; m < n; ++m) targs[m].wcls[0] = targs[0].wcls[0], targs[m].bcls[0] = targs[0].bcls[0];  // lanes hold identical copies
      if (!net->ev_convs) HIP_THROW(hipEventCreateWithFlags(&net->ev_convs, hipEventDisableTiming));
      if (shared) {
        // the members' tail workspaces were last used by the predecessor head's tails (its own stream)
        if (net->pred && net->pred->ev_mark) HIP_THROW(hipStreamWaitEvent(cs, net->pred->ev_mark, 0));
        {
          ProfScope ps(net->prof, cs, PC_TAIL, tfl, tby);
          CHECK_RC(launch_tail_group(targs, tws, tb, tp, n, cs, nullptr, 1));
        }
        // the feature maps are consumed: the conv stream is free for the next image
        HIP_THROW(hipEventRecord(net->ev_convs, cs));
        HIP_THROW(hipStreamWaitEvent(net->stream, net->ev_convs, 0));
      } else {
        // (every n: a one-unit pass over two heads needs the same hand-over as a ten-unit one)
        if (net->pred && net->pred->ev_mark) HIP_THROW(hipStreamWaitEvent(net->stream, net->pred->ev_mark, 0));
        {
          ProfScope ps(net->prof, net->stream, PC_TAIL, tfl, tby);
          CHECK_RC(launch_tail_group(targs, tws, tb, tp, n, net->stream, nullptr, 1));
        }
        // recorded AFTER phase 1: its reset kernel zeroes the member lanes' activation-exponent slots, which the
        // successor head's first convolutions (early_start waits for this event only) publish into and read
        HIP_THROW(hipEventRecord(net->ev_convs, net->stream));
      }
      for (int m = 0; m < n; ++m) {  // hand-over mark for passes issued from another head without a pipeline
        shf_net* mb = members[m];
        if (shared) {
          mb->logits_done = net->ev_convs;   // recorded on the conv stream right after the logits launch above
          continue;
        }
        if (!mb->ev_logits) HIP_THROW(hipEventCreateWithFlags(&mb->ev_logits, hipEventDisableTiming));
        HIP_THROW(hipEventRecord(mb->ev_logits, net->stream));
        mb->logits_done = mb->ev_logits;
      }
      {
        ProfScope ps(net->prof, net->stream, PC_TAIL, 0, 0);
        CHECK_RC(launch_tail_group(targs, tws, tb, tp, n, net->stream, nullptr, 2));
      }
    } else {
      for (int m = 0; m < n; ++m)
        members[m]->forward_ops(true, (float)im_h[m], (float)im_w[m], im_scale[m], cs, &net->prof, (int)li, nullptr);
    }
  }
  for (int m = 0; m < n; ++m) members[m]->blobs[members[m]->data_blob].ext_dev = nullptr;
  // units of different images (per_member_lists): each member keeps its own list
  append_units(net, members, n, im_w, im_scale, flip, thresh, per_member_lists != 0, net->stream, &net->prof);
  return 0;
  API_END(-1)
}

// rows gathered so far + (same synchronisation) the split-fp16 range flag of the passes enqueued on `net`
static int detect_count_checked(shf_net* net, bool check_range) {
  int c[2] = {0, 0}, flag = 0;
  HIP_THROW(hipMemcpyAsync(c, net->img_count.p, 8, hipMemcpyDeviceToHost, net->stream));
  if (check_range && net->conv_mode >= 1)
    HIP_THROW(hipMemcpyAsync(&flag, net->range_flag.p, 4, hipMemcpyDeviceToHost, net->stream));
  HIP_THROW(hipStreamSynchronize(net->stream));
  if (flag) throw std::runtime_error(kRangeMsg);
  return c[net->img_pass & 1];
}

int shf_detect_count(shf_net* net) {
  API_BEGIN
  return detect_count_checked(net, false);
  API_END(-1)
}

int shf_detect_export(shf_net* net, float* dst_dev5, int cap_rows, int* n_rows) {
  API_BEGIN
  const int n = detect_count_checked(net, true);
  *n_rows = n;
  const int w = std::min(n, cap_rows);
  if (w > 0) {
    HIP_THROW(hipMemcpyAsync(dst_dev5, net->img_dets.p, (size_t)w * 5 * 4, hipMemcpyDeviceToDevice, net->stream));
    HIP_THROW(hipStreamSynchronize(net->stream));
  }
  return 0;
  API_END(-1)
}

int shf_detect_export_many(shf_net* net, int n, shf_net** members, float* const* dst_dev5, int cap_rows,
                           int* n_rows) {
  API_BEGIN
  // after a per_member_lists pass: everything was enqueued on `net`'s stream -> one sync, then all
  // counts, then the row copies, then one more sync
  throw_if_out_of_range(net);  // (synchronises net's stream)
  for (int m = 0; m < n; ++m) {
    int c[2] = {0, 0};
    HIP_THROW(hipMemcpyAsync(c, members[m]->img_count.p, 8, hipMemcpyDeviceToHost, net->stream));
    HIP_THROW(hipStreamSynchronize(net->stream));
    n_rows[m] = c[members[m]->img_pass & 1];
  }
  for (int m = 0; m < n; ++m) {
    const int w = std::min(n_rows[m], cap_rows);
    if (w > 0)
      HIP_THROW(hipMemcpyAsync(dst_dev5[m], members[m]->img_dets.p, (size_t)w * 5 * 4, hipMemcpyDeviceToDevice,
                               net->stream));
  }
  HIP_THROW(hipStreamSynchronize(net->stream));
  return 0;
  API_END(-1)
}

int shf_detect_import(shf_net* net, const float* src_dev5, int n_rows) {
  API_BEGIN
  if (n_rows <= 0) return 0;
  const int have = shf_detect_count(net);
  if (have < 0) return -1;
  const int need = have + n_rows;
  if (need > net->img_cap) {
    const int ncap = std::max(need, net->img_cap * 2);
    DevBuf nd, nk;
    nd.ensure((size_t)ncap * 5 * 4);
    size_t npad = 1;
    while (npad < (size_t)ncap) npad <<= 1;
    nk.ensure(npad * 8);
    if (net->img_dets.p && have > 0)
    {   // (on the list's own stream, and finished before the old buffer is released below)
      HIP_THROW(hipMemcpyAsync(nd.p, net->img_dets.p, (size_t)have * 5 * 4, hipMemcpyDeviceToDevice, net->stream));
      HIP_THROW(hipStreamSynchronize(net->stream));
    }
    std::swap(net->img_dets.p, nd.p); std::swap(net->img_dets.cap, nd.cap);
    std::swap(net->img_keys.p, nk.p); std::swap(net->img_keys.cap, nk.cap);
    net->img_cap = ncap;
  }
  HIP_THROW(hipMemcpyAsync((float*)net->img_dets.p + (size_t)have * 5, src_dev5, (size_t)n_rows * 5 * 4,
                           hipMemcpyDeviceToDevice, net->stream));
  int c[2] = {need, need};
  HIP_THROW(hipMemcpyAsync(net->img_count.p, c, 8, hipMemcpyHostToDevice, net->stream));
  HIP_THROW(hipStreamSynchronize(net->stream));
  return 0;
  API_END(-1)
}

int shf_detect_finish(shf_net* net, int method, float nms_thresh, double* out5, int cap, int* n_out) {
  API_BEGIN
  *n_out = 0;
  const int n = detect_count_checked(net, true);
  if (n == 0) {
    if (method == 0) {  // bbox_vote on an empty set (test.py:184-186)
      const double d[5] = {10, 10, 20, 20, 0.0001};
      if (cap > 0) memcpy(out5, d, sizeof(d));
      *n_out = 1;
    }
    return 0;
  }
  ProfScope ps(net->prof, net->stream, PC_MERGE, 0, 0);
  return net->merge.run((const float*)net->img_dets.p, n, method, nms_thresh, out5, cap, n_out, nullptr, net->stream);
  API_END(-1)
}

static void box_ctx_init() {
  if (g_box_ctx) return;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw std::runtime_error("no HIP device available: box merging has no CPU fallback");
  g_box_ctx = new MergeCtx();
  g_box_in = new DevBuf();
  HIP_THROW(hipStreamCreateWithFlags(&g_box_stream, hipStreamNonBlocking));
}

int shf_nms(const float* dets5, int n, float thresh, int device_id, int32_t* keep, int* n_keep) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(g_box_mu);
  *n_keep = 0;
  if (n <= 0) return 0;
  if (device_id >= 0) {
    int cur = -1;
    HIP_THROW(hipGetDevice(&cur));
    if (cur != device_id) HIP_THROW(hipSetDevice(device_id));  // _set_device, nms_kernel.cu:91-100
  }
  box_ctx_init();
  g_box_in->ensure((size_t)n * 5 * 4);
  HIP_THROW(hipMemcpyAsync(g_box_in->p, dets5, (size_t)n * 5 * 4, hipMemcpyHostToDevice, g_box_stream));
  return g_box_ctx->run((const float*)g_box_in->p, n, 1, thresh, nullptr, 0, n_keep, keep, g_box_stream);
  API_END(-1)
}

int shf_bbox_vote(const float* dets5, int n, float thresh, double* out5, int cap, int* n_out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(g_box_mu);
  *n_out = 0;
  if (n <= 0) {
    const double d[5] = {10, 10, 20, 20, 0.0001};
    if (cap > 0) memcpy(out5, d, sizeof(d));
    *n_out = 1;
    return 0;
  }
  box_ctx_init();
  g_box_in->ensure((size_t)n * 5 * 4);
  HIP_THROW(hipMemcpyAsync(g_box_in->p, dets5, (size_t)n * 5 * 4, hipMemcpyHostToDevice, g_box_stream));
  return g_box_ctx->run((const float*)g_box_in->p, n, 0, thresh, out5, cap, n_out, nullptr, g_box_stream);
  API_END(-1)
}

int shf_caffemodel_read_blob(const char* path, const char* layer, int idx, float* out, int cap, int* dims,
                             int* ndim) {
  API_BEGIN
  auto src = read_caffemodel(path);
  for (auto& L : src) {
    if (L.name != layer) continue;
    if (idx < 0 || idx >= (int)L.blobs.size()) throw std::runtime_error("caffemodel: blob index out of range");
    const WireBlob& b = L.blobs[idx];
    *ndim = (int)std::min<size_t>(b.shape.size(), 8);
    for (int i = 0; i < *ndim; ++i) dims[i] = (int)b.shape[i];
    if (out) std::copy(b.data.begin(), b.data.begin() + std::min<size_t>(b.data.size(), (size_t)cap), out);
    return (int)b.data.size();
  }
  throw std::runtime_error(std::string("caffemodel: no layer named '") + layer + "'");
  API_END(-1)
}

// diagnostics: run the merge pipeline and hand back its intermediates (tests only)
int shf_debug_merge(const float* dets5, int n, float thresh, int ge_pred, unsigned long long* mask_out,
                    int* cluster_out, int* heads_out, int* n_heads, float* sorted_out, int* perm_out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(g_box_mu);
  box_ctx_init();
  g_box_in->ensure((size_t)n * 5 * 4);
  HIP_THROW(hipMemcpyAsync(g_box_in->p, dets5, (size_t)n * 5 * 4, hipMemcpyHostToDevice, g_box_stream));
  int nk = 0;
  std::vector<int32_t> keep(n);
  std::vector<double> tmp((size_t)n * 5);
  CHECK_RC(g_box_ctx->run((const float*)g_box_in->p, n, ge_pred ? 0 : 1, thresh, tmp.data(), n, &nk, keep.data(),
                          g_box_stream));
  const size_t nw = ((size_t)n + 63) / 64;
  HIP_THROW(hipMemcpy(mask_out, g_box_ctx->mask.p, (size_t)n * nw * 8, hipMemcpyDeviceToHost));
  HIP_THROW(hipMemcpy(cluster_out, g_box_ctx->cluster.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_THROW(hipMemcpy(sorted_out, g_box_ctx->sorted.p, (size_t)n * 5 * 4, hipMemcpyDeviceToHost));
  HIP_THROW(hipMemcpy(perm_out, g_box_ctx->perm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  int cnt[2];
  HIP_THROW(hipMemcpy(cnt, g_box_ctx->counters.p, 8, hipMemcpyDeviceToHost));
  *n_heads = cnt[0];
  HIP_THROW(hipMemcpy(heads_out, g_box_ctx->heads.p, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
  return 0;
  API_END(-1)
}

// diagnostics: the proposal stage alone on injected blobs (tests only)
int shf_debug_proposal(shf_net* net, const float* scores, const float* deltas, int h, int w, const float* im_info3,
                       float* out_boxes5, float* out_probs2, int cap, int* n_out, int* overflow) {
  API_BEGIN
  if (net->tail_layer < 0) throw std::runtime_error("net has no proposal layer");
  if (h < 1 || w < 1) throw std::runtime_error("debug_proposal: bad map size");
  // (any (h, w): the layer itself does not care that the real graph only produces even head maps)
  const int A = net->tail_A;
  const size_t K = (size_t)h * w;
  net->ensure_tail_workspace(K * A);
  DevBuf ds, dd;
  ds.ensure(K * 2 * A * 4);
  dd.ensure(K * 4 * A * 4);
  hipStream_t st = net->stream;
  HIP_THROW(hipMemcpyAsync(ds.p, scores, K * 2 * A * 4, hipMemcpyHostToDevice, st));
  HIP_THROW(hipMemcpyAsync(dd.p, deltas, K * 4 * A * 4, hipMemcpyHostToDevice, st));
  TailArgs t;
  t.A = A; t.heads = net->tail_heads; t.Cf = net->tail_Cf;
  t.h = h; t.w = w;
  for (int i = 0; i < A * 4; ++i) t.anchors[i] = (float)net->anchors[i];
  for (int i = 0; i < A; ++i) t.sub_stride[i] = net->sub_stride[i];
  t.feat_stride = net->feat_stride;
  t.im_h = im_info3[0]; t.im_w = im_info3[1]; t.im_scale = im_info3[2];
  t.min_size = net->min_size; t.score_thresh = net->score_thresh; t.pre_nms_topN = net->pre_nms_topN;
  t.probs_given = 1;
  float* boxes = (float*)net->blobs[net->boxes_blob].dev.p;
  float* probs = net->prob_blob >= 0 ? (float*)net->blobs[net->prob_blob].dev.p : (float*)net->tw_rec.p;
  CHECK_RC(launch_tail_inject(t, net->tw, (const float*)ds.p, (const float*)dd.p, st));
  CHECK_RC(launch_tail(t, net->tw, boxes, probs, st, nullptr, 2));
  int cnt[8];
  HIP_THROW(hipMemcpyAsync(cnt, net->tw.counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
  HIP_THROW(hipStreamSynchronize(st));
  const int R = cnt[2];
  if (overflow) *overflow = cnt[1];
  // ProposalLayer's tops: (max(R,1), 5) with the dummy roi when R == 0, and (R, 2)   proposal_layer.py:207-220
  const int rows_b = std::max(R, 1);
  if (n_out) *n_out = R;
  HIP_THROW(hipMemcpy(out_boxes5, boxes, (size_t)std::min(rows_b, cap) * 5 * 4, hipMemcpyDeviceToHost));
  if (R > 0) HIP_THROW(hipMemcpy(out_probs2, probs, (size_t)std::min(R, cap) * 2 * 4, hipMemcpyDeviceToHost));
  net->blobs[net->boxes_blob].shape = {rows_b, 5};
  if (net->prob_blob >= 0) net->blobs[net->prob_blob].shape = {R, 2};
  return 0;
  API_END(-1)
}

// diagnostics: forward_net's flip fix + unscale and detect()'s >thresh cut (append_dets_kernel) on injected
// proposals, appended to the current image's list (shf_detect_begin first; export with shf_detect_export)
int shf_debug_append(shf_net* net, const float* boxes5, const float* probs2, int R, int im_w, float im_scale,
                     int flip, float thresh) {
  API_BEGIN
  if (net->tail_layer < 0) throw std::runtime_error("net has no proposal layer");
  if (R < 0) throw std::runtime_error("debug_append: R < 0");
  net->tw_counters.ensure(64);
  net->tw.counters = (int*)net->tw_counters.p;
  Blob& bb = net->blobs[net->boxes_blob];
  bb.dev.ensure((size_t)std::max(R, 1) * 5 * 4);
  float* probs;
  if (net->prob_blob >= 0) {
    net->blobs[net->prob_blob].dev.ensure((size_t)std::max(R, 1) * 2 * 4);
    probs = (float*)net->blobs[net->prob_blob].dev.p;
  } else {
    net->tw_rec.ensure((size_t)std::max(R, 1) * 2 * 4);
    probs = (float*)net->tw_rec.p;
  }
  hipStream_t st = net->stream;
  if (R > 0) {
    HIP_THROW(hipMemcpyAsync(bb.dev.p, boxes5, (size_t)R * 5 * 4, hipMemcpyHostToDevice, st));
    HIP_THROW(hipMemcpyAsync(probs, probs2, (size_t)R * 2 * 4, hipMemcpyHostToDevice, st));
  }
  const int cnt[8] = {R, 0, R, 0, 0, 0, 0, 0};  // C candidates (all kept: topN is raised below), published R
  HIP_THROW(hipMemcpyAsync(net->tw.counters, cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
  HIP_THROW(hipStreamSynchronize(st));
  const int saved = net->pre_nms_topN;
  net->pre_nms_topN = std::max(R, 1);  // append_unit sizes its launch and the list growth from it
  try {
    append_units(net, &net, 1, &im_w, &im_scale, &flip, thresh, false);
  } catch (...) {
    net->pre_nms_topN = saved;
    throw;
  }
  net->pre_nms_topN = saved;
  return 0;
  API_END(-1)
}

int shf_generate_anchors(int base_size, const double* ratios, int n_ratios, const double* scales, int n_scales,
                         const double* shifts, int n_shifts, const double* strides, double* out, int cap_rows) {
  API_BEGIN
  std::vector<double> a;
  gen_anchors(base_size, std::vector<double>(ratios, ratios + n_ratios), std::vector<double>(scales, scales + n_scales),
              std::vector<double>(shifts, shifts + n_shifts), std::vector<double>(strides, strides + n_scales), a);
  const int rows = (int)a.size() / 4;
  if (rows > cap_rows) throw std::runtime_error("anchor output buffer too small");
  std::copy(a.begin(), a.end(), out);
  return rows;
  API_END(-1)
}

int shf_prof_enable(shf_net* net, int enable) {
  net->prof.on = enable != 0;
  return 0;
}
int shf_prof_only(shf_net* net, int cls) {
  net->prof.only = (cls >= 0 && cls < PC_COUNT) ? cls : -1;
  return 0;
}
int shf_prof_num_classes(shf_net*) { return PC_COUNT; }
const char* shf_prof_class_name(shf_net*, int cls) {
  return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : nullptr;
}
int shf_prof_read(shf_net* net, int cls, int64_t* launches, double* total_ms, double* flops, double* bytes) {
  API_BEGIN
  if (cls < 0 || cls >= PC_COUNT) throw std::runtime_error("bad profile class");
  net->prof.drain();
  *launches = net->prof.launches[cls];
  *total_ms = net->prof.ms[cls];
  *flops = net->prof.flops[cls];
  *bytes = net->prof.bytes[cls];
  return 0;
  API_END(-1)
}
int shf_prof_reset(shf_net* net) {
  API_BEGIN
  net->prof.drain();
  for (int i = 0; i < PC_COUNT; ++i) {
    net->prof.launches[i] = 0;
    net->prof.ms[i] = net->prof.flops[i] = net->prof.bytes[i] = 0;
  }
  return 0;
  API_END(-1)
}
int shf_calib_matrix_pipe(int bf16, int zero_eighths, int constant_operands, int iters, int reps, double* tflops) {
  API_BEGIN
  if (!tflops || iters < 1 || reps < 1 || zero_eighths < 0 || zero_eighths > 8)
    throw std::runtime_error("calib_matrix_pipe: bad arguments");
  const int rc = calib_matrix_pipe(bf16, zero_eighths, constant_operands, iters, reps, tflops);
  if (rc != 0) throw std::runtime_error(std::string("calib_matrix_pipe: ") + hipGetErrorString((hipError_t)rc));
  return 0;
  API_END(-1)
}
int shf_net_sync(shf_net* net) {
  API_BEGIN
  HIP_THROW(hipStreamSynchronize(net->stream));
  return 0;
  API_END(-1)
}

}  // extern "C"
