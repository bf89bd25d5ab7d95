// Internal declarations of the Net runtime (net_graph.cpp: build / shapes / parameters; net_forward.cpp: the per-layer
// executor and the profiler classes; net_detect.cpp: the fused per-image detection pipeline; net_api.cpp: the rest of the
// C ABI of include/shf_hip.h).  One struct, four translation units (split in round 5 out of a 2 400-line net.cpp).
#pragma once
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

extern thread_local std::string g_err;     // net_api.cpp (shf_last_error)

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
inline void fill_now(void* p, int byte, size_t n) {
  HIP_THROW(hipMemsetAsync(p, byte, n, nullptr));
  HIP_THROW(hipStreamSynchronize(nullptr));
}

// (re)allocations made by the grow-only buffers of this process: a new level shape re-plans sizes and pointers and only
// allocates where a buffer has to grow (shf_alloc_counts; bench.py's mixed-shape leg reports them after its first pass)
extern std::atomic<long long> g_dev_allocs, g_host_allocs;   // net_api.cpp

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
  // BK_FUSED blobs (tops of the layers folded into the detection tail) are materialised ON DEMAND when Blob.data is read
  // after Net.forward() (shf_net::materialize_fused): what they hold, and for a predictor's top which head it belongs to
  int fused_role = 0;              // FusedRole
  int fused_head = -1;
  size_t count() const {
    size_t c = 1;
    for (int d : shape) c *= (size_t)d;
    return c;
  }
};

// what a tail-fused blob holds (pycaffe exposes every blob after forward(), pycaffe.py:24-32 / _caffe.cpp:222-242)
enum FusedRole {
  FR_NONE = 0,
  FR_CLS_CONV,     // top of a class-score 1x1 conv: (1, 2, h, w) per head, or (1, 2A, h, w) channel = cls * A + a (plain template)
  FR_BOX_CONV,     // top of a bbox 1x1 conv: (1, 4, h, w) per head, or (1, 4A, h, w) channel = 4a + j
  FR_CLS_PLANES,   // the pre-softmax (1, 2, A*h, w) tensor: axis-2 Concat of the heads' scores, or the plain template's Reshape
  FR_PROB_PLANES   // the Softmax's top, (1, 2, A*h, w): the memory of the materialised (1, 2A, h, w) cls_prob blob
};

void check_blob_dims(const std::vector<int>& shp, const std::string& name);   // net_graph.cpp: Blob::Reshape's limits

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
enum ProfClass { PC_CONV_MFMA, PC_CONV_MFMA_1 = 1, PC_CONV_MFMA_7 = 7, PC_CONV_F16X3_128, PC_CONV_F16X3_W4, PC_CONV_F16X3_W4_SPLIT, PC_CONV_F16X3_W4_MT2, PC_CONV_F16X3_W4_SPLIT_MT2, PC_CONV_F16X3_64, PC_CONV_F16X3_64_FUSE1, PC_CONV_F16X3_64_D2, PC_CONV_F16X3_64_D4, PC_CONV_F16X3_128_K1, PC_CONV_F16X3_64_K1, PC_CONV_F16X3_PC, PC_CONV_F16X3_W4D_0, PC_CONV_F16X3_W4D_7 = PC_CONV_F16X3_W4D_0 + 7, PC_CONV_F16X3_PCP, PC_CONV_F16X3_K1G, PC_CONV_F16X3_W4D_D2, PC_CONV_F16X3_W4D_D4, PC_CONV_F16X3_H3, PC_CONV_FIRST, PC_CONV_DIRECT, PC_POOL, PC_DECONV, PC_TAIL, PC_MERGE, PC_LAYOUT, PC_H2D, PC_D2H, PC_COUNT };
extern const char* const kProfNames[PC_COUNT];   // net_forward.cpp

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
int f16x3_prof_class(const ConvArgs& a, int nout, const ConvArgs* group = nullptr, int n = 1);   // net_forward.cpp
int conv_prof_class(int k, int dil, int nout);

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

// net_graph.cpp
void gen_anchors(int base_size, const std::vector<double>& ratios, const std::vector<double>& scales,
                 const std::vector<double>& shifts, const std::vector<double>& strides, std::vector<double>& out);
std::map<std::string, std::vector<double>> parse_param_str(const std::string& s);
int conv_out(int n, int k, int pad, int stride, int dil);

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
  // Net.forward() in a split-fp16 mode on a detector graph runs the FUSED path's kernels (fused first pair, pools in the
  // convolutions' epilogues, split activation format: forward_ops(true) -- the very pass shf_detect_add_level runs, so
  // lib/test.py's ten net.forward() calls give the detections of the device-resident path bit for bit) and leaves the
  // intermediate blobs unmaterialised: `plain_stale`.  Reading one of them (Blob.data) then runs the per-layer kernels
  // once, everything but the proposal tail (ensure_plain), so every name in net.blobs stays readable (pycaffe.py:24-32).
  bool in_net_forward = false, plain_stale = false, plain_skip_tail = false, inputs_reshaped = false;
  float last_im_info[3] = {0.f, 0.f, 1.f};
  bool forward_fast_eligible() const;
  void ensure_plain();
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
  void materialize_fused(int bi);   // Blob.data of a tail-fused blob, re-ordered on the host from the tail workspace
  void load_caffemodel(const std::string& path);
};

constexpr int kMaxGroup = 16;  // units per grouped pass (conv_common.h MAX_GROUP, tail.hip TG)

double conv_flops(const Layer& L, const std::vector<int>& in, const std::vector<int>& out);   // net_forward.cpp
// net_detect.cpp: append a group of finished units to `net`'s image list (or, per_member, to each member's own) in ONE launch
void append_units(shf_net* net, shf_net* const* srcs, int n, const int* im_w, const float* im_scale, const int* flip,
                  float thresh, bool per_member, hipStream_t st = nullptr, Prof* pf = nullptr);

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
