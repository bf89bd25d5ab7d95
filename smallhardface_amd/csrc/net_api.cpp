// Net runtime, part 4: the C ABI of include/shf_hip.h (everything but the detection pipeline, net_detect.cpp).
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
#include "net_internal.h"

namespace shf {
thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
std::atomic<long long> g_dev_allocs{0}, g_host_allocs{0};
}  // namespace shf

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
  if (ndim < 0 || ndim > 32) throw std::runtime_error("Blob.reshape: 0 .. 32 axes");
  std::vector<int> s(dims, dims + ndim);
  check_blob_dims(s, b.name);
  if (s != b.shape) {
    b.shape = s;
    if (std::count(net->inputs.begin(), net->inputs.end(), blob)) {
      b.dev.ensure(std::max<size_t>(b.count(), 1) * 4);
      b.host.ensure(std::max<size_t>(b.count(), 1) * 4);
      b.host_newer = true;
      net->inputs_reshaped = true;   // (the device copy of the last forward's input may have been re-allocated: ensure_plain)
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

static void box_ctx_init();

// _get_image_blob of a whole scale list for callers that work with HOST blobs (lib/test.py's detect(): the reference calls
// cv2.resize here -- a native library as well): the uint8 image goes up once, every level is formed by pre.hip's kernel
// (the arithmetic of the host mirror, bit for bit) on the box context's stream and comes back as an UNPADDED (1,3,h,w) blob.
int shf_image_blobs(const uint8_t* im_bgr_host, int im_h, int im_w, int n, const double* scales, const double* pixel_means,
                    float* const* out_host, const int* lvl_h, const int* lvl_w) {
  API_BEGIN
  if (!im_bgr_host || !scales || !pixel_means || !out_host || !lvl_h || !lvl_w || n < 1 || im_h < 1 || im_w < 1)
    throw std::runtime_error("image_blobs: bad argument");
  std::lock_guard<std::mutex> lk(g_box_mu);
  box_ctx_init();
  static DevBuf* im_dev = new DevBuf();
  static DevBuf* lv_dev = new DevBuf();
  const size_t im_bytes = (size_t)im_h * im_w * 3;
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (lvl_h[i] < 1 || lvl_w[i] < 1 || !(scales[i] > 0)) throw std::runtime_error("image_blobs: bad level geometry");
    total += (size_t)3 * lvl_h[i] * lvl_w[i];
  }
  im_dev->ensure(im_bytes);
  lv_dev->ensure(total * 4);
  HIP_THROW(hipMemcpyAsync(im_dev->p, im_bgr_host, im_bytes, hipMemcpyHostToDevice, g_box_stream));
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    float* o = (float*)lv_dev->p + off;
    CHECK_RC(launch_pyramid_level((const uint8_t*)im_dev->p, im_h, im_w, scales[i], 0, pixel_means, o, lvl_h[i], lvl_w[i],
                                  lvl_h[i], lvl_w[i], g_box_stream));
    HIP_THROW(hipMemcpyAsync(out_host[i], o, (size_t)3 * lvl_h[i] * lvl_w[i] * 4, hipMemcpyDeviceToHost, g_box_stream));
    off += (size_t)3 * lvl_h[i] * lvl_w[i];
  }
  HIP_THROW(hipStreamSynchronize(g_box_stream));
  return 0;
  API_END(-1)
}

int shf_device_pci_bus_id(char* out, int cap) {
  API_BEGIN
  if (!out || cap < 16) throw std::runtime_error("device_pci_bus_id: buffer too small");
  int dev = 0;
  HIP_THROW(hipGetDevice(&dev));
  HIP_THROW(hipDeviceGetPCIBusId(out, cap, dev));
  return 0;
  API_END(-1)
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
