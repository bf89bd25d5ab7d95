// Internal declarations shared by the HIP kernels and the Net runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace shf {

void set_error(const std::string& msg);
#define SHF_HIP_OK(expr)                                                              \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      ::shf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));            \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)

// ---- tensor view: NHWC fp32 on the device, possibly a channel slice of a wider buffer
struct View {
  float* p = nullptr;  // base of the buffer
  int B = 1, H = 1, W = 1, C = 1;
  int cstride = 1;  // floats per pixel in the underlying buffer (>= C)
  int coff = 0;     // first channel of this view inside a pixel
};

// ---- convolution (stride 1, "same" geometry: pad == dil*(k-1)/2) ------------
struct ConvArgs {
  View in, out;
  const float* wpacked = nullptr;  // [Cin/32][k*k][Cout][32] (see pack_conv_weights)
  const float* wraw = nullptr;     // Caffe layout (Cout,Cin,k,k) for the direct kernels
  const float* wfirst = nullptr;   // first layer: weights transposed to [Cin*k*k][Cout]
  const void* wsplit16 = nullptr;  // split-fp16 pack [Cin/32][tap][Cout][hi32|lo32] (conv_f16x3.hip)
  const void* wsplit16r = nullptr; // fused first pair (conv_f16x3_pc.h): [Cin/32][tap][Cout][8 rotated 16-byte pieces], 128-byte rows
  const void* wsplit16h = nullptr; // dual-tile 4-wave kernel: [Cin/16][tap][Cout][hi16|lo16], lo UNSCALED, weights x 1/wscale_inv
  float wscale_inv = 1.f;          // ... and the power of two the epilogue multiplies back
  const float* bias = nullptr;     // [Cout] or null
  int k = 3, dil = 1, pad = 1;
  int relu = 0;
  const float* img = nullptr;  // fused first layer (f16x3 only): raw NCHW image, transposed weights, bias
  const float* w1t = nullptr;
  const void* w1f = nullptr;   // first-layer weights as split-fp16 MFMA B fragments (pack_first_conv_frags)
  const float* b1 = nullptr;
  View pool;           // optional fused MAX 2x2/2 pool output (p == nullptr: none)
  int write_main = 1;  // 0: the un-pooled output has no other reader and is not written
  // split-fp16 activation format (fused path, conv_f16x3.hip): a blob whose only readers are 4-wave split-fp16
  // convs is stored as [pixel][32-channel chunk][hi 32 halfs | lo 32 halfs] -- the same 4 B per element, already
  // split, so the consumer's halo staging is a plain copy
  int in_split = 0, out_split = 0, pool_split = 0;
  int bf16 = 0;   // conv mode "bf16": one bf16 product per fp32 product (nprod = 1), the packs hold bf16 bit patterns, fp32
                  // activations in HBM, no fp16 range guard and no activation exponent (bf16 has fp32's range)
  int nprod = 3;  // split-fp16 kernels: fp16 products formed per fp32 product -- 3 (hi*hi + hi*lo + lo*hi: fp32-class),
                  // 2 (drops a_lo*b_hi: activations effectively fp16) or 1 (hi*hi only: plain fp16 operands)
  int* range_flag = nullptr;  // split-fp16 kernels raise it when an output leaves the fp16 range (net_forward.cpp: fp32 re-run)
  // activation-exponent slots (conv_common.h ConvMember): max |value| of the input blob as left by its producers, and
  // where this launch raises the max of what it writes (main output / fused pool output); null = not tracked
  const unsigned* in_amax = nullptr;
  unsigned* out_amax = nullptr;
  unsigned* pool_amax = nullptr;
  // dual-tile family: a layer may take two launches (two tiles per block, then single tiles); the profiler wants one
  // record per KERNEL: hook(ctx, 0, variant, share) before and hook(ctx, 1, ..) after each, variant = in_split * 4 +
  // (rows == 8) * 2 + (tiles per block == 1), share = its fraction of the launch's pixel tiles
  void (*sub_hook)(void* ctx, int after, int variant, double share) = nullptr;
  void* sub_ctx = nullptr;
};
// which kernel class a conv will use: 0 = mfma implicit GEMM, 1 = first-layer direct (NCHW in), 2 = generic direct
int conv_kernel_class(int Cin, int Cout, int k, int pad, int dil, bool in_nchw);
int launch_conv_mfma(const ConvArgs& a, hipStream_t s);
// one grid over up to 16 problems that share the layer (weights/channels) but not H x W
int launch_conv_mfma_group(const ConvArgs* as, int n, hipStream_t s);
// first layer: input is the NCHW 'data' blob (B,Cin,H,W), Cin <= 8, Cout % 16 == 0
int launch_conv_first(const float* in_nchw, const ConvArgs& a, hipStream_t s);
int launch_conv_direct(const ConvArgs& a, hipStream_t s);
int conv_init_attributes();
// split-fp16 (3 x fp16 MFMA, fp32-class accuracy) variant for 3x3 / dilation 1 layers
bool conv_f16x3_eligible(int Cin, int Cout, int k, int pad, int dil);
bool conv_f16x3_uses_pc();         // fused first pair: producer/consumer kernel (SHF_F16X3_PC=0 disables)
bool conv_f16x3_pc_persistent();   // ... as one block per CU walking the tiles (SHF_F16X3_PC_PERSIST)
bool conv_f16x3_uses_w4(int Cin);  // Cout % 128 == 0, 3x3 / dilation 1 layers: the 4-wave dual-tile family (Cin >= 128) or the 8-wave kernel
int conv_f16x3_w4_mt(const ConvArgs* as, int n);  // 4-wave family: 16-row (4) or 8-row (2) tiles for this launch
int conv_f16x3_init_attributes();
int launch_conv_f16x3_group(const ConvArgs* as, int n, hipStream_t s);
size_t split16_conv_weight_halfs(int Cout, int Cin, int k);
size_t split16h_conv_weight_halfs(int Cout, int Cin, int k);
size_t split16r_conv_weight_halfs(int Cout, int Cin, int k);
void pack_conv_weights_split16r(const float* w, int Cout, int Cin, int k, void* dst, bool bf = false);
float pack_conv_weights_split16h(const float* w, int Cout, int Cin, int k, void* dst, bool bf = false);  // returns 1 / scale
bool conv_f16x3_group_is_dual(const ConvArgs* as, int n);
bool conv_f16x3_group_is_dilated_w4(const ConvArgs* as, int n);
bool conv_f16x3_k1_gemm_shape(int Cin, int Cout);   // a 1x1 layer of this shape runs on the GEMM kernel: the family's pack, split-format input
bool conv_f16x3_group_is_k1_gemm(const ConvArgs* as, int n);
// the three shared-weight dilated heads (dilation 1 / 2 / 4, same input, same weights) as ONE launch (conv_f16x3_h3.h; SHF_F16X3_HEADS3)
bool conv_f16x3_group_is_heads3(const ConvArgs* a1, const ConvArgs* a2, const ConvArgs* a4, int n);
int launch_conv_f16x3_heads3(const ConvArgs* a1, const ConvArgs* a2, const ConvArgs* a4, int n, hipStream_t s);
void pack_conv_weights_split16(const float* w, int Cout, int Cin, int k, void* dst, bool bf = false);
// first layer (64, 27) as the B operand of v_mfma_f32_32x32x16_f16: [n 2][kk 2][hi/lo 2][lane 64][8 halfs], K padded 27 -> 32
constexpr size_t kFirstConvFragHalfs = 2 * 2 * 2 * 64 * 8;
void pack_first_conv_frags(const float* w, void* dst, bool bf = false);
// host-side weight re-pack for the mfma kernel
void pack_conv_weights(const float* w, int Cout, int Cin, int k, float* dst);
size_t packed_conv_weight_floats(int Cout, int Cin, int k);

// ---- misc layers -------------------------------------------------------------
int launch_maxpool(const View& in, const View& out, int k, int stride, int pad, hipStream_t s);
int launch_amax_raise(unsigned* dst, const unsigned* src, hipStream_t s);
// depthwise transposed conv (group == C), weights (C,1,k,k) Caffe layout
int launch_deconv_depthwise(const View& in, const View& out, const float* w, const float* bias, int k,
                            int stride, int pad, hipStream_t s, int* range_flag = nullptr, unsigned* out_amax = nullptr);
int launch_deconv_depthwise_group(const View* ins, const View* outs, int n, const float* w, const float* bias, int k,
                                  int stride, int pad, hipStream_t s, int* range_flag = nullptr,
                                  unsigned* const* out_amax = nullptr);
int launch_copy_view(const View& in, const View& out, hipStream_t s);       // concat fallback
int launch_nhwc_to_nchw(const View& in, float* out_nchw, hipStream_t s);    // blob.data read-back
int launch_nchw_to_nhwc(const float* in_nchw, const View& out, hipStream_t s);
// pyramid unit from the raw BGR uint8 image (pre.hip)
int launch_pyramid_level(const uint8_t* im, int im_h, int im_w, double scale, int flip, const double* means,
                         float* out, int H, int W, int lvl_h, int lvl_w, hipStream_t s);

// ---- detection tail ------------------------------------------------------------
struct TailArgs {
  // per-anchor-set (dilation head) features and 1x1 weights
  int A = 3;                 // anchors per cell (== number of heads, or 1 head with A outputs)
  int heads = 3;             // number of distinct feature maps (3: one per dilation; 1: plain template)
  View feat[8];              // head feature maps (h,w,Cf)
  const float* wcls[8];      // heads==A: (2,Cf) per head ; heads==1: (2A,Cf)
  const float* bcls[8];
  const float* wbox[8];      // heads==A: (4,Cf) per head ; heads==1: (4A,Cf)
  const float* bbox[8];
  int h = 0, w = 0, Cf = 128;
  float anchors[8 * 4];      // base anchors (A,4) as float32 (bbox_transform.py:37 casts)
  int feat_stride = 8;
  int sub_stride[8];         // per-anchor subsampling stride ratio (proposal_layer.py:160-169)
  float im_h = 0, im_w = 0, im_scale = 1;  // im_info
  float min_size = 0;        // cfg.TEST.ANCHOR_MIN_SIZE (scaled by im_scale in-kernel)
  float score_thresh = 0.002f;
  int pre_nms_topN = 10000;
  // optional materialised Caffe blobs (NCHW), may be null
  float* cls_prob_reshape_nchw = nullptr;  // (1,2A,h,w)
  float* bbox_pred_nchw = nullptr;         // (1,4A,h,w)
  int probs_given = 0;       // diagnostics: the logits workspace already holds bg/fg probabilities (launch_tail_inject)
};
struct TailWork {  // device workspace owned by the net, sized for the largest level seen
  float* logits = nullptr;           // [K][A][6]
  float* rec = nullptr;              // [K*A][6] = bg, fg, x1,y1,x2,y2
  unsigned long long* keys = nullptr;  // [pow2 >= K*A]
  int* counters = nullptr;           // [8]: 0 = n candidates, 1 = overflow flag, 2 = R, 3 = argmax idx
  size_t cap_anchors = 0, cap_keys = 0;
  // the lane's activation-exponent slots (conv_common.h): the tail's reset kernel runs behind the last convolution of
  // a pass and zeroes them for the next one (ten 6-us fill kernels per image otherwise)
  unsigned* amax = nullptr;
  int n_amax = 0;
};
// runs logits -> decode -> select -> sort; leaves R (device counters[2]) rows in out_boxes/out_probs
int launch_tail(const TailArgs& a, TailWork& ws, float* out_boxes5, float* out_probs2, hipStream_t s,
                hipEvent_t after_logits = nullptr, int phase = 0);
// diagnostics (shf_debug_proposal): fill the logits workspace from injected (1,2A,h,w) probabilities and
// (1,4A,h,w) deltas (device pointers, NCHW) and raise the overflow flag; follow with launch_tail(phase 2, probs_given)
int launch_tail_inject(const TailArgs& a, TailWork& ws, const float* scores_nchw, const float* deltas_nchw,
                       hipStream_t s);
// after_logits: recorded once the feature maps are consumed; phase 1 = only up to there, 2 = only the rest

// the same for a GROUP of independent units (the units of an image's pyramid): ONE launch per stage.  phase 0:
// everything; 1: counters reset + logits kernel (after_logits recorded behind it); 2: decode -> sort -> gather
int launch_tail_group(const TailArgs* as, TailWork* const* wss, float* const* out_boxes5, float* const* out_probs2,
                      int n, hipStream_t s, hipEvent_t after_logits = nullptr, int phase = 0);

// generic device sort of u64 keys, descending; n_dev points at the element count on the device,
// n_max is a host-known upper bound (sizes the launch sequence)
int launch_sort_desc_u64(unsigned long long* keys, const int* n_dev, size_t n_max, hipStream_t s);
int launch_sort_desc_u64_group(unsigned long long* const* keys, const int* const* n_dev, const size_t* n_max, int n,
                               hipStream_t s);

// append the >thresh detections of a group of finished units to an image list (test.py:52-66,163-167): flip fix,
// unscale, threshold cut.  count[pass & 1] = list length before the pass, count[(pass + 1) & 1] after it; with
// per_member != 0 every unit has its own (reset) list and count[1] receives its length.
struct AppendUnit {
  const float* boxes5;
  const float* probs2;
  const int* counters;  // the unit's tail counters (TailWork::counters)
  int r_max;            // host-side bound of the unit's rows (sizes the launch)
  float im_w, im_scale;
  int flip;
  float* dets5;
  unsigned long long* keys;
  int* count;
  int cap;
};
int launch_append_dets_group(const AppendUnit* us, int n, int topN, float thresh, int pass, int per_member,
                             hipStream_t s);

// ---- box merging ----------------------------------------------------------------
struct MergeWork {
  float* sorted = nullptr;                // [n][5] score-descending
  unsigned long long* mask = nullptr;     // [n][ceil(n/64)]
  int* cluster = nullptr;                 // [n] cluster head of each box (-1: none)
  int* heads = nullptr;                   // [n] kept head indices (sorted order)
  int* counters = nullptr;                // [4]: 0 = number of heads
  double* out = nullptr;                  // [n][5]
  int* out_idx = nullptr;                 // [n]
  size_t cap_n = 0, cap_mask_words = 0;
};
// dets sorted descending on the device -> greedy clustering.
// ge_pred: 1 -> IoU >= thr (bbox_vote), 0 -> IoU > thr (nms)
int launch_iou_mask(const float* sorted5, int n, float thr, int ge_pred, unsigned long long* mask, hipStream_t s);
int launch_greedy_scan(const unsigned long long* mask, int n, int* cluster, int* heads, int* counters, hipStream_t s);
int launch_vote_accumulate(const float* sorted5, const unsigned long long* mask, const int* cluster, int n,
                           const int* heads, const int* counters, double* out5, int* n_out, hipStream_t s);
int launch_gather_sorted(const float* dets5, const unsigned long long* keys, int n, float* sorted5, int* perm,
                         hipStream_t s);
int launch_make_keys(const float* dets5, int n, unsigned long long* keys, hipStream_t s);

}  // namespace shf
