/* libshf_hip.so -- C ABI of the MI355X-native detection runtime.
 *
 * This is the drop-in boundary for the reference's inference hot path.  Each entry
 * point names the reference interface it replaces (paths relative to the reference
 * repo bairdzhang/smallhardface).  All functions return 0 on success (or a valid
 * pointer) and non-zero / NULL on failure with a message in shf_last_error();
 * nothing aborts the process (Caffe's glog CHECKs do).
 *
 * Layout contract: host-visible blob data is fp32 NCHW exactly like pycaffe
 * (caffe/python/caffe/_caffe.cpp:222-242); the device keeps activations NHWC.
 */
#ifndef SHF_HIP_H_
#define SHF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct shf_net shf_net;

/* ---- process / device ------------------------------------------------------ */
/* caffe.set_mode_gpu()            caffe/python/caffe/_caffe.cpp:52,394 (set_mode_gpu) */
int shf_set_mode_gpu(void);
/* caffe.set_device(id)            caffe/python/caffe/_caffe.cpp:396 (Caffe::SetDevice) */
int shf_set_device(int device_id);
int shf_device_count(void);
const char* shf_last_error(void);
const char* shf_version(void);

/* ---- Net ------------------------------------------------------------------- */
/* caffe.Net(proto, weights, phase) -- legacy 3-arg ctor, _caffe.cpp:137-151
 * (Net::Net net.cpp:28 + CopyTrainedLayersFrom net.cpp:733).  prototxt_path may
 * be NULL when prototxt_text is given.  caffemodel may be NULL/"" (parameters are
 * then zero until set through shf_net_param_*).  phase: 1 = TEST. */
shf_net* shf_net_create(const char* prototxt_path, const char* prototxt_text,
                        const char* caffemodel_path, int phase);
/* Execution lane: a second net over the SAME parameter tensors (Net::ShareTrainedLayersWith,
 * net.cpp:665-685) with its own activations, workspace and HIP stream, so independent
 * pyramid units overlap on the GPU.  Destroy lanes before the net they were cloned from. */
shf_net* shf_net_clone(shf_net* src);
void shf_net_destroy(shf_net* net);

/* Net._blob_names / Net._inputs / Net._outputs          _caffe.cpp:432-440 */
int shf_net_num_blobs(shf_net* net);
const char* shf_net_blob_name(shf_net* net, int i);
int shf_net_num_inputs(shf_net* net);
int shf_net_input_blob(shf_net* net, int i);   /* index into blobs */
int shf_net_num_outputs(shf_net* net);
int shf_net_output_blob(shf_net* net, int i);
/* Net._layer_names / Layer.type / Layer.blobs           _caffe.cpp:434,483-487 */
int shf_net_num_layers(shf_net* net);
const char* shf_net_layer_name(shf_net* net, int i);
const char* shf_net_layer_type(shf_net* net, int i);
int shf_net_layer_num_params(shf_net* net, int layer);
/* shape of param blob `idx` of layer `layer`; returns ndim (<=4) */
int shf_net_param_shape(shf_net* net, int layer, int idx, int* dims);
/* Host pointer to the parameter (Caffe layout: conv (Cout,Cin/g,kh,kw), bias (Cout)).
 * Writing through it and then calling shf_net_param_commit re-packs it on the device
 * (equivalent of writing net.params[name][i].data[...]).  Shared params
 * (param { name: } in the prototxt, net.cpp:421-513) alias one buffer. */
float* shf_net_param_data(shf_net* net, int layer, int idx);
int shf_net_param_commit(shf_net* net, int layer);

/* Blob.reshape(*dims)              _caffe.cpp:244-256 -> Blob::Reshape blob.cpp:23 */
int shf_blob_reshape(shf_net* net, int blob, const int* dims, int ndim);
/* Blob.shape                       _caffe.cpp:455-459; returns ndim */
int shf_blob_shape(shf_net* net, int blob, int* dims);
/* Blob.data -> mutable_cpu_data()  _caffe.cpp:222-242, syncedmem.cpp:39-64:
 * syncs device->host (NHWC->NCHW) if the device copy is newer, marks host as head.
 * The pointer stays valid until the blob grows. */
float* shf_blob_mutable_host_data(shf_net* net, int blob);
/* Net._forward(0, n-1)             _caffe.cpp:414 -> Net::ForwardFromTo net.cpp:516 */
int shf_net_forward(shf_net* net);

/* The in-graph Python layer reads cfg.TEST.{N_DETS_PER_MODULE,SCORE_THRESH,
 * ANCHOR_MIN_SIZE} from the global config (lib/layers/proposal_layer.py:88-92);
 * the native proposal stage takes them here. */
int shf_net_set_proposal_cfg(shf_net* net, int pre_nms_topN, float score_thresh, float min_size);

/* Arithmetic of the MFMA convolutions (3x3 at any dilation and 1x1): 0 = exact fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain); 1 = split-fp16 (2, 3: see the ladder below): x = hi + lo*2^-11 with
 * three fp16 MFMAs per product, fp32 accumulate (2^-22 relative per product: fp32-class, 5.3x
 * the fp32 MFMA rate).  Default 0, or the SHF_CONV_MODE environment variable at creation.  The mode is shared by
 * a net and all lanes cloned from it (like the proposal configuration above). */
int shf_net_set_conv_mode(shf_net* net, int mode);
int shf_net_get_conv_mode(shf_net* net);
/* The reduced-precision ladder (BASELINE configs C3 / C5 name bf16): modes 2 and 3 of shf_net_set_conv_mode form two
 * (a_hi*b_hi + a_hi*b_lo: activations rounded to fp16, weights still split) or one (plain fp16 operands) of the three
 * fp16 products, at 2/3 and 1/3 of the matrix-core work; their score drift against the fp32 reference is measured,
 * not assumed (tools/precision_ladder.py -> profiles/).  shf_net_set_layer_products overrides the count for ONE
 * layer by name (1..3; 0 clears), so a mode can be relaxed only where the measured contribution to the score error is
 * negligible.  Shared by a net and its lanes.  Every split-fp16 kernel (4-wave family, 8-wave, fused first pair) has
 * the 2- and 1-product instantiations; the fused pair's conv1_1 (0.5 % of the FLOPs) always forms three.
 * Mode 4 = bf16: ONE v_mfma_f32_32x32x16_bf16 product per fp32 product, operands rounded to bf16 (8 mantissa bits), fp32
 * accumulate, fp32 activations in HBM.  bf16 has fp32's exponent range: no range guard, no weight refusal.  Like modes 2
 * and 3 it is a drift-labelled throughput mode, not a parity mode (bench.py reports its drift beside the headline). */
int shf_net_set_layer_products(shf_net* net, const char* layer, int nprod);
/* fp16 range guard of mode 1 (the reference is fp32 everywhere, caffe/python/caffe/_caffe.cpp:46-48): hi = fp16(x)
 * overflows above 65504.  Weights are checked when they are packed (shf_net_param_commit / shf_net_set_conv_mode
 * fail with a message).  Every split-fp16 convolution raises a device flag when one of its outputs leaves the range:
 * shf_net_forward then re-runs that forward on the exact fp32 kernels (the count of such re-runs is returned here,
 * shared by a net and its lanes); on the fused path shf_detect_finish / shf_detect_export(_many) fail with
 * "split-fp16 range exceeded ..." so that the caller can re-run the image in mode 0 -- never a silent inf/NaN. */
long long shf_net_range_fallbacks(shf_net* net);
/* (Re)allocations this process has made so far in the runtime's grow-only device / pinned-host buffers (Blob::Reshape
 * semantics, caffe/src/caffe/blob.cpp:22-50: capacity only ever grows; syncedmem.cpp:15-50 is where Caffe allocates).  A
 * stream of images whose shapes were all seen before must leave both counts unchanged.  Measurement only. */
void shf_alloc_counts(long long* device_allocs, long long* pinned_host_allocs);
/* _get_image_blob (lib/utils/test_utils.py:29-46) + im_list_to_blob (lib/utils/blob.py:16-32) for a whole scale list, for
 * callers that keep HOST blobs (lib/test.py:109-178 detect() -> forward_net): `im_bgr_host` uint8 HxWx3 is uploaded once,
 * level i = mean subtraction + cv2.resize(fx = fy = scales[i], INTER_LINEAR) restated (csrc/pre.hip: the host mirror's
 * arithmetic bit for bit; scale 1.0 = no resize) lands in out_host[i] as an UNPADDED, unflipped (1,3,lvl_h[i],lvl_w[i])
 * fp32 blob; lvl_h / lvl_w from shf_pyramid_level_shape.  The reference calls a native library (OpenCV) at this point too. */
int shf_image_blobs(const uint8_t* im_bgr_host, int im_h, int im_w, int n, const double* scales, const double* pixel_means,
                    float* const* out_host, const int* lvl_h, const int* lvl_w);
/* PCI bus id ("0000:c5:00.0") of the device the runtime is set to (shf_set_device; caffe.set_device,
 * caffe/python/caffe/_caffe.cpp:394-396), so that a host-side sampler can find the card's sysfs hwmon files (clock, socket
 * power) without a GPU call of its own.  Measurement only (bench.py `telemetry`). */
int shf_device_pci_bus_id(char* out, int cap);

/* ---- fused per-image path (device-resident pyramid; lib/test.py:109-178) ---- */
/* detect(): begin an image */
int shf_detect_begin(shf_net* net);
/* forward_net() for one (scale, flip) unit, lib/test.py:21-66: `data` is the level
 * blob (1,3,H,W) fp32 NCHW already padded to MAX_RESOLUTION, on the device when
 * data_on_device != 0; im_h/im_w are the UNPADDED dims that go to im_info and
 * the flip fix; detections with fg prob > thresh are un-flipped, unscaled and
 * appended to the image's device-side list (test.py:52-54,59-66,163-167). */
int shf_detect_add_level(shf_net* net, const float* data, int data_on_device,
                         int H, int W, int im_h, int im_w, float im_scale, int flip, float thresh);
/* The same for a GROUP of units at once (the whole pyramid of an image): every MFMA conv
 * layer runs as ONE grid over all units (they share the layer's weights, not the spatial
 * size), so the small levels do not serialise latency-bound launches.  members[i] supplies
 * the activation buffers of unit i: `net` itself and/or lanes made with shf_net_clone, all
 * distinct, at most 16 per call (one kernel-argument member table; more units = several calls, the lists
 * concatenate); the work is enqueued on `net`'s stream and the detections land in `net`'s image
 * list in unit order -- or, with per_member_lists != 0 (units of DIFFERENT images, the multi-GPU
 * window), in each member's own list (reset first); synchronise `net` before exporting them. */
int shf_detect_add_levels(shf_net* net, int n, shf_net** members, const float* const* data,
                          int data_on_device, const int* H, const int* W, const int* im_h,
                          const int* im_w, const float* im_scale, const int* flip, float thresh,
                          int per_member_lists);
/* ---- image pre-processing ----------------------------------- */
/* Geometry of one pyramid unit: the resized level is (lvl_h, lvl_w) = round-half-even(im * scale)
 * (cv2.resize dsize rule, lib/utils/test_utils.py:40-44) and the net input (H, W) is that rounded
 * up to a multiple of max_resolution (cfg.MAX_RESOLUTION, lib/test.py:35-38).  Needs no GPU. */
int shf_pyramid_level_shape(int im_h, int im_w, double scale, int max_resolution, int* lvl_h, int* lvl_w,
                            int* H, int* W);
/* _get_image_blob for one (scale, flip) unit on the device (lib/utils/test_utils.py:29-46,
 * lib/utils/blob.py:16-32, flip lib/test.py:150, pad lib/test.py:35-38): im_bgr_dev is the raw
 * im_h x im_w x 3 uint8 image (cv2.imread layout) in device memory, pixel_means 3 doubles
 * (cfg.PIXEL_MEANS), out_dev receives the (1,3,H,W) fp32 blob forward_net would build, H/W/lvl_*
 * from shf_pyramid_level_shape.  Enqueued on `net`'s stream; pass out_dev to
 * shf_detect_add_level(s) with data_on_device = 1. */
int shf_make_pyramid_level(shf_net* net, const uint8_t* im_bgr_dev, int im_h, int im_w, double scale, int flip,
                           const double* pixel_means, float* out_dev, int H, int W, int lvl_h, int lvl_w);
/* Cross-lane ordering for software-pipelining images over two head lanes: record marks the
 * current end of `net`'s stream; wait makes `net`'s stream wait for `other`'s last mark (e.g. the
 * next image's convolutions, which reuse the member lanes' buffers, wait for the previous
 * image's appends while its box merging still runs). */
int shf_net_record_event(shf_net* net);
int shf_net_wait_event(shf_net* net, shf_net* other);
/* Finer hand-over for the same pipeline: with `prev` set as `net`'s predecessor head,
 * shf_detect_add_levels on `net` starts its convolutions as soon as the member lanes' previous tails
 * have consumed the head feature maps (their logits kernels), and waits for `prev`'s last
 * shf_net_record_event mark only before its own tails, which reuse the members' tail buffers.
 * prev = NULL clears it. */
int shf_net_set_predecessor(shf_net* net, shf_net* prev);
/* Image pipeline over head lanes, robust form: with enable != 0 the grouped passes of `net` put their convolutions
 * and logits kernels on ONE in-order stream shared by the net and all its lanes (consecutive images queue behind
 * each other, no cross-stream hand-over of the activation buffers), and only the rest of the tails, the appends and
 * the merge run on `net`'s own stream, which is re-created with the highest stream priority so that its tiny kernels
 * are dispatched beside the next image's convolutions.  Combine with shf_net_set_predecessor (the members' tail
 * workspaces are handed from head to head). */
int shf_net_set_pipeline(shf_net* net, int enable);
/* Box merging for the image (test.py:168-175): method 0 = BBOX_VOTE (test.py:181),
 * 1 = NMS (lib/nms).  out5 rows are (x1,y1,x2,y2,score) as double (bbox_vote
 * returns float64).  *n_out = number of rows (may exceed cap; only cap written). */
int shf_detect_finish(shf_net* net, int method, float nms_thresh, double* out5, int cap, int* n_out);
/* number of >thresh detections gathered so far for the current image */
int shf_detect_count(shf_net* net);
/* Pyramid sharding across GPUs (the reference gathers per-worker results through a
 * multiprocessing.Queue, lib/test.py:327-344; here units of ONE image may live on several
 * GPUs): export copies the current image's (x1,y1,x2,y2,score) fp32 rows to a caller-owned
 * DEVICE buffer (for an RCCL gather); import appends rows gathered from other ranks (device
 * pointer) to the current image before shf_detect_finish. */
int shf_detect_export(shf_net* net, float* dst_dev5, int cap_rows, int* n_rows);
int shf_detect_import(shf_net* net, const float* src_dev5, int n_rows);
/* export of all the per-member lists of one shf_detect_add_levels(..., per_member_lists=1) pass */
int shf_detect_export_many(shf_net* net, int n, shf_net** members, float* const* dst_dev5, int cap_rows,
                           int* n_rows);

/* ---- stand-alone box ops ----------------------------------------------------- */
/* nms(dets, thresh)  lib/nms/nms_wrapper.py:13 -> gpu_nms lib/nms/gpu_nms.pyx:16-31
 * -> _nms lib/nms/gpu_nms.hpp:1 (nms_kernel.cu:102-155).  Takes UNSORTED dets
 * (n,5) fp32 host memory and returns indices into it, score-descending, like
 * gpu_nms.pyx:31 (`order[keep]`).  Suppression predicate IoU > thresh
 * (nms_kernel.cu:82).  keep must hold n ints. */
int shf_nms(const float* dets5, int n, float thresh, int device_id, int32_t* keep, int* n_keep);
/* bbox_vote(det)     lib/test.py:181-217, thresh = cfg.TEST.NMS_THRESH.
 * dets5 (n,5) fp32 host; out5 (cap,5) double. */
int shf_bbox_vote(const float* dets5, int n, float thresh, double* out5, int cap, int* n_out);
/* generate_anchors() lib/layers/generate_anchors.py:11-24; out (n_ratios*n_scales*n_shifts^2, 4) */
int shf_generate_anchors(int base_size, const double* ratios, int n_ratios, const double* scales,
                         int n_scales, const double* shifts, int n_shifts, const double* strides,
                         double* out, int cap_rows);

/* Read parameter blob `idx` of layer `layer` from a binary .caffemodel (the reader behind
 * shf_net_create's weight loading; caffe.proto NetParameter.layer=100 / BlobProto data=5).
 * Returns the element count (out may be NULL to query), fills dims/ndim.  Needs no GPU. */
int shf_caffemodel_read_blob(const char* path, const char* layer, int idx, float* out, int cap,
                             int* dims, int* ndim);

/* ---- diagnostics (tests) ------------------------------------------------------ */
/* runs the merge pipeline on host dets and returns its intermediates: the IoU bit matrix
 * (n x ceil(n/64) words, upper triangle), cluster head per box, kept heads, sorted dets, perm */
int shf_debug_merge(const float* dets5, int n, float thresh, int ge_pred, unsigned long long* mask_out,
                    int* cluster_out, int* heads_out, int* n_heads, float* sorted_out, int* perm_out);

/* The proposal stage alone -- tail decode -> candidate select -> sort -> gather -- on INJECTED blobs, so the
 * reference's own ProposalLayer.forward vectors (lib/layers/proposal_layer.py:60-220, incl. the
 * np.seterr(over='raise') clamp branch of lib/utils/bbox_transform.py:52-65) reach the HIP kernels: scores =
 * cls_prob_reshape_output (1,2A,h,w), deltas = bbox_pred_output (1,4A,h,w), im_info3 = (h, w, scale), all host
 * fp32.  Anchors and cfg.TEST.* come from `net` (any net holding the proposal layer).  out_boxes5 (cap,5) /
 * out_probs2 (cap,2) receive the layer's tops: *n_out = R rows (when R == 0 out_boxes5 holds the dummy roi);
 * *overflow != 0 when the clamp branch was taken. */
int shf_debug_proposal(shf_net* net, const float* scores, const float* deltas, int h, int w, const float* im_info3,
                       float* out_boxes5, float* out_probs2, int cap, int* n_out, int* overflow);
/* forward_net's flip fix + unscale (lib/test.py:52-54,59-66) and detect()'s > thresh cut (:163-167) on injected
 * proposals boxes5 (R,5) / probs2 (R,2) (host, score-descending like the layer emits them): appended to the
 * current image's device list as one more unit (after shf_detect_begin; read back with shf_detect_export). */
int shf_debug_append(shf_net* net, const float* boxes5, const float* probs2, int R, int im_w, float im_scale,
                     int flip, float thresh);

/* ---- measurement ------------------------------------------------------------- */
/* Per-kernel-class HIP-event timing on the net's stream.  enable!=0 starts recording
 * an event pair around every launch; shf_prof_read drains them (synchronises) and
 * returns, for class `cls`, the number of launches, total ms and algorithmic FLOPs. */
int shf_prof_enable(shf_net* net, int enable);
/* Restrict the bracketing to launches of ONE class (cls < 0: every class again).  An event pair around a launch costs the
 * stream a few microseconds; bench.py surveys every class in an untimed pass and brackets only the dominant kernel inside
 * its timed region (measured: 79.5 images/s with every launch bracketed, 82.3 with none). */
int shf_prof_only(shf_net* net, int cls);
int shf_prof_num_classes(shf_net* net);
const char* shf_prof_class_name(shf_net* net, int cls);
int shf_prof_read(shf_net* net, int cls, int64_t* launches, double* total_ms, double* flops, double* bytes);
int shf_prof_reset(shf_net* net);
/* Calibration of the roofline's MFMA bound on the box at hand: the issued TFLOP/s a pure stream of
 * v_mfma_f32_32x32x16_f16 (bf16 != 0: _bf16) sustains with the convolution kernels' register diet (one wave per SIMD,
 * 8 accumulator tiles, 3 products per fragment pair) and NO memory traffic, for operands with random signs and
 * mantissas of which zero_eighths/8 of the activation fragments are zero (constant_operands != 0: every element 1.0 --
 * the nominal-peak case).  The chip is power-limited under matrix load and the limit depends on the operands' bit
 * toggling, so this -- not the 2.5 PFLOP/s of non-toggling operands -- is what a kernel that moved its bytes for free
 * could reach.  Runs (reps + 1) / 2 settling launches, then times `reps` launches of `iters` x 24 MFMAs per wave on a
 * private stream; synchronises.  No counterpart in the reference (measurement only). */
int shf_calib_matrix_pipe(int bf16, int zero_eighths, int constant_operands, int iters, int reps, double* tflops);
/* stream synchronise (end of a timed region) */
int shf_net_sync(shf_net* net);

#ifdef __cplusplus
}
#endif
#endif /* SHF_HIP_H_ */
