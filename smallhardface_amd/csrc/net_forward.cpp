// Net runtime, part 2: the per-layer executor (forward_ops: one unit, or the arguments of one layer for a grouped launch),
// Net.forward() with its fp32 redo, Blob.data read-back, and the profiler's kernel classes.
#include "net_internal.h"

namespace shf {

const char* const kProfNames[PC_COUNT] = {"conv_mfma_f32_kernel<3, 1, 128, 8, 16>", "conv_mfma_f32_kernel<3, 2, 128, 8, 16>",
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
                                           "deconv_depthwise", "detect_tail", "box_merge", "layout", "h2d_copy", "d2h_copy"};

// which split-fp16 kernel launch_conv_f16x3_group picks for these arguments -- decided by the launcher's OWN predicates on the
// actual arguments, so that the 8-wave fallbacks (unaligned views, Cout % 256, bf16 1x1s ...) are not booked under the name of
// the kernel the knobs would normally select
int f16x3_prof_class(const ConvArgs& a, int nout, const ConvArgs* group, int n) {
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

int conv_prof_class(int k, int dil, int nout) {
  const int bn64 = (nout % 128 == 0) ? 0 : 1;
  if (k == 1) return 6 + bn64;
  const int d = dil == 1 ? 0 : dil == 2 ? 1 : 2;
  return bn64 * 3 + d;
}
}  // namespace shf

double conv_flops(const Layer& L, const std::vector<int>& in, const std::vector<int>& out) {
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
  if (materialize_tail && (!fused_path || in_net_forward)) {
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
        if (plain_skip_tail) break;   // (ensure_plain: the intermediates only -- the tail's outputs stay the forward's own)
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

// the fused path's kernels behind Net.forward(): a split-fp16 mode, a detector graph whose outputs are the proposal
// layer's (nothing else is an output: a conv top that is a net output must hold plain fp32 after forward())
bool shf_net::forward_fast_eligible() const {
  static const bool knob = !(getenv("SHF_FORWARD_FAST") && atoi(getenv("SHF_FORWARD_FAST")) == 0);
  if (!knob || conv_mode < 1 || tail_layer < 0 || data_blob < 0) return false;
  for (int o : outputs)
    if (o != boxes_blob && o != prob_blob) return false;
  return true;
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
      ProfScope ps(prof, stream, PC_H2D, 0, 4.0 * b.count());
      HIP_THROW(hipMemcpyAsync(b.dev.p, b.host.p, b.count() * 4, hipMemcpyHostToDevice, stream));
      b.host_newer = false;
    }
  }
  float ii[3] = {0, 0, 1};
  if (im_info_blob >= 0 && blobs[im_info_blob].host.p && blobs[im_info_blob].count() >= 3)
    memcpy(ii, blobs[im_info_blob].host.p, 12);
  memcpy(last_im_info, ii, 12);
  inputs_reshaped = false;
  const bool fast = forward_fast_eligible();
  struct Scope {   // (forward_ops may throw)
    bool& f;
    explicit Scope(bool& f_) : f(f_) { f = true; }
    ~Scope() { f = false; }
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (conv_mode >= 1) HIP_THROW(hipMemsetAsync(range_flag.p, 0, 4, stream));
    reset_amax(stream);
    {
      Scope sc(in_net_forward);
      forward_ops(fast, ii[0], ii[1], ii[2]);
    }
    plain_stale = fast;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flag = 0;
    {
      ProfScope ps(prof, stream, PC_D2H, 0, sizeof(cnt) + 4);
      if (tail_layer >= 0) HIP_THROW(hipMemcpyAsync(cnt, tw.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
      if (conv_mode >= 1) HIP_THROW(hipMemcpyAsync(&flag, range_flag.p, 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_THROW(hipStreamSynchronize(stream));
    if (flag && conv_mode >= 1) {
      const int mode_was = conv_mode;
      // a convolution produced |x| > 65504: fp16(hi) of the split overflowed somewhere downstream.  The reference
      // computes in fp32 (_caffe.cpp:46-48): redo THIS forward on the exact fp32 matrix-core kernels (per-layer path:
      // every blob materialised).
      ++sh->range_fallbacks;
      conv_mode = 0;
      try {
        reset_amax(stream);
        forward_ops(false, ii[0], ii[1], ii[2]);
        plain_stale = false;
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
  // (tail-fused blobs too: "newer" for them means the tail workspace holds this forward's logits -- read on demand)
  for (size_t i = 0; i < blobs.size(); ++i)
    if (!std::count(inputs.begin(), inputs.end(), (int)i)) blobs[i].dev_newer = true;
}

// the intermediate blobs after a fast forward (see net_internal.h `plain_stale`): run the per-layer kernels once, in the
// forward's own arithmetic mode, on the inputs still resident on the device -- everything except the proposal tail
void shf_net::ensure_plain() {
  if (!plain_stale) return;
  if (inputs_reshaped || (data_blob >= 0 && blobs[data_blob].shape != last_data_shape))
    throw std::runtime_error("an input was reshaped after the last forward(): call forward() before reading intermediate blobs");
  struct Scope {
    bool& f;
    explicit Scope(bool& f_) : f(f_) { f = true; }
    ~Scope() { f = false; }
  } sc(plain_skip_tail);
  reset_amax(stream);
  forward_ops(false, last_im_info[0], last_im_info[1], last_im_info[2]);
  HIP_THROW(hipStreamSynchronize(stream));
  plain_stale = false;
}

// Blob.data of a blob whose producer was folded into the detection tail (the cls / bbox 1x1 convs, the score concat /
// reshape, the softmax): pycaffe exposes every blob after forward() (pycaffe.py:24-32, _caffe.cpp:222-242), and someone
// debugging through the shim reads them.  Nothing extra is computed in forward(): the logits kernel leaves
// [K][A][cls0, cls1, dx, dy, dw, dh] in the tail workspace and the tail writes the softmax as the (1, 2A, h, w) blob the
// proposal layer reads; the read-back re-orders those on the host into the blob's own NCHW shape.
void shf_net::materialize_fused(int bi) {
  Blob& b = blobs[bi];
  const size_t n = b.count();
  b.host.ensure(std::max<size_t>(n, 1) * 4);
  if (!b.dev_newer || n == 0) return;          // (never forwarded: zeros, like a Caffe blob before its first forward)
  const int A = tail_A;
  const int h = blobs[tail_feat_blobs[0]].shape[2], w = blobs[tail_feat_blobs[0]].shape[3];
  const size_t K = (size_t)h * w;
  float* out = b.host.p;
  if (b.fused_role == FR_PROB_PLANES) {
    if (n != K * A * 2) throw std::runtime_error("blob '" + b.name + "': unexpected shape for the softmax output");
    HIP_THROW(hipMemcpyAsync(out, blobs[tail_cls_blob].dev.p, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_THROW(hipStreamSynchronize(stream));
    b.dev_newer = false;
    return;
  }
  std::vector<float> lg(K * A * 6);
  HIP_THROW(hipMemcpyAsync(lg.data(), tw.logits, lg.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_THROW(hipStreamSynchronize(stream));
  auto L = [&](size_t k, int a, int o) { return lg[(k * A + a) * 6 + o]; };
  const bool per_head = tail_heads != 1;
  switch (b.fused_role) {
    case FR_CLS_CONV:
      if (n != (per_head ? 2 : 2 * (size_t)A) * K) throw std::runtime_error("blob '" + b.name + "': unexpected shape");
      if (per_head) {
        for (int c = 0; c < 2; ++c)
          for (size_t k = 0; k < K; ++k) out[c * K + k] = L(k, b.fused_head, c);
      } else {
        for (int c = 0; c < 2; ++c)
          for (int a = 0; a < A; ++a)
            for (size_t k = 0; k < K; ++k) out[((size_t)c * A + a) * K + k] = L(k, a, c);
      }
      break;
    case FR_BOX_CONV:
      if (n != (per_head ? 4 : 4 * (size_t)A) * K) throw std::runtime_error("blob '" + b.name + "': unexpected shape");
      if (per_head) {
        for (int j = 0; j < 4; ++j)
          for (size_t k = 0; k < K; ++k) out[j * K + k] = L(k, b.fused_head, 2 + j);
      } else {
        for (int a = 0; a < A; ++a)
          for (int j = 0; j < 4; ++j)
            for (size_t k = 0; k < K; ++k) out[((size_t)a * 4 + j) * K + k] = L(k, a, 2 + j);
      }
      break;
    case FR_CLS_PLANES:   // (1, 2, A*h, w): plane c, rows a*h .. a*h + h - 1 = head / anchor a
      if (n != 2 * (size_t)A * K) throw std::runtime_error("blob '" + b.name + "': unexpected shape");
      for (int c = 0; c < 2; ++c)
        for (int a = 0; a < A; ++a)
          for (size_t k = 0; k < K; ++k) out[((size_t)c * A + a) * K + k] = L(k, a, c);
      break;
    default:
      throw std::runtime_error("blob '" + b.name + "' is fused into the detection tail and has no read-back rule");
  }
  b.dev_newer = false;
}

float* shf_net::host_data(int bi) {
  Blob& b = blobs[bi];
  if (b.kind == BK_FUSED) {
    materialize_fused(bi);
    return b.host.p;
  }
  const size_t n = b.count();
  b.host.ensure(std::max<size_t>(n, 1) * 4);
  const bool is_input = std::count(inputs.begin(), inputs.end(), bi) > 0;
  if (b.dev_newer && n > 0) {
    if (b.kind == BK_NHWC) {
      if (!is_input) ensure_plain();   // (after a fast forward: the activations are not plain fp32 tensors yet)
      b.stage.ensure(n * 4);
      {
        ProfScope ps(prof, stream, PC_LAYOUT, 0, 8.0 * n);
        CHECK_RC(launch_nhwc_to_nchw(view_of(bi), (float*)b.stage.p, stream));
      }
      ProfScope ps(prof, stream, PC_D2H, 0, 4.0 * n);
      HIP_THROW(hipMemcpyAsync(b.host.p, b.stage.p, n * 4, hipMemcpyDeviceToHost, stream));
    } else {
      ProfScope ps(prof, stream, PC_D2H, 0, 4.0 * n);
      HIP_THROW(hipMemcpyAsync(b.host.p, b.dev.p, n * 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_THROW(hipStreamSynchronize(stream));
    b.dev_newer = false;
  }
  if (is_input) b.host_newer = true;
  return b.host.p;
}
