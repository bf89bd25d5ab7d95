// Net runtime, part 3: the fused per-image detection pipeline of the C ABI (shf_detect_*): pyramid levels on the device,
// single units and grouped passes over an image's units, per-unit appends, export / import between lanes and ranks, merge.
#include "net_internal.h"

extern "C" {

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


}  // extern "C"

// append a group of finished units (their proposals sit in the members' output blobs) to `net`'s image list -- or,
// per_member, to each member's own (reset) list -- in ONE launch
void append_units(shf_net* net, shf_net* const* srcs, int n, const int* im_w, const float* im_scale,
                  const int* flip, float thresh, bool per_member, hipStream_t st, Prof* pf) {
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


extern "C" {

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

}  // extern "C"
