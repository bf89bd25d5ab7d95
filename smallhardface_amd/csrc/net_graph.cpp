// Net runtime, part 1: the graph -- prototxt -> layers / blobs / parameter sharing, the fusion plans (conv + pool, fused
// first pair, split-format blobs, the three shared-weight heads), shape inference, buffers, parameter packs.
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

// generate_anchors.py:11-86 in double precision
void gen_anchors(int base_size, const std::vector<double>& ratios, const std::vector<double>& scales,
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
std::map<std::string, std::vector<double>> parse_param_str(const std::string& s) {
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

int conv_out(int n, int k, int pad, int stride, int dil) {
  const int kext = dil * (k - 1) + 1;
  return (n + 2 * pad - kext) / stride + 1;
}

}  // namespace shf

static int geti(const PMsg* m, const char* n, int d) {
  if (!m) return d;
  const long v = m->num(n, d);          // (a value outside int would wrap: clamp, the range checks below refuse it by name)
  return v > 2147483647L ? 2147483647 : (v < -2147483647L ? -2147483647 : (int)v);
}

// Blob::Reshape (blob.cpp:23-51): at most 32 axes, every dim >= 0, count <= INT_MAX
void shf::check_blob_dims(const std::vector<int>& shp, const std::string& name) {
  if (shp.size() > 32) throw std::runtime_error("blob '" + name + "': more than 32 axes");
  long long count = 1;
  for (int d : shp) {
    if (d < 0) throw std::runtime_error("blob '" + name + "': negative dimension");
    if (d != 0 && count > 2147483647LL / d) throw std::runtime_error("blob '" + name + "': size exceeds INT_MAX");
    count *= d;
  }
}

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
    check_blob_dims(shp, in_names[i]->scalar);
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
        check_blob_dims(shp, tops[i]->scalar);
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
      // BaseConvolutionLayer::LayerSetUp's CHECKs (base_conv_layer.cpp:21-120): Caffe aborts on these, here the graph is refused
      // with the layer's name -- a hostile prototxt must not reach a division by a zero stride or a 2^31-channel allocation
      if (L.nout < 1 || L.nout > (1 << 20)) throw std::runtime_error("layer '" + L.name + "': num_output must be in 1 .. 2^20");
      if (L.k < 1 || L.k > 64) throw std::runtime_error("layer '" + L.name + "': kernel_size must be in 1 .. 64");
      if (L.stride < 1 || L.stride > 64) throw std::runtime_error("layer '" + L.name + "': stride must be in 1 .. 64");
      if (L.dil < 1 || L.dil > 64) throw std::runtime_error("layer '" + L.name + "': dilation must be in 1 .. 64");
      if (L.pad < 0 || L.pad > 4096) throw std::runtime_error("layer '" + L.name + "': pad must be in 0 .. 4096");
      if (L.group < 1 || L.nout % L.group) throw std::runtime_error("layer '" + L.name + "': group must divide num_output");
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
      // PoolingLayer::LayerSetUp (pooling_layer.cpp:20-77): kernel > 0, stride > 0, pad < kernel
      if (L.k < 1 || L.k > 64 || L.stride < 1 || L.stride > 64 || L.pad < 0 || L.pad >= L.k)
        throw std::runtime_error("layer '" + L.name + "': pooling needs kernel_size 1 .. 64, stride 1 .. 64 and 0 <= pad < kernel_size");
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
      // what the top holds, for the on-demand read-back (shf_net::materialize_fused)
      int role = FR_NONE, head = -1;
      for (int i = 0; i < tail_heads; ++i) {
        if (tail_cls_layers[i] == li2) role = FR_CLS_CONV, head = i;
        if (tail_box_layers[i] == li2) role = FR_BOX_CONV, head = i;
      }
      if (role == FR_NONE && layers[li2].type == "Softmax") role = FR_PROB_PLANES;
      if (role == FR_NONE && (layers[li2].type == "Concat" || layers[li2].type == "Reshape")) role = FR_CLS_PLANES;
      for (int t : layers[li2].tops) {
        blobs[t].kind = BK_FUSED;
        blobs[t].fused_role = role;
        blobs[t].fused_head = head;
      }
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
    if (layers[d2].heads3_lead >= 0 || layers[d4].heads3_lead >= 0 || layers[d2].heads3_d2 >= 0 || layers[d4].heads3_d2 >= 0)
      continue;   // a sibling already claimed by another dilation-1 layer
    // the fused launch writes the siblings' tops at THIS layer's place in the schedule: legal only when nothing in between
    // reads or writes those tops, and nothing in between (or the siblings' own in-place ReLUs aside) rewrites the shared bottom
    bool safe = true;
    const int t2 = layers[d2].tops[0], t4 = layers[d4].tops[0], shared_bottom = A.bottoms[0];
    for (int lj = (int)li + 1; lj < std::max(d2, d4) && safe; ++lj) {
      if (lj == d2 || lj == d4) continue;
      const Layer& Q = layers[lj];
      // (the three convs' own in-place ReLUs are folded into them: Layer::relu)
      if (Q.op == OP_SKIP && Q.type == "ReLU" && Q.bottoms.size() == 1 && Q.tops.size() == 1 && Q.bottoms[0] == Q.tops[0] &&
          (Q.tops[0] == A.tops[0] || Q.tops[0] == t2 || Q.tops[0] == t4))
        continue;
      for (int bb : Q.bottoms) safe = safe && bb != t2 && bb != t4;
      for (int tt : Q.tops) safe = safe && tt != t2 && tt != t4 && tt != shared_bottom;
    }
    if (!safe) continue;
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
