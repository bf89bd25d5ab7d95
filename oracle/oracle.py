"""CPU oracle: a numpy restatement of the reference's inference hot path.

*** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg may import this module; it is the CHECKER, never
the thing measured or shipped.  The product path (smallhardface_amd/) never
imports it and fails loudly when the HIP library is missing.

Pinning status
  * Python half (anchors, ProposalLayer, bbox_transform_inv/clip, bbox_vote,
    py_cpu_nms, forward_net post-processing, pyramid scales): PINNED against
    golden vectors produced by the reference's own code
    (tests/golden/*.npz, generator tests/golden/make_golden.py,
    checked in tests/test_oracle_golden.py).
  * Caffe C++ half (conv/relu/pool/deconv/concat/softmax/reshape/Net wiring):
    the vendored Caffe needs boost/glog/gflags/protobuf/BLAS/CUDA and is
    UNBUILDABLE here; its SGEMM lives in third-party OpenBLAS
    (caffe/Makefile.config:50, version unpinned).  This restatement follows the
    cited Caffe sources and is pinned by Caffe's own known-answer tests restated
    in tests/test_oracle_caffe_ops.py (max-pool literal, deconv overlap counts,
    bilinear filler formula, naive-loop conv incl. dilation, softmax).
    End-to-end network output: PARITY UNPINNED (no trained weights and no golden
    activations exist in the reference tree).

All arithmetic is fp32 like pycaffe (``_caffe.cpp:46-48``), NCHW, batch as given.
"""
import numpy as np

F32 = np.float32


# ==========================================================================
# Caffe layer restatements
# ==========================================================================
def conv_out_size(n, k, pad, stride, dil):
    """caffe/src/caffe/layers/conv_layer.cpp:8-22."""
    kext = dil * (k - 1) + 1
    return (n + 2 * pad - kext) // stride + 1


def im2col(x, kh, kw, pad, stride, dil, y0=0, y1=None):
    """caffe/src/caffe/util/im2col.cpp:19-55 for output rows [y0,y1): (C*kh*kw, rows*Wo)."""
    C, H, W = x.shape
    Ho = conv_out_size(H, kh, pad, stride, dil)
    Wo = conv_out_size(W, kw, pad, stride, dil)
    if y1 is None:
        y1 = Ho
    xp = np.zeros((C, H + 2 * pad, W + 2 * pad), dtype=F32)
    xp[:, pad:pad + H, pad:pad + W] = x
    col = np.empty((C, kh, kw, y1 - y0, Wo), dtype=F32)
    for ky in range(kh):
        for kx in range(kw):
            ys = y0 * stride + ky * dil
            col[:, ky, kx] = xp[:, ys: ys + (y1 - y0 - 1) * stride + 1: stride,
                                kx * dil: kx * dil + (Wo - 1) * stride + 1: stride]
    return col.reshape(C * kh * kw, (y1 - y0) * Wo)


def convolution(x, w, b, pad=0, stride=1, dil=1, group=1, col_bytes=256 << 20):
    """ConvolutionLayer::Forward_cpu (conv_layer.cpp:25-40) = per image
    forward_cpu_gemm (base_conv_layer.cpp:256-271: im2col then W @ col, skipped
    im2col for 1x1/s1/p0) + forward_cpu_bias (:273-279).  The GEMM is whatever
    BLAS numpy carries (OpenBLAS sgemm, as ``BLAS := open`` in the reference).
    Output rows are processed in chunks only to bound the col buffer."""
    N, C, H, W = x.shape
    Co, Cg, kh, kw = w.shape
    assert C == Cg * group and Co % group == 0
    Ho = conv_out_size(H, kh, pad, stride, dil)
    Wo = conv_out_size(W, kw, pad, stride, dil)
    y = np.empty((N, Co, Ho, Wo), dtype=F32)
    wmat = np.ascontiguousarray(w.reshape(group, Co // group, Cg * kh * kw), dtype=F32)
    rows = max(1, min(Ho, col_bytes // max(1, C * kh * kw * Wo * 4)))
    for n in range(N):
        for y0 in range(0, Ho, rows):
            y1 = min(Ho, y0 + rows)
            for g in range(group):
                xg = x[n, g * Cg:(g + 1) * Cg]
                if kh == 1 and kw == 1 and pad == 0 and stride == 1:
                    col = xg[:, y0:y1].reshape(Cg, -1)
                else:
                    col = im2col(xg, kh, kw, pad, stride, dil, y0, y1)
                out = wmat[g] @ col
                y[n, g * (Co // group):(g + 1) * (Co // group), y0:y1] = out.reshape(-1, y1 - y0, Wo)
        if b is not None:
            y[n] += b.astype(F32)[:, None, None]
    return y


def relu(x):
    """relu_layer.cpp:9-19 with negative_slope 0."""
    return np.maximum(x, F32(0))


def max_pool(x, k=2, stride=2, pad=0):
    """PoolingLayer::Reshape/Forward_cpu MAX (pooling_layer.cpp:79-123,128-187):
    ceil output size, windows clipped to the input, first max wins."""
    N, C, H, W = x.shape
    Ho = int(np.ceil((H + 2 * pad - k) / float(stride))) + 1
    Wo = int(np.ceil((W + 2 * pad - k) / float(stride))) + 1
    if pad:
        if (Ho - 1) * stride >= H + pad:
            Ho -= 1
        if (Wo - 1) * stride >= W + pad:
            Wo -= 1
    y = np.full((N, C, Ho, Wo), -np.finfo(F32).max, dtype=F32)
    for ph in range(Ho):
        hs = max(ph * stride - pad, 0)
        he = min(ph * stride - pad + k, H)
        for pw in range(Wo):
            ws = max(pw * stride - pad, 0)
            we = min(pw * stride - pad + k, W)
            y[:, :, ph, pw] = x[:, :, hs:he, ws:we].max(axis=(2, 3))
    return y


def max_pool_2x2_fast(x):
    """Vectorised special case of ``max_pool`` (k=2,s=2,p=0) for big maps."""
    N, C, H, W = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    xp = np.full((N, C, Ho * 2, Wo * 2), -np.finfo(F32).max, dtype=F32)
    xp[:, :, :H, :W] = x
    return xp.reshape(N, C, Ho, 2, Wo, 2).max(axis=(3, 5))


def bilinear_filler(shape):
    """BilinearFiller::Fill (include/caffe/filler.hpp:248-258)."""
    w = np.empty(shape, dtype=F32)
    kw = shape[3]
    kh = shape[2]
    f = int(np.ceil(kw / 2.))
    c = F32((kw - 1) / (2. * f))
    flat = w.reshape(-1)
    for i in range(flat.size):
        x = F32(i % kw)
        yv = F32((i // kw) % kh)
        flat[i] = (F32(1) - abs(x / F32(f) - c)) * (F32(1) - abs(yv / F32(f) - c))
    return w


def deconvolution(x, w, b, pad, stride, group):
    """DeconvolutionLayer::Forward_cpu (deconv_layer.cpp:8-40): per image
    backward_cpu_gemm (col = W^T x ; base_conv_layer.cpp:281-297) then col2im
    (im2col.cpp:153-…).  w: (Cin, Cout/group, kh, kw)."""
    N, C, H, W = x.shape
    Ci, Cog, kh, kw = w.shape
    assert Ci == C
    Cig = C // group
    Co = Cog * group
    Ho = stride * (H - 1) + kh - 2 * pad
    Wo = stride * (W - 1) + kw - 2 * pad
    y = np.zeros((N, Co, Ho + 2 * pad, Wo + 2 * pad), dtype=F32)
    for n in range(N):
        for g in range(group):
            xg = x[n, g * Cig:(g + 1) * Cig].reshape(Cig, H * W)
            wg = w[g * Cig:(g + 1) * Cig].reshape(Cig, Cog * kh * kw)
            col = (wg.T @ xg).reshape(Cog, kh, kw, H, W).astype(F32)
            for a in range(kh):
                for bb in range(kw):
                    y[n, g * Cog:(g + 1) * Cog, a: a + stride * (H - 1) + 1: stride,
                      bb: bb + stride * (W - 1) + 1: stride] += col[:, a, bb]
    y = y[:, :, pad:pad + Ho, pad:pad + Wo]
    if b is not None:
        y = y + b.astype(F32)[None, :, None, None]
    return np.ascontiguousarray(y, dtype=F32)


def softmax(x, axis=1):
    """SoftmaxLayer::Forward_cpu (softmax_layer.cpp:27-60): max, subtract, exp, sum, divide."""
    m = x.max(axis=axis, keepdims=True)
    e = np.exp((x - m).astype(F32)).astype(F32)
    return (e / e.sum(axis=axis, keepdims=True, dtype=F32)).astype(F32)


def caffe_reshape(shape_in, dims):
    """ReshapeLayer::Reshape (reshape_layer.cpp:32-91): 0 copies, -1 infers."""
    out = []
    infer = -1
    for i, d in enumerate(dims):
        if d == 0:
            out.append(shape_in[i])
        elif d == -1:
            infer = i
            out.append(1)
        else:
            out.append(d)
    if infer >= 0:
        out[infer] = int(np.prod(shape_in)) // int(np.prod(out))
    return tuple(out)


# ==========================================================================
# lib/ Python half
# ==========================================================================
def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32), shifts=(0,), strides=(0,)):
    """lib/layers/generate_anchors.py:11-86 (float64, like the reference)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    shifts = np.asarray(shifts, dtype=np.float64)

    def whctrs(a):
        w = a[2] - a[0] + 1
        h = a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws = np.atleast_1d(ws)[:, None]
        hs = np.atleast_1d(hs)[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    base = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, xc, yc = whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = mk(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        for j, s in zip(scales, strides):
            w, h, xc, yc = whctrs(ratio_anchors[i])
            a = mk(np.array([w * j]), np.array([h * j]), xc, yc)
            sx, sy = np.meshgrid(shifts * s, shifts * s)
            mesh = np.vstack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()]).T
            out.append(a + mesh)
    return np.vstack(out)


def bbox_transform_inv(boxes, deltas):
    """lib/utils/bbox_transform.py:33-77, including the overflow/clamp branch
    (np.seterr(over='raise') at :9 turns an fp32 exp/multiply overflow into the
    clamp of every dw,dh > 50 to 5)."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    deltas = deltas.copy()
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx, dy, dw, dh = deltas[:, 0::4], deltas[:, 1::4], deltas[:, 2::4], deltas[:, 3::4]
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    with np.errstate(over='raise'):
        try:
            pred_w = np.exp(dw) * widths[:, None]
            pred_h = np.exp(dh) * heights[:, None]
        except FloatingPointError:
            dw[dw > 50] = 5
            dh[dh > 50] = 5
            pred_w = np.exp(dw) * widths[:, None]
            pred_h = np.exp(dh) * heights[:, None]
    pred = np.zeros(deltas.shape, dtype=deltas.dtype)
    pred[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    pred[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    pred[:, 2::4] = pred_ctr_x + 0.5 * pred_w
    pred[:, 3::4] = pred_ctr_y + 0.5 * pred_h
    return pred


def clip_boxes(boxes, im_shape):
    """lib/utils/bbox_transform.py:80-93."""
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def canonical_order(score):
    """Deterministic total order used wherever the reference relies on
    ``argsort()[::-1]`` (tie order implementation-defined): score descending,
    then original index ascending."""
    idx = np.arange(score.shape[0])
    return np.lexsort((idx, -score.astype(np.float64)))


class ProposalParams(object):
    def __init__(self, feat_stride=(8, 8, 8), scales=(1, 2, 4), ratios=(1,), base_size=16,
                 shifts=(0,), subsampled=True, num_feats=1,
                 pre_nms_topN=10000, score_thresh=0.002, min_size=0):
        self.feat_stride = list(feat_stride)
        self.scales = list(scales)
        self.ratios = list(ratios)
        self.base_size = base_size
        self.shifts = list(shifts)
        self.subsampled = subsampled
        self.num_feats = num_feats
        self.pre_nms_topN = pre_nms_topN
        self.score_thresh = score_thresh
        self.min_size = min_size


def proposal_forward(scores, bbox_deltas, im_info, pp=None):
    """ProposalLayer.forward, TEST phase (lib/layers/proposal_layer.py:60-220).

    scores (1,2A,h,w) [bg block then fg block], bbox_deltas (1,4A,h,w), im_info (1,3)
    -> boxes (R,5) [0,x1,y1,x2,y2], cls_prob (R,2).  Ties are ordered canonically.
    """
    pp = pp or ProposalParams()
    anchors0 = generate_anchors(base_size=pp.base_size, ratios=pp.ratios, scales=pp.scales,
                                shifts=pp.shifts, strides=pp.feat_stride)
    A = anchors0.shape[0]
    assert scores.shape[0] == 1, 'Only single item batches are supported'
    im_info = im_info[0, :]
    height, width = scores.shape[-2:]
    shift_x = np.arange(0, width) * pp.feat_stride[0]
    shift_y = np.arange(0, height) * pp.feat_stride[0]
    shift_x, shift_y = np.meshgrid(shift_x, shift_y)
    shifts = np.vstack((shift_x.ravel(), shift_y.ravel(), shift_x.ravel(), shift_y.ravel())).transpose()
    K = shifts.shape[0]
    num_classes = scores.shape[1] // (A * pp.num_feats)
    anchors = anchors0.reshape((1, A, 4)) + shifts.reshape((1, K, 4)).transpose((1, 0, 2))
    anchors = anchors.reshape((K * A, 4))
    deltas = bbox_deltas.transpose((0, 2, 3, 1)).reshape((-1, 4))
    sc = scores.transpose((0, 2, 3, 1)).reshape((-1, num_classes, A * pp.num_feats)) \
        .transpose((0, 2, 1)).reshape((-1, num_classes))
    proposals = bbox_transform_inv(anchors, deltas)
    proposals = clip_boxes(proposals, im_info[:2])
    if pp.subsampled:
        amap = np.zeros((height, width, A))
        for i in range(A):
            stride = pp.feat_stride[i // len(pp.shifts) ** 2] // pp.feat_stride[0]
            amap[::stride, ::stride, i] = 1
        inds = np.where(amap.reshape(K * A))[0]
        proposals = proposals[inds, :]
        sc = sc[inds, :]
    ws = proposals[:, 2] - proposals[:, 0] + 1
    hs = proposals[:, 3] - proposals[:, 1] + 1
    ms = pp.min_size * im_info[2]
    keep = np.where((ws >= ms) & (hs >= ms))[0]
    proposals = proposals[keep, :]
    sc = sc[keep, :]
    max_score = np.max(sc[:, 1:], axis=1).ravel()
    order = canonical_order(max_score)
    ge = np.where(max_score[order] >= pp.score_thresh)[0]
    thresh_idx = ge.max() if ge.size else 0
    if pp.pre_nms_topN > 0:
        order = order[:pp.pre_nms_topN]
    order = order[:thresh_idx + 1]
    proposals = proposals[order, :]
    sc = sc[order, :]
    if proposals.shape[0] == 0:
        blob = np.array([[0, 0, 0, 16, 16]], dtype=F32)
    else:
        blob = np.hstack((np.zeros((proposals.shape[0], 1), dtype=F32), proposals.astype(F32, copy=False)))
    return blob.astype(F32), sc.astype(F32)


def iou_row(box, boxes):
    """fp32 IoU with the +1 pixel convention, op order of devIoU
    (lib/nms/nms_kernel.cu:24-32) == py_cpu_nms.py:17,25-32 == test.py:188-196."""
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    a0 = (box[2] - box[0] + 1) * (box[3] - box[1] + 1)
    xx1 = np.maximum(box[0], boxes[:, 0])
    yy1 = np.maximum(box[1], boxes[:, 1])
    xx2 = np.minimum(box[2], boxes[:, 2])
    yy2 = np.minimum(box[3], boxes[:, 3])
    w = np.maximum(F32(0.0), xx2 - xx1 + 1)
    h = np.maximum(F32(0.0), yy2 - yy1 + 1)
    inter = w * h
    return inter / (a0 + area - inter)


def bbox_vote(det, thresh=0.4, order=None):
    """bbox_vote (lib/test.py:181-217).  det (N,5) fp32.  Same arithmetic: fp32
    products, numpy row-order sum of the weighted boxes, numpy pairwise sum of the
    scores, fp32 divide, result widened to fp64.  Ties ordered canonically."""
    det = np.asarray(det)
    if order is None:
        order = canonical_order(det[:, 4].ravel()) if det.shape[0] else np.zeros(0, np.int64)
    det = det[order, :]
    dets = None
    if det.shape[0] == 0:
        return np.array([[10, 10, 20, 20, 0.0001]])
    while det.shape[0] > 0:
        o = iou_row(det[0], det)
        merge_index = np.where(o >= thresh)[0]
        det_accu = det[merge_index, :]
        det = np.delete(det, merge_index, 0)
        if merge_index.shape[0] <= 1:
            if det.shape[0] == 0:
                dets = det_accu if dets is None else np.vstack((dets, det_accu))
            continue
        det_accu[:, 0:4] = det_accu[:, 0:4] * np.tile(det_accu[:, -1:], (1, 4))
        max_score = np.max(det_accu[:, 4])
        s = np.zeros((1, 5))
        s[:, 0:4] = np.sum(det_accu[:, 0:4], axis=0) / np.sum(det_accu[:, -1:])
        s[:, 4] = max_score
        dets = s if dets is None else np.vstack((dets, s))
    return dets


def nms(dets, thresh, order=None):
    """Greedy NMS with the canonical ``IoU > thresh`` suppression
    (lib/nms/nms_kernel.cu:82 == lib/nms/py_cpu_nms.py:35 keeps ``ovr <= thresh``).
    Returns indices into the unsorted input (gpu_nms.pyx:31)."""
    dets = np.asarray(dets, dtype=F32)
    if dets.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    if order is None:
        order = canonical_order(dets[:, 4])
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        ovr = iou_row(dets[i], dets[order[1:]])
        order = order[np.where(ovr <= thresh)[0] + 1]
    return np.asarray(keep, dtype=np.int64)


def nms_ge(dets, thresh):
    """The Cython variant's ``ovr >= thresh`` predicate (lib/nms/cpu_nms.pyx:65)."""
    dets = np.asarray(dets, dtype=F32)
    if dets.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    order = canonical_order(dets[:, 4])
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        ovr = iou_row(dets[i], dets[order[1:]])
        order = order[np.where(~(ovr >= thresh))[0] + 1]
    return np.asarray(keep, dtype=np.int64)


# ==========================================================================
# Net runtime restatement (caffe/src/caffe/net.cpp) on a parsed prototxt
# ==========================================================================
class OBlob(object):
    def __init__(self, shape=(1,)):
        self.data = np.zeros(shape, dtype=F32)

    def reshape(self, *dims):
        if tuple(dims) != self.data.shape:
            self.data = np.zeros(dims, dtype=F32)

    @property
    def shape(self):
        return self.data.shape


def _ints(msg, name, default):
    v = msg.getall(name) if msg is not None else []
    return int(v[0]) if v else default


from smallhardface_amd.weights import synth_params, infer_channels  # noqa: E402,F401  (shared generator)


class OracleNet(object):
    """numpy restatement of caffe.Net for the layer types of the test graph
    (Net::Init net.cpp:44-257, ForwardFromTo :516-532, param sharing :421-513,
    legacy input upgrade upgrade_proto.cpp:966-1000).  Same Python surface as
    the pycaffe subset lib/test.py uses."""

    TEST = 1

    def __init__(self, net_msg, params=None, proposal_cfg=None):
        from collections import OrderedDict
        self.msg = net_msg
        self.layers = net_msg.getall("layer")
        self.blobs = OrderedDict()
        self.inputs = [str(n) for n in net_msg.getall("input")]
        for n, shp in zip(self.inputs, net_msg.getall("input_shape")):
            self.blobs[n] = OBlob(tuple(int(d) for d in shp.getall("dim")))
        for L in self.layers:
            if str(L.get("type")) == "Input":
                for tp, shp in zip(L.getall("top"), L.get("input_param").getall("shape")):
                    self.inputs.append(str(tp))
                    self.blobs[str(tp)] = OBlob(tuple(int(d) for d in shp.getall("dim")))
        consumed = set()
        for L in self.layers:
            for b in L.getall("bottom"):
                consumed.add(str(b))
            for tp in L.getall("top"):
                if str(tp) not in self.blobs:
                    self.blobs[str(tp)] = OBlob()
        produced_last = {}
        for L in self.layers:
            for tp in L.getall("top"):
                produced_last[str(tp)] = True
        # Net::Init available_blobs (net.cpp:95-110,240-246): name-ordered set
        avail = set(self.inputs)
        for L in self.layers:
            if str(L.get("type")) == "Input":
                continue
            for b in L.getall("bottom"):
                avail.discard(str(b))
            for tp in L.getall("top"):
                avail.add(str(tp))
        self.outputs = sorted(avail)
        self.params = params if params is not None else synth_params(net_msg)
        self.pp = proposal_cfg or ProposalParams()

    def forward(self, **kwargs):
        if kwargs:
            if set(kwargs.keys()) != set(self.inputs):
                raise Exception('Input blob arguments do not match net inputs.')
            for in_, blob in kwargs.items():
                if blob.shape[0] != self.blobs[in_].shape[0]:
                    raise Exception('Input is not batch sized')
                self.blobs[in_].data[...] = blob
        B = self.blobs
        for L in self.layers:
            t = str(L.get("type"))
            name = str(L.get("name"))
            bots = [B[str(b)].data for b in L.getall("bottom")]
            tops = [str(x) for x in L.getall("top")]
            if t == "Input":
                continue
            if t == "Convolution":
                cp = L.get("convolution_param")
                p = self.params[name]
                out = convolution(bots[0], p[0], p[1] if len(p) > 1 else None,
                                  pad=_ints(cp, "pad", 0), stride=_ints(cp, "stride", 1),
                                  dil=_ints(cp, "dilation", 1), group=_ints(cp, "group", 1))
            elif t == "Deconvolution":
                cp = L.get("convolution_param")
                p = self.params[name]
                out = deconvolution(bots[0], p[0], p[1] if len(p) > 1 else None,
                                    pad=_ints(cp, "pad", 0), stride=_ints(cp, "stride", 1),
                                    group=_ints(cp, "group", 1))
            elif t == "ReLU":
                out = relu(bots[0])
            elif t == "Pooling":
                pq = L.get("pooling_param")
                assert str(pq.get("pool", "MAX")) == "MAX"
                k, s, pd = _ints(pq, "kernel_size", 2), _ints(pq, "stride", 1), _ints(pq, "pad", 0)
                out = max_pool_2x2_fast(bots[0]) if (k, s, pd) == (2, 2, 0) else max_pool(bots[0], k, s, pd)
            elif t == "Concat":
                out = np.concatenate(bots, axis=_ints(L.get("concat_param"), "axis", 1))
            elif t == "Softmax":
                out = softmax(bots[0], axis=_ints(L.get("softmax_param"), "axis", 1))
            elif t == "Reshape":
                dims = [int(d) for d in L.get("reshape_param").get("shape").getall("dim")]
                out = bots[0].reshape(caffe_reshape(bots[0].shape, dims))
            elif t == "Python":
                assert str(L.get("python_param").get("layer")) == "ProposalLayer"
                pstr = _parse_param_str(str(L.get("python_param").get("param_str")))
                pp = ProposalParams(feat_stride=pstr.get("feat_stride"), scales=pstr.get("scales", (8, 16, 32)),
                                    ratios=pstr.get("ratios", (0.5, 1, 2)), base_size=pstr.get("base_size", 16),
                                    shifts=pstr.get("shifts", [0]), subsampled=pstr.get("subsampled", True),
                                    num_feats=pstr.get("num_feats", 1), pre_nms_topN=self.pp.pre_nms_topN,
                                    score_thresh=self.pp.score_thresh, min_size=self.pp.min_size)
                boxes, probs = proposal_forward(bots[-3], bots[-2], bots[-1], pp)
                B[tops[0]].data = boxes
                if len(tops) > 1:
                    B[tops[1]].data = probs
                continue
            else:
                raise NotImplementedError("oracle: layer type %s" % t)
            B[tops[0]].data = np.ascontiguousarray(out, dtype=F32)
        return {o: B[o].data for o in self.outputs}


def _parse_param_str(s):
    import yaml
    return yaml.safe_load(s)


def flops_per_level(H, W):
    """Algorithmic conv FLOPs of one pyramid level: 2*361460*H*W."""
    return 2.0 * 361460.0 * H * W
