"""pycaffe-compatible surface over libshf_hip.so -- exactly the subset the
reference's inference driver touches:

  caffe.set_mode_gpu / set_device / TEST      caffe/python/caffe/__init__.py:1-8, _caffe.cpp:394-396
  caffe.Net(proto, weights, phase)            _caffe.cpp:137-151,413
  net.blobs (OrderedDict, topological order)  pycaffe.py:24-32
  net.inputs / net.outputs / net.params       pycaffe.py:59-85
  net.forward(**inputs) + its two exceptions  pycaffe.py:88-134
  Blob.data (writable zero-copy fp32 view), .shape, .reshape(*dims), .count/.num/...
                                              _caffe.cpp:222-256,453-477
  caffe.Layer (param_str / phase attributes)  include/caffe/layers/python_layer.hpp:27-30

Solvers, backward, forward_all, io, Classifier, NCCL are not part of the hot path.
"""
import ctypes as C
import time
from collections import OrderedDict

import numpy as np

from .. import _lib

TRAIN = 0
TEST = 1

__all__ = ["Net", "Blob", "Layer", "TEST", "TRAIN", "set_mode_gpu", "set_mode_cpu", "set_device"]


def set_mode_gpu():
    _lib.check(_lib.load().shf_set_mode_gpu(), "set_mode_gpu")


def set_mode_cpu():
    raise _lib.ShfError("smallhardface_amd is a GPU-only runtime (no CPU mode)")


def set_device(device_id):
    _lib.check(_lib.load().shf_set_device(int(device_id)), "set_device")


# arithmetic of the MFMA convolutions (C ABI shf_net_set_conv_mode): exact fp32; split-fp16 with three products
# (fp32-class: the mode every parity test runs); the reduced ladder with two / one product (drift-labelled)
CONV_MODES = {"fp32": 0, "f16x3": 1, "f16x2": 2, "f16": 3, "bf16": 4, 0: 0, 1: 1, 2: 2, 3: 3, 4: 4}


class Layer(object):
    """Base class of Python layers (kept for source compatibility: the proposal
    layer runs natively inside the runtime, nothing is called back)."""
    param_str = ""
    phase = TEST

    def setup(self, bottom, top):
        pass

    def reshape(self, bottom, top):
        pass

    def forward(self, bottom, top):
        pass

    def backward(self, top, propagate_down, bottom):
        pass


class Blob(object):
    def __init__(self, net, index, name):
        self._net = net
        self._i = index
        self.name = name

    @property
    def shape(self):
        dims = (C.c_int * 8)()
        n = self._net._lib.shf_blob_shape(self._net._h, self._i, dims)
        return tuple(int(dims[i]) for i in range(n))

    def reshape(self, *dims):
        arr = (C.c_int * len(dims))(*[int(d) for d in dims])
        _lib.check(self._net._lib.shf_blob_reshape(self._net._h, self._i, arr, len(dims)), "Blob.reshape")

    @property
    def data(self):
        """Writable fp32 view of the blob's host mirror (NCHW), like mutable_cpu_data()."""
        p = self._net._lib.shf_blob_mutable_host_data(self._net._h, self._i)
        if not p:
            raise _lib.ShfError(_lib.last_error())
        shape = self.shape
        n = int(np.prod(shape)) if len(shape) else 1
        if n == 0:
            return np.zeros(shape, dtype=np.float32)
        a = np.ctypeslib.as_array(p, shape=(n,)).reshape(shape)
        a.flags.writeable = True
        self._keep = self._net  # the view borrows from the net
        return a

    @property
    def count(self):
        return int(np.prod(self.shape))

    num = property(lambda self: self.shape[0])
    channels = property(lambda self: self.shape[1])
    height = property(lambda self: self.shape[2])
    width = property(lambda self: self.shape[3])


class _ParamBlob(object):
    """net.params[name][i]: writing through ``.data[...]`` and calling
    ``net.commit_params()`` (or the next forward) re-packs the weights on the GPU."""

    def __init__(self, net, layer, idx):
        self._net, self._layer, self._idx = net, layer, idx

    @property
    def shape(self):
        dims = (C.c_int * 4)()
        n = self._net._lib.shf_net_param_shape(self._net._h, self._layer, self._idx, dims)
        return tuple(int(dims[i]) for i in range(n))

    @property
    def data(self):
        p = self._net._lib.shf_net_param_data(self._net._h, self._layer, self._idx)
        shape = self.shape
        self._net._dirty_layers.add(self._layer)
        return np.ctypeslib.as_array(p, shape=(int(np.prod(shape)),)).reshape(shape)


class Net(object):
    def __init__(self, network_file, weights=None, phase=TEST, prototxt_text=None):
        if isinstance(weights, int) and phase == TEST and weights in (TRAIN, TEST):
            # caffe.Net(proto, phase) 2-arg form (_caffe.cpp:152-…)
            weights, phase = None, weights
        self._lib = _lib.load()
        enc = lambda s: None if s is None else str(s).encode()
        self._h = self._lib.shf_net_create(enc(network_file), enc(prototxt_text), enc(weights or ""), int(phase))
        if not self._h:
            raise RuntimeError(_lib.last_error())
        L = self._lib
        self._blob_names = [L.shf_net_blob_name(self._h, i).decode() for i in range(L.shf_net_num_blobs(self._h))]
        self._blobs = [Blob(self, i, n) for i, n in enumerate(self._blob_names)]
        self._inputs = [L.shf_net_input_blob(self._h, i) for i in range(L.shf_net_num_inputs(self._h))]
        self._outputs = [L.shf_net_output_blob(self._h, i) for i in range(L.shf_net_num_outputs(self._h))]
        self._layer_names = [L.shf_net_layer_name(self._h, i).decode() for i in range(L.shf_net_num_layers(self._h))]
        self._layer_types = [L.shf_net_layer_type(self._h, i).decode() for i in range(L.shf_net_num_layers(self._h))]
        self._dirty_layers = set()
        self._apply_cfg()

    def clone(self):
        """A lane: same parameter tensors, own activations / workspace / HIP stream
        (cf. Net::ShareTrainedLayersWith).  Keep ``self`` alive while the lane is used."""
        self.commit_params()
        h = self._lib.shf_net_clone(self._h)
        if not h:
            raise RuntimeError(_lib.last_error())
        lane = Net.__new__(Net)
        lane._lib, lane._h, lane._parent = self._lib, h, self
        for k in ("_blob_names", "_inputs", "_outputs", "_layer_names", "_layer_types"):
            setattr(lane, k, getattr(self, k))
        lane._blobs = [Blob(lane, i, n) for i, n in enumerate(lane._blob_names)]
        lane._dirty_layers = set()
        return lane

    def _apply_cfg(self):
        """The reference's Python layer reads cfg.TEST.* at EVERY forward (lib/layers/proposal_layer.py:88-92): called
        from forward() / detect_begin(), pushed to the runtime (shared by all lanes) only when a value changed."""
        from ..config import cfg
        cur = (int(cfg.TEST.N_DETS_PER_MODULE), float(cfg.TEST.SCORE_THRESH), float(cfg.TEST.ANCHOR_MIN_SIZE))
        root = getattr(self, "_parent", self)
        if getattr(root, "_cfg_applied", None) != cur:
            self.set_proposal_cfg(*cur)
            root._cfg_applied = cur

    def set_proposal_cfg(self, pre_nms_topN, score_thresh, min_size):
        getattr(self, "_parent", self)._cfg_applied = None   # an explicit override: re-read cfg at the next forward
        _lib.check(self._lib.shf_net_set_proposal_cfg(self._h, int(pre_nms_topN), float(score_thresh),
                                                      float(min_size)), "set_proposal_cfg")

    def set_conv_mode(self, mode):
        """Arithmetic of the MFMA convolutions: "fp32" (exact fp32 MFMA), "f16x3" (split-fp16, three fp16 products per
        fp32 product: fp32-class accuracy, the parity mode), and the reduced, drift-labelled modes "f16x2", "f16" (two /
        one fp16 product; fp16 range guard applies) and "bf16" (one bf16 product; fp32's exponent range, no guard)."""
        m = CONV_MODES[mode]
        self.commit_params()
        _lib.check(self._lib.shf_net_set_conv_mode(self._h, m), "set_conv_mode")

    @property
    def conv_mode(self):
        return {0: "fp32", 1: "f16x3", 2: "f16x2", 3: "f16", 4: "bf16"}[int(self._lib.shf_net_get_conv_mode(self._h))]

    def set_layer_products(self, table):
        """{layer name: 1 | 2 | 3 (0 clears)}: fp16 products per fp32 product for single layers of a split-fp16 mode
        (C ABI shf_net_set_layer_products)."""
        for name, n in dict(table).items():
            _lib.check(self._lib.shf_net_set_layer_products(self._h, str(name).encode(), int(n)), "set_layer_products")

    @property
    def range_fallbacks(self):
        """Forwards this net (and its lanes) re-ran on the exact fp32 kernels because a split-fp16 convolution left
        the fp16 range (C ABI shf_net_range_fallbacks)."""
        return int(self._lib.shf_net_range_fallbacks(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.shf_net_destroy(h)

    # -- pycaffe.py:24-85 ------------------------------------------------------
    @property
    def blobs(self):
        if not hasattr(self, "_blobs_dict"):
            self._blobs_dict = OrderedDict(zip(self._blob_names, self._blobs))
        return self._blobs_dict

    @property
    def params(self):
        if not hasattr(self, "_params_dict"):
            d = OrderedDict()
            for li, name in enumerate(self._layer_names):
                n = self._lib.shf_net_layer_num_params(self._h, li)
                if n > 0:
                    d[name] = [_ParamBlob(self, li, i) for i in range(n)]
            self._params_dict = d
        return self._params_dict

    @property
    def inputs(self):
        return [self._blob_names[i] for i in self._inputs]

    @property
    def outputs(self):
        return [self._blob_names[i] for i in self._outputs]

    def commit_params(self):
        for li in sorted(self._dirty_layers):
            _lib.check(self._lib.shf_net_param_commit(self._h, li), "param_commit")
        self._dirty_layers.clear()

    def _forward(self, start=0, end=None):
        self.commit_params()
        self._apply_cfg()
        _lib.check(self._lib.shf_net_forward(self._h), "Net.forward")

    def forward(self, blobs=None, start=None, end=None, **kwargs):
        """pycaffe.py:88-134 (whole-net forward only: start/end are not supported)."""
        if start is not None or end is not None:
            raise NotImplementedError("partial forward (start/end) is outside the inference hot path")
        if blobs is None:
            blobs = []
        outputs = set(self.outputs + blobs)
        tm = getattr(self, "timing", None)     # measurement only (bench.py net_forward_path): a dict collects host seconds
        t0 = time.perf_counter() if tm is not None else 0.0
        if kwargs:
            if set(kwargs.keys()) != set(self.inputs):
                raise Exception('Input blob arguments do not match net inputs.')
            for in_, blob in kwargs.items():
                if blob.shape[0] != self.blobs[in_].shape[0]:
                    raise Exception('Input is not batch sized')
                dst = self.blobs[in_].data
                # (a caller that filled the blob's own host mirror in place -- test.forward_net -- hands that view back)
                if not (isinstance(blob, np.ndarray) and blob.ctypes.data == dst.ctypes.data and blob.shape == dst.shape
                        and blob.strides == dst.strides):
                    dst[...] = blob
        t1 = time.perf_counter() if tm is not None else 0.0
        self._forward()
        t2 = time.perf_counter() if tm is not None else 0.0
        out = {out: self.blobs[out].data for out in outputs}
        if tm is not None:
            t3 = time.perf_counter()
            tm["calls"] = tm.get("calls", 0) + 1
            tm["input_copy_s"] = tm.get("input_copy_s", 0.0) + (t1 - t0)     # host blob -> the pinned mirror (Blob.data[...] = x)
            tm["forward_call_s"] = tm.get("forward_call_s", 0.0) + (t2 - t1)  # shf_net_forward: H2D + kernels + the count read-back
            tm["output_read_s"] = tm.get("output_read_s", 0.0) + (t3 - t2)   # Blob.data of the outputs: D2H
        return out

    # -- measurement helpers ------------------------------------------------------
    def sync(self):
        _lib.check(self._lib.shf_net_sync(self._h), "sync")

    def prof_enable(self, on=True):
        self._lib.shf_prof_enable(self._h, 1 if on else 0)

    def prof_only(self, class_name=None):
        """Bracket only launches of the named kernel class (None: every class) -- C ABI shf_prof_only."""
        cls = -1
        if class_name is not None:
            names = [self._lib.shf_prof_class_name(self._h, c).decode() for c in range(self._lib.shf_prof_num_classes(self._h))]
            cls = names.index(class_name)
        self._lib.shf_prof_only(self._h, cls)

    def prof_reset(self):
        _lib.check(self._lib.shf_prof_reset(self._h), "prof_reset")

    def prof_read(self):
        out = OrderedDict()
        for c in range(self._lib.shf_prof_num_classes(self._h)):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _lib.check(self._lib.shf_prof_read(self._h, c, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out[self._lib.shf_prof_class_name(self._h, c).decode()] = dict(
                launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    # -- fused per-image path (device-resident pyramid) ----------------------------
    def detect_begin(self):
        self.commit_params()
        self._apply_cfg()
        _lib.check(self._lib.shf_detect_begin(self._h), "detect_begin")

    def detect_add_level(self, data, H, W, im_h, im_w, im_scale, flip, thresh, on_device=False):
        """``data``: device pointer (int) when on_device else a C-contiguous fp32 (1,3,H,W) array."""
        if on_device:
            ptr = C.c_void_p(int(data))
        else:
            data = np.ascontiguousarray(data, dtype=np.float32)
            ptr = data.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.shf_detect_add_level(self._h, ptr, 1 if on_device else 0, int(H), int(W), int(im_h),
                                                  int(im_w), float(im_scale), 1 if flip else 0, float(thresh)),
                   "detect_add_level")

    def detect_add_levels(self, members, units, thresh, on_device=False, per_member_lists=False):
        """One grouped pass over several units (C ABI shf_detect_add_levels).  ``members``: distinct
        nets (self and/or lanes) lending their activation buffers, one per unit."""
        n = len(units)
        keep = []
        ptrs = (C.c_void_p * n)()
        for i, u in enumerate(units):
            if on_device:
                ptrs[i] = int(u[0])
            else:
                a = np.ascontiguousarray(u[0], dtype=np.float32)
                keep.append(a)
                ptrs[i] = a.ctypes.data
        mem = (C.c_void_p * n)(*[m._h for m in members[:n]])
        ia = lambda k: (C.c_int * n)(*[int(u[k]) for u in units])
        sc = (C.c_float * n)(*[float(u[5]) for u in units])
        fl = (C.c_int * n)(*[1 if u[6] else 0 for u in units])
        _lib.check(self._lib.shf_detect_add_levels(self._h, n, mem, ptrs, 1 if on_device else 0, ia(1), ia(2),
                                                   ia(3), ia(4), sc, fl, float(thresh),
                                                   1 if per_member_lists else 0), "detect_add_levels")

    def make_pyramid_level(self, im_dev, im_h, im_w, scale, flip, pixel_means, out_dev, H, W, lvl_h, lvl_w):
        """_get_image_blob + flip + pad for one unit on this net's stream (C ABI shf_make_pyramid_level):
        ``im_dev`` raw BGR uint8 HxWx3 device pointer -> ``out_dev`` (1,3,H,W) fp32 device pointer."""
        pm = (C.c_double * 3)(*[float(v) for v in np.asarray(pixel_means).reshape(-1)[:3]])
        _lib.check(self._lib.shf_make_pyramid_level(self._h, int(im_dev), int(im_h), int(im_w), float(scale),
                                                    1 if flip else 0, pm, int(out_dev), int(H), int(W), int(lvl_h),
                                                    int(lvl_w)), "make_pyramid_level")

    def set_predecessor(self, prev):
        """Pipeline hand-over at logits granularity (C ABI shf_net_set_predecessor)."""
        _lib.check(self._lib.shf_net_set_predecessor(self._h, prev._h if prev is not None else None), "set_predecessor")

    def set_pipeline(self, enable=True):
        """Shared in-order conv stream + high-priority own stream for this head (C ABI shf_net_set_pipeline)."""
        _lib.check(self._lib.shf_net_set_pipeline(self._h, 1 if enable else 0), "set_pipeline")

    def record_event(self):
        _lib.check(self._lib.shf_net_record_event(self._h), "record_event")

    def wait_event(self, other):
        _lib.check(self._lib.shf_net_wait_event(self._h, other._h), "wait_event")

    def detect_count(self):
        n = self._lib.shf_detect_count(self._h)
        if n < 0:
            raise _lib.ShfError(_lib.last_error())
        return n

    def detect_export(self, dst_ptr, cap_rows):
        """Copy this image's rows to a device buffer (e.g. a torch tensor's data_ptr()); returns the row count."""
        n = C.c_int(0)
        _lib.check(self._lib.shf_detect_export(self._h, C.c_void_p(int(dst_ptr)), int(cap_rows), C.byref(n)),
                   "detect_export")
        return n.value

    def detect_export_many(self, members, dst_ptrs, cap_rows):
        """Row counts of the members' lists after a per_member_lists pass; rows land in dst_ptrs[m]."""
        n = len(members)
        mem = (C.c_void_p * n)(*[m._h for m in members])
        dst = (C.c_void_p * n)(*[int(p) for p in dst_ptrs])
        cnt = (C.c_int * n)()
        _lib.check(self._lib.shf_detect_export_many(self._h, n, mem, dst, int(cap_rows), cnt), "detect_export_many")
        return [int(cnt[i]) for i in range(n)]

    def detect_import(self, src_ptr, n_rows):
        _lib.check(self._lib.shf_detect_import(self._h, C.c_void_p(int(src_ptr)), int(n_rows)), "detect_import")

    # -- diagnostics (tests) -----------------------------------------------------------
    def debug_proposal(self, scores, deltas, im_info):
        """ProposalLayer.forward on injected blobs through the HIP tail (C ABI shf_debug_proposal):
        scores (1,2A,h,w), deltas (1,4A,h,w), im_info (1,3) -> (boxes (max(R,1),5), probs (R,2), overflow flag)."""
        sc = np.ascontiguousarray(scores, dtype=np.float32)
        dl = np.ascontiguousarray(deltas, dtype=np.float32)
        ii = np.ascontiguousarray(im_info, dtype=np.float32).reshape(-1)
        h, w = sc.shape[2:]
        A = sc.shape[1] // 2
        assert dl.shape == (1, 4 * A, h, w), dl.shape
        cap = h * w * A
        boxes = np.zeros((cap, 5), np.float32)
        probs = np.zeros((cap, 2), np.float32)
        n, of = C.c_int(0), C.c_int(0)
        F = C.POINTER(C.c_float)
        _lib.check(self._lib.shf_debug_proposal(self._h, sc.ctypes.data_as(F), dl.ctypes.data_as(F), h, w,
                                                ii.ctypes.data_as(F), boxes.ctypes.data_as(F), probs.ctypes.data_as(F),
                                                cap, C.byref(n), C.byref(of)), "debug_proposal")
        return boxes[:max(n.value, 1)].copy(), probs[:n.value].copy(), bool(of.value)

    def debug_append(self, boxes5, probs2, im_w, im_scale, flip, thresh):
        """forward_net's flip fix / unscale + the > thresh cut on injected proposals (C ABI shf_debug_append)."""
        b = np.ascontiguousarray(boxes5, dtype=np.float32).reshape(-1, 5)
        p = np.ascontiguousarray(probs2, dtype=np.float32).reshape(-1, 2)
        assert len(b) == len(p)
        F = C.POINTER(C.c_float)
        _lib.check(self._lib.shf_debug_append(self._h, b.ctypes.data_as(F), p.ctypes.data_as(F), len(b), int(im_w),
                                              float(im_scale), 1 if flip else 0, float(thresh)), "debug_append")

    def detect_finish(self, method="BBOX_VOTE", nms_thresh=0.4, cap=None):
        m = {"BBOX_VOTE": 0, "NMS": 1}[method]
        cap = cap or 4096
        while True:
            out = np.empty((cap, 5), dtype=np.float64)
            n = C.c_int(0)
            _lib.check(self._lib.shf_detect_finish(self._h, m, float(nms_thresh),
                                                   out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)),
                       "detect_finish")
            if n.value <= cap:
                return out[:n.value]
            cap = n.value  # rare: more merged boxes than expected -> the merge is re-run


def pyramid_level_shape(im_h, im_w, scale, max_resolution):
    """(lvl_h, lvl_w, H, W) of one pyramid unit (C ABI shf_pyramid_level_shape; needs no GPU)."""
    lib = _lib.load(require_gpu=False)
    o = [C.c_int() for _ in range(4)]
    if lib.shf_pyramid_level_shape(int(im_h), int(im_w), float(scale), int(max_resolution), *[C.byref(v) for v in o]):
        raise ValueError("pyramid_level_shape: bad geometry %r" % ((im_h, im_w, scale, max_resolution),))
    return tuple(v.value for v in o)


def alloc_counts():
    """(device, pinned host) (re)allocations the runtime's grow-only buffers have made in this process so far
    (C ABI shf_alloc_counts): unchanged by a stream of images whose shapes were all seen before."""
    lib = _lib.load(require_gpu=False)
    d, h = C.c_longlong(0), C.c_longlong(0)
    lib.shf_alloc_counts(C.byref(d), C.byref(h))
    return int(d.value), int(h.value)
