"""Box merging entry points of the driver, backed by the HIP runtime.

  nms(dets, thresh)      lib/nms/nms_wrapper.py:13-21 -> gpu_nms (lib/nms/gpu_nms.pyx:16-31)
  bbox_vote(det)         lib/test.py:181-217
  generate_anchors(...)  lib/layers/generate_anchors.py:11-24

Both box ops take the UNSORTED (N,5) float32 array the reference passes and do the
score sort on the device; ties are broken by the lower input index (the reference's
``argsort()[::-1]`` leaves tie order implementation-defined).
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import cfg


def nms(dets, thresh, force_cpu=False):
    """Greedy NMS, suppression when IoU > thresh (the canonical GPU predicate,
    lib/nms/nms_kernel.cu:82).  Returns a list of indices into ``dets``."""
    if dets.shape[0] == 0:
        return []
    if force_cpu or not cfg.USE_GPU_NMS:
        raise _lib.ShfError("the Cython CPU NMS path (lib/nms/cpu_nms.pyx) is not part of this runtime; "
                            "set USE_GPU_NMS = true")
    lib = _lib.load()
    d = np.ascontiguousarray(dets[:, :5], dtype=np.float32)
    keep = np.zeros(d.shape[0], dtype=np.int32)
    n = C.c_int(0)
    dev = cfg.get("GPU_ID", -1)
    dev = dev if isinstance(dev, int) else -1
    _lib.check(lib.shf_nms(d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], float(thresh), int(dev),
                           keep.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)), "nms")
    return list(keep[:n.value].astype(np.int64))


def bbox_vote(det, thresh=None):
    """bbox_vote from PyramidBox as the reference implements it; returns (M,5) float64."""
    lib = _lib.load()
    thr = float(cfg.TEST.NMS_THRESH if thresh is None else thresh)
    d = np.ascontiguousarray(np.asarray(det)[:, :5], dtype=np.float32).reshape(-1, 5)
    cap = max(int(d.shape[0]), 1)
    out = np.empty((cap, 5), dtype=np.float64)
    n = C.c_int(0)
    _lib.check(lib.shf_bbox_vote(d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[0], thr,
                                 out.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)), "bbox_vote")
    return out[:n.value].copy()


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6), shifts=np.array([0]),
                     strides=np.array([0])):
    """Base anchors (float64), computed by the same native routine the net uses."""
    lib = _lib.load(require_gpu=False)
    r = np.ascontiguousarray(ratios, dtype=np.float64)
    s = np.ascontiguousarray(scales, dtype=np.float64)
    sh = np.ascontiguousarray(shifts, dtype=np.float64)
    st = np.ascontiguousarray(strides, dtype=np.float64)
    nsc = min(len(s), len(st))  # zip(scales, strides)
    rows = len(r) * nsc * len(sh) ** 2
    out = np.empty((rows, 4), dtype=np.float64)
    P = C.POINTER(C.c_double)
    n = lib.shf_generate_anchors(int(base_size), r.ctypes.data_as(P), len(r), s.ctypes.data_as(P), nsc,
                                 sh.ctypes.data_as(P), len(sh), st.ctypes.data_as(P), out.ctypes.data_as(P), rows)
    if n < 0:
        raise _lib.ShfError(_lib.last_error())
    return out[:n]
