"""The plain-C restatement (oracle/caffe_ops.c) against the numpy oracle and the golden
vectors -- two independent restatements of the same reference code must agree."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oc():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_c.so"))
    lib.oc_iou.restype = C.c_float
    return lib


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.mark.parametrize("k,pad,dil,stride", [(3, 1, 1, 1), (3, 2, 2, 1), (3, 4, 4, 1), (1, 0, 1, 1), (3, 0, 1, 2)])
def test_conv(oc, k, pad, dil, stride):
    rng = np.random.default_rng(k + pad)
    x = rng.normal(size=(1, 5, 9, 11)).astype(np.float32)
    w = rng.normal(size=(4, 5, k, k)).astype(np.float32)
    b = rng.normal(size=4).astype(np.float32)
    ref = O.convolution(x, w, b, pad, stride, dil)
    y = np.zeros(ref.shape[1:], np.float32)
    assert oc.oc_conv(fp(x), 5, 9, 11, fp(w), fp(b), 4, k, pad, stride, dil, fp(y)) == 0
    np.testing.assert_allclose(y, ref[0], atol=1e-4)


def test_maxpool(oc):
    row = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    y = np.zeros((1, 2, 4), np.float32)
    ho, wo = C.c_int(), C.c_int()
    oc.oc_maxpool(fp(row), 1, 3, 5, 2, 1, 0, fp(y), C.byref(ho), C.byref(wo))
    assert (ho.value, wo.value) == (2, 4)
    np.testing.assert_array_equal(y[0], [[9, 5, 5, 8], [9, 5, 5, 8]])   # test_pooling_layer.cpp:49-119
    x = np.random.default_rng(0).normal(size=(3, 7, 9)).astype(np.float32)
    y = np.zeros((3, 4, 5), np.float32)
    oc.oc_maxpool(fp(x), 3, 7, 9, 2, 2, 0, fp(y), C.byref(ho), C.byref(wo))
    np.testing.assert_array_equal(y, O.max_pool(x[None], 2, 2, 0)[0])


@pytest.mark.parametrize("name", ["clusters_small", "clusters_mid", "clusters_big", "dense", "iou_exact_0p4", "single"])
@pytest.mark.parametrize("thr", [0.4, 0.7])
def test_nms_bitmask_vs_golden(oc, golden, name, thr):
    g = golden("vote_nms.npz")
    d = g[name + "_dets"]
    order = g[name + "_order"]
    srt = np.ascontiguousarray(d[order], dtype=np.float32)
    keep = np.zeros(len(d), np.int32)
    n = oc.oc_nms_bitmask(fp(srt), len(d), C.c_float(thr), keep.ctypes.data_as(C.POINTER(C.c_int)))
    # gpu_nms.pyx:31 -> order[keep]; py_cpu_nms (golden) has the same '>' semantics
    np.testing.assert_array_equal(order[keep[:n]], g[name + "_nms_%02d" % int(thr * 100)])


def test_cpu_nms_ge_variant(oc, golden):
    g = golden("vote_nms.npz")
    for name in ("iou_exact_0p4", "clusters_mid"):
        d = np.ascontiguousarray(g[name + "_dets"], dtype=np.float32)
        order = np.ascontiguousarray(O.canonical_order(d[:, 4]), dtype=np.int64)
        keep = np.zeros(len(d), np.int32)
        n = oc.oc_cpu_nms(fp(d), order.ctypes.data_as(C.POINTER(C.c_longlong)), len(d), C.c_float(0.4),
                          keep.ctypes.data_as(C.POINTER(C.c_int)))
        np.testing.assert_array_equal(keep[:n], O.nms_ge(d, 0.4))
    # the '>=' predicate suppresses the IoU == thr pairs that '>' keeps
    assert len(O.nms_ge(g["iou_exact_0p4_dets"], 0.4)) == 2 and len(O.nms(g["iou_exact_0p4_dets"], 0.4)) == 4


def test_iou(oc):
    a = np.array([0, 0, 9, 9], np.float32)
    b = np.array([0, 0, 9, 3], np.float32)
    assert oc.oc_iou(fp(a), fp(b)) == np.float32(0.4)
    assert O.iou_row(a, b[None])[0] == np.float32(0.4)
