"""Minimal ``.caffemodel`` (binary NetParameter) writer + probe.

The reader the runtime uses lives in csrc/proto_text.h (``read_caffemodel``; it follows
Net::CopyTrainedLayersFrom, caffe/src/caffe/net.cpp:733-768, and caffe.proto's
``NetParameter.layer = 100``, ``LayerParameter{name=1,type=2,blobs=7}``,
``BlobProto{shape=7{dim=1 packed}, data=5 packed float}``).  The writer exists so tests and
benchmarks can materialise seeded synthetic weights as a real model file (no protoc in the
image, no trained model in the reference tree).
"""
import ctypes as C
import struct

import numpy as np

from . import _lib


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(no, wire, payload):
    if wire == 2:
        return _varint((no << 3) | 2) + _varint(len(payload)) + payload
    return _varint((no << 3) | wire) + payload


def _blob(arr):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    shape = _field(1, 2, b"".join(_varint(d) for d in arr.shape))  # BlobShape.dim (packed int64)
    return _field(7, 2, shape) + _field(5, 2, arr.tobytes())       # BlobProto.shape, .data (packed float)


def write_caffemodel(path, layers, net_name="face", layer_types=None):
    """``layers``: {layer_name: [ndarray, ...]} (Caffe blob shapes), written in dict order."""
    layer_types = layer_types or {}
    body = _field(1, 2, net_name.encode())
    for name, blobs in layers.items():
        lp = _field(1, 2, name.encode()) + _field(2, 2, layer_types.get(name, "Convolution").encode())
        for b in blobs:
            lp += _field(7, 2, _blob(b))
        body += _field(100, 2, lp)
    with open(path, "wb") as f:
        f.write(body)
    return path


def read_blob(path, layer, index):
    """Read one parameter blob back through the runtime's own reader (no GPU needed)."""
    lib = _lib.load(require_gpu=False)
    dims = (C.c_int * 8)()
    nd = C.c_int(0)
    n = lib.shf_caffemodel_read_blob(str(path).encode(), layer.encode(), int(index), None, 0, dims, C.byref(nd))
    if n < 0:
        raise _lib.ShfError(_lib.last_error())
    out = np.empty(n, dtype=np.float32)
    n2 = lib.shf_caffemodel_read_blob(str(path).encode(), layer.encode(), int(index),
                                      out.ctypes.data_as(C.POINTER(C.c_float)), n, dims, C.byref(nd))
    if n2 < 0:
        raise _lib.ShfError(_lib.last_error())
    return out.reshape([dims[i] for i in range(nd.value)])
