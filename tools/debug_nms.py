import ctypes as C
import numpy as np
from oracle import oracle as O
from smallhardface_amd import _lib

lib = _lib.load()
g = np.load("tests/golden/vote_nms.npz")
for name in ["clusters_small", "clusters_big"]:
    d = np.ascontiguousarray(g[name + "_dets"], dtype=np.float32)
    n = len(d); nw = (n + 63) // 64
    mask = np.zeros((n, nw), np.uint64); cl = np.zeros(n, np.int32); heads = np.zeros(n, np.int32)
    srt = np.zeros((n, 5), np.float32); perm = np.zeros(n, np.int32); nh = C.c_int(0)
    lib.shf_debug_merge.restype = C.c_int
    rc = lib.shf_debug_merge(d.ctypes.data_as(C.c_void_p), n, C.c_float(0.4), 0, mask.ctypes.data_as(C.c_void_p),
                             cl.ctypes.data_as(C.c_void_p), heads.ctypes.data_as(C.c_void_p), C.byref(nh),
                             srt.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p))
    print(name, "rc", rc, "n", n, "nheads", nh.value)
    order = O.canonical_order(d[:, 4])
    print(" perm ok", np.array_equal(perm, order), " sorted ok", np.array_equal(srt, d[order]))
    # expected mask
    bad = 0
    for i in range(min(n, 200)):
        iou = O.iou_row(srt[i], srt)
        exp = (iou > np.float32(0.4))
        exp[: i + 1] = False
        got = np.zeros(n, bool)
        for w in range(i // 64, nw):
            word = int(mask[i, w])
            for b in range(64):
                if (word >> b) & 1 and w * 64 + b < n:
                    got[w * 64 + b] = True
        if not np.array_equal(exp, got):
            bad += 1
            if bad < 4:
                print("  row", i, "exp", np.where(exp)[0][:10], "got", np.where(got)[0][:10])
    print(" mask rows bad (first 200):", bad)
    print(" heads[:40]", heads[:40])
    ref_heads = []
    rem = np.zeros(n, bool)
    for i in range(n):
        if rem[i]:
            continue
        ref_heads.append(i)
        iou = O.iou_row(srt[i], srt)
        m = iou > np.float32(0.4); m[: i + 1] = False
        rem |= m
    print(" ref  [:40]", np.array(ref_heads[:40]))
