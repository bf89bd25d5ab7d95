"""Diagnostic: how far are the split-fp16 kernel families from the exact fp32 mode on one small pyramid?
usage: python tools/diag_arith.py   (env knobs select the kernels: SHF_F16X3_W4=0 -> 8-wave two-accumulator everywhere)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smallhardface_amd.config import cfg
from smallhardface_amd import test as T
from tests import helpers as H

cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
cfg.TEST.SCALES = [100, 300, 500]
gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
im = np.random.default_rng(5).integers(0, 256, (150, 200, 3)).astype(np.uint8)
units = list(T.pyramid_units(im))
out = {}
for mode in ("fp32", "f16x3"):
    gnet.set_conv_mode(mode)
    scores = []
    for (data, Hh, Ww, im_h, im_w, s, flip) in units:
        gnet.blobs['data'].reshape(*data.shape)
        gnet.blobs['im_info'].reshape(1, 3)
        gnet.forward(data=data, im_info=np.array([[im_h, im_w, s]], np.float32))
        scores.append(gnet.blobs["cls_prob_reshape_output"].data.copy())
    fd = T.FusedDetector(gnet, n_lanes=6, mode="group")
    dets = fd.detect(units, thresh=0.05)[0]
    out[mode] = (scores, dets)
d = max(float(np.abs(a - b).max()) for a, b in zip(out["fp32"][0], out["f16x3"][0]))
print("max |dscore| f16x3 vs fp32 over all anchors (Net.forward path): %.3e" % d)
a, b = out["fp32"][1], out["f16x3"][1]
print("voted boxes fp32 %d, f16x3 (fused) %d" % (len(a), len(b)))
n = min(len(a), len(b))
print("rank-matched max |dscore| %.3e" % float(np.abs(a[:n, 4] - b[:n, 4]).max()))
if len(sys.argv) > 1:
    np.save(sys.argv[1], b)
