"""Fraction of exactly-zero activations feeding each convolution of the bench workload's net (synthetic weights, one
pyramid level) -- the operand statistics tools/mfma_power.hip / shf_calib_matrix_pipe should be read against.

    python tools/diag_zero_fraction.py [level_index]      (GPU)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import build_units  # noqa: E402
from smallhardface_amd import caffe, prototxt as P, weights  # noqa: E402
from smallhardface_amd.config import cfg_from_file  # noqa: E402

cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
caffe.set_mode_gpu()
caffe.set_device(0)
msg = P._add_dimension_reduction(P.build_test_template(True))
params = weights.synth_params(msg, seed=1234)
net = caffe.Net(None, prototxt_text=P.dumps(msg))
for name, blobs in params.items():
    for i, arr in enumerate(blobs):
        net.params[name][i].data[...] = arr
net.commit_params()
net.set_conv_mode("f16x3")
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data, H, W, im_h, im_w, s, flip = build_units(0)[2 * lvl]
net.blobs["data"].reshape(*data.shape)
net.blobs["data"].data[...] = data
net.blobs["im_info"].reshape(1, 3)
net.blobs["im_info"].data[...] = np.array([[im_h, im_w, s]], np.float32)
net.forward()
out = {}
for L in msg.getall("layer"):
    if L.get("type") != "Convolution":
        continue
    b = L.getall("bottom")[0]
    a = np.asarray(net.blobs[b].data)
    out[L.get("name")] = {"input": b, "zero_fraction": round(float((a == 0).mean()), 4), "cin": int(a.shape[1])}
w = {k: v for k, v in out.items() if v["cin"] >= 64}
print(json.dumps({"level": "%dx%d" % (H, W), "layers": out,
                  "mean_zero_fraction_cin_ge_64": round(float(np.mean([v["zero_fraction"] for v in w.values()])), 4)}, indent=1))
