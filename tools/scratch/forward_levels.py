"""Per-level device time of ONE Net.forward() (the literal drop-in path's unit of work) by kernel class, against the level's
FLOP share of the grouped image pass: where do ten separate forwards lose against one grouped pass?"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np
from smallhardface_amd import caffe, prototxt as P, weights, pyramid
from smallhardface_amd.config import cfg, cfg_from_file
cfg_from_file("configs/smallhardface.toml")
caffe.set_mode_gpu(); caffe.set_device(0)
msg = P._add_dimension_reduction(P.build_test_template(True))
net = caffe.Net(None, prototxt_text=P.dumps(msg))
for name, blobs in weights.synth_params(msg, seed=1234).items():
    for i, arr in enumerate(blobs):
        net.params[name][i].data[...] = arr
net.commit_params(); net.set_conv_mode("f16x3")
rng = np.random.default_rng(0)
out = {}
for side in (1408, 1008, 608, 304, 112):
    data = (rng.integers(0, 256, (1, 3, side, side)).astype(np.float32) - 115.0)
    net.blobs['data'].reshape(*data.shape); net.blobs['im_info'].reshape(1, 3)
    info = np.array([[side, side, side / 1024.0]], np.float32)
    for _ in range(2):
        net.forward(data=data, im_info=info)
    net.prof_enable(True); net.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        net.forward(data=data, im_info=info)
    wall = (time.perf_counter() - t0) / 3
    pr = net.prof_read(); net.prof_enable(False)
    dev = {k: round(v["ms"] / 3, 3) for k, v in pr.items() if v["launches"]}
    tot = sum(v for k, v in dev.items() if k not in ("h2d_copy", "d2h_copy"))
    gf = pyramid.level_flops(side, side) / 1e9
    out[side] = {"wall_ms": round(1000 * wall, 2), "kernels_ms": round(tot, 3), "gflop": round(gf, 1),
                 "tflops": round(gf / tot, 1), "by_class": dev}
    print(side, json.dumps(out[side]))
