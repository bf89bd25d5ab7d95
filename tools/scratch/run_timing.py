import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from smallhardface_amd import caffe, prototxt as P, weights, test as T
from smallhardface_amd.config import cfg, cfg_from_file
cfg_from_file("configs/smallhardface.toml")
caffe.set_mode_gpu(); caffe.set_device(0)
msg = P._add_dimension_reduction(P.build_test_template(True))
net = caffe.Net(None, prototxt_text=P.dumps(msg))
params = weights.synth_params(msg, seed=1234)
for name, blobs in params.items():
    for i, arr in enumerate(blobs):
        net.params[name][i].data[...] = arr
net.commit_params(); net.set_conv_mode("f16x3")
im = np.random.default_rng(1000).integers(0, 256, (1024, 1024, 3)).astype(np.uint8)
units = list(T.pyramid_units(im))
fd = T.FusedDetector(net, n_lanes=10, mode="group")
for _ in range(2):
    fd.detect(units, thresh=0.05)
