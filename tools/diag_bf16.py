import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smallhardface_amd import prototxt as P
from tests import helpers as H
from tests.test_gpu_parity import conv_layer
h, w = 16, 16
txt = H.single_layer_net(conv_layer("c0", "data", 64, 3, 1) + conv_layer("c1", "c0", 64, 1, 0, relu=False), 3, h, w)
gnet, onet = H.make_pair(P.parse(txt), seed=11)
W = np.zeros((64, 64, 1, 1), np.float32)
for o in range(64):
    W[o, o, 0, 0] = 1.0
onet.params["c1"][0][...] = W
onet.params["c1"][1][...] = 0
H.load_params(gnet, onet.params)
data = np.random.default_rng(2).normal(0, 1, (1, 3, h, w)).astype(np.float32)
for mode in ("f16", "bf16"):
    gnet.set_conv_mode(mode)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    a, b = go["c1"][0, :, 3, 5], oo["c1"][0, :, 3, 5]
    print(mode, "rel", H.rel_err(go["c1"], oo["c1"]))
    print(" got ", np.round(a[:12], 4))
    print(" want", np.round(b[:12], 4))
    print(" ratio", np.round(a[:12] / np.where(b[:12] == 0, 1, b[:12]), 3))
