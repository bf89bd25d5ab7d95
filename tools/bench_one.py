"""Run ONE conv shape a few times (for rocprofv3 --pmc).  python3 tests/bench_one.py cin cout k dil h w [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smallhardface_amd import caffe, prototxt as P, weights
from tests.test_gpu_parity import conv_layer
from tests import helpers as H
cin, cout, k, dil, h, w = [int(x) for x in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
pad = dil if k == 3 else 0
msg = P.parse(H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, k, pad, dil), 3, h, w))
net = caffe.Net(None, prototxt_text=P.dumps(msg))
H.load_params(net, weights.synth_params(msg, seed=1))
net.blobs['data'].reshape(1, 3, h, w); net.blobs['im_info'].reshape(1, 3)
net.forward(data=np.random.default_rng(0).normal(0, 1, (1, 3, h, w)).astype(np.float32), im_info=np.zeros((1, 3), np.float32))
for _ in range(reps):
    net._forward()
