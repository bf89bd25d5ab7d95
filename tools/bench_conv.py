"""Per-layer micro-benchmark of the MFMA conv kernel (HIP events on the runtime's stream).

    python -m tools.bench_conv [reps]

Shapes are the detector's layers at the 1408x1408 pyramid level (the unit that carries
57 % of an image's FLOPs) plus a grouped launch over all ten units of the bench image.
"""
import sys

import numpy as np

from smallhardface_amd import caffe, prototxt as P, weights
from tests.test_gpu_parity import conv_layer
from tests import helpers as H

SHAPES = [  # name, cin, cout, k, dil, h, w
    ("conv1_2", 64, 64, 3, 1, 1408, 1408),
    ("conv2_1", 64, 128, 3, 1, 704, 704),
    ("conv2_2", 128, 128, 3, 1, 704, 704),
    ("conv3_2", 256, 256, 3, 1, 352, 352),
    ("conv4_2", 512, 512, 3, 1, 176, 176),
    ("conv5_2", 512, 512, 3, 1, 88, 88),
    ("dim_red", 512, 128, 3, 1, 176, 176),
    ("head_4", 128, 128, 3, 4, 176, 176),
    ("conv4_256", 512, 256, 1, 1, 176, 176),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    for name, cin, cout, k, dil, h, w in SHAPES:
        if only and name not in only:
            continue
        pad = dil if k == 3 else 0
        txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, k, pad, dil), 3, h, w)
        msg = P.parse(txt)
        net = caffe.Net(None, prototxt_text=P.dumps(msg))
        H.load_params(net, weights.synth_params(msg, seed=1))
        net.blobs['data'].reshape(1, 3, h, w)
        net.blobs['im_info'].reshape(1, 3)
        data = np.random.default_rng(0).normal(0, 1, (1, 3, h, w)).astype(np.float32)
        net.forward(data=data, im_info=np.zeros((1, 3), np.float32))
        net.prof_enable(True)
        net.prof_reset()
        for _ in range(reps):
            net._forward()
        pr = net.prof_read()
        net.prof_enable(False)
        for cls, v in pr.items():
            if cls.startswith("conv_mfma") and v["launches"]:
                print("%-10s %-28s %4d->%-4d k%d d%d %4dx%-4d  %8.1f us  %6.1f TF/s" % (
                    name, cls, cin, cout, k, dil, h, w, 1e3 * v["ms"] / v["launches"],
                    v["flops"] / (v["ms"] * 1e-3) / 1e12))
        del net


if __name__ == "__main__":
    main()
