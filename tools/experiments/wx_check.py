"""Winograd-x experiment (SHF_F16X3_WX=1): parity of single layers against the oracle + per-layer timing.
    SHF_F16X3_WX=1 python tools/wx_check.py parity
    [SHF_F16X3_WX=1] python tools/wx_check.py time [reps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smallhardface_amd import caffe, prototxt as P, weights
from tests.test_gpu_parity import conv_layer
from tests import helpers as H


def parity():
    worst = 0.0
    for (cin, cout, h, w, relu, scale) in [(256, 128, 16, 16, True, 1.0), (256, 256, 35, 41, True, 1.0), (512, 512, 16, 24, False, 1.0),
                                           (272, 384, 19, 33, True, 1.0), (256, 128, 9, 70, True, 2.0 ** -12), (512, 128, 50, 17, True, 64.0)]:
        txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, 3, 1, 1, relu) +
                                 conv_layer("c2", "c1", 64, 1, 0), 3, h, w)
        gnet, onet = H.make_pair(P.parse(txt), seed=5)
        gnet.set_conv_mode("f16x3")
        rng = np.random.default_rng(3)
        sc = np.float32(scale)
        for name in ("c0", "c1", "c2"):
            onet.params[name][1][...] = (rng.normal(0, 0.5, onet.params[name][1].shape) * sc).astype(np.float32)
        H.load_params(gnet, onet.params)
        data = (rng.normal(0, 1, (1, 3, h, w)) * sc).astype(np.float32)
        go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
        gnet.prof_enable(True); gnet.prof_reset(); gnet._forward(); pr = gnet.prof_read(); gnet.prof_enable(False)
        used = [k for k, v in pr.items() if v["launches"] and k.startswith("conv_mfma")]
        e1, e2 = H.rel_err(gnet.blobs["c1"].data, onet.blobs["c1"].data), H.rel_err(go["c2"], oo["c2"])
        worst = max(worst, e1, e2)
        print("cin %d cout %d %dx%d relu %d scale %g: rel err c1 %.2e c2 %.2e  kernels %s" % (cin, cout, h, w, relu, scale, e1, e2, used))
    print("WORST %.3e (bar 2e-5)" % worst)


def timing(reps):
    shapes = [("conv3_2", 256, 256, 352, 352), ("conv4_1", 256, 512, 176, 176), ("conv4_2", 512, 512, 176, 176),
              ("conv5_2", 512, 512, 88, 88), ("dim_red", 512, 128, 176, 176)]
    for name, cin, cout, h, w in shapes:
        txt = H.single_layer_net(conv_layer("c0", "data", cin, 3, 1) + conv_layer("c1", "c0", cout, 3, 1), 3, h, w)
        msg = P.parse(txt)
        net = caffe.Net(None, prototxt_text=P.dumps(msg))
        H.load_params(net, weights.synth_params(msg, seed=1))
        net.set_conv_mode("f16x3")
        net.blobs['data'].reshape(1, 3, h, w)
        net.blobs['im_info'].reshape(1, 3)
        data = np.random.default_rng(0).normal(0, 1, (1, 3, h, w)).astype(np.float32)
        net.forward(data=data, im_info=np.zeros((1, 3), np.float32))
        net.prof_enable(True); net.prof_reset()
        for _ in range(reps):
            net._forward()
        pr = net.prof_read(); net.prof_enable(False)
        tot = sum(v["ms"] for k, v in pr.items() if k.startswith("conv_mfma") and v["launches"]) / reps
        ks = [k for k, v in pr.items() if k.startswith("conv_mfma") and v["launches"]]
        print("%-8s %4d->%-4d %dx%d  %8.1f us  %s" % (name, cin, cout, h, w, 1e3 * tot, ks))
        del net


if __name__ == "__main__":
    if sys.argv[1] == "parity":
        parity()
    else:
        timing(int(sys.argv[2]) if len(sys.argv) > 2 else 10)
