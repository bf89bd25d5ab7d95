"""Prints, per pyramid unit of the bench image, how many proposals / >0.05 detections the
synthetic weights give (used to pick the synthetic cls bias)."""
import sys
import numpy as np
from smallhardface_amd import caffe, prototxt as P, weights
from smallhardface_amd.config import cfg, cfg_from_file
import bench

cfg_from_file("configs/smallhardface.toml")
msg = P._add_dimension_reduction(P.build_test_template(True))
for bias in [float(a) for a in sys.argv[1:]] or [6.0]:
    params = weights.synth_params(msg, seed=1234, cls_bias=bias)
    net = caffe.Net(None, prototxt_text=P.dumps(msg))
    for name, blobs in params.items():
        for i, arr in enumerate(blobs):
            net.params[name][i].data[...] = arr
    net.commit_params()
    units = bench.build_units(0)
    tot = 0
    for (data, H, W, im_h, im_w, s, flip) in units[::2]:
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(1, 3)
        out = net.forward(data=data, im_info=np.array([[im_h, im_w, s]], np.float32))
        p = out['cls_prob'][:, 1]
        lg = net.blobs['cls_prob_reshape_output'].data[0, 3:]
        tot += int((p > 0.05).sum())
        print("bias %.1f level %4d: anchors %6d R %5d >0.05 %5d  fg mean %.4f max %.4f" % (
            bias, H, lg.size, len(p), (p > 0.05).sum(), lg.mean(), lg.max()))
    print("bias %.1f total >0.05 (x2 flips): %d" % (bias, 2 * tot))
