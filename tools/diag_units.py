"""Unit 2 of the knob-test image, several times: is the fused path deterministic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smallhardface_amd.config import cfg
from smallhardface_amd import test as T
from tests import helpers as H
cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
cfg.TEST.SCALES = [100, 300, 500]
gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
gnet.set_conv_mode("f16x3")
im = np.random.default_rng(5).integers(0, 256, (150, 200, 3)).astype(np.uint8)
units = list(T.pyramid_units(im))
res = []
for rep in range(4):
    d = np.asarray(T.detect_fused(gnet, [units[2]], thresh=0.05)[0])
    res.append(d)
    print(rep, len(d), "equal to first:", d.shape == res[0].shape and bool(np.array_equal(d, res[0])))
# other units in between (different lanes state)
d4 = T.detect_fused(gnet, [units[4]], thresh=0.05)[0]
d = np.asarray(T.detect_fused(gnet, [units[2]], thresh=0.05)[0])
print("after unit 4:", len(d), d.shape == res[0].shape and bool(np.array_equal(d, res[0])))
np.save(sys.argv[1], res[0])
