"""fp16 range guard of the split-fp16 conv mode (ADVICE r1 / VERDICT r1 #6).  The reference computes in fp32
everywhere (caffe/python/caffe/_caffe.cpp:46-48); hi = fp16(x) of the split overflows above 65504.  Whatever the
magnitudes, the split-fp16 path must give the right answer (by re-running on the exact fp32 kernels) or a clean
error -- never a silent inf / NaN."""
import numpy as np
import pytest

from smallhardface_amd import prototxt as P
from smallhardface_amd.config import cfg
from tests import helpers as H
from tests.test_gpu_parity import conv_layer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("target,expect_fallback", [(None, False), (3.0e5, True)])
def test_forward_with_activations_beyond_fp16_range(target, expect_fallback):
    """c0 -> c1 -> pool -> c2 with inputs scaled so that c0's outputs reach ~1e5: Net.forward() in split-fp16 mode
    still matches the oracle (the forward is redone on the fp32 kernels) and says so."""
    h, w = 40, 56
    txt = H.single_layer_net(
        conv_layer("c0", "data", 64, 3, 1) + conv_layer("c1", "c0", 128, 3, 1) +
        'layer { name: "p" type: "Pooling" bottom: "c1" top: "p" pooling_param { pool: MAX kernel_size: 2 stride: 2 } }\n' +
        conv_layer("c2", "p", 128, 3, 1), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=11)
    gnet.set_conv_mode("f16x3")
    data = np.random.default_rng(2).normal(0, 1, (1, 3, h, w)).astype(np.float32)
    if target:   # scale the input so that c0's largest output is `target` (> 65504)
        onet.blobs['data'].reshape(*data.shape)
        onet.blobs['im_info'].reshape(1, 3)
        onet.forward(data=data, im_info=np.array([[h, w, 1]], np.float32))
        data = data * np.float32(target / np.abs(onet.blobs["c0"].data).max())
    before = gnet.range_fallbacks
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    assert np.isfinite(go["c2"]).all()
    if expect_fallback:
        assert np.abs(onet.blobs["c0"].data).max() > 65504          # the case really leaves the fp16 range
        assert gnet.range_fallbacks == before + 1
    else:
        assert gnet.range_fallbacks == before
    for name in ("c0", "c1", "p", "c2"):
        assert H.rel_err(gnet.blobs[name].data, onet.blobs[name].data) < 2e-5, name
    # the mode is unchanged for the next call, which is clean again
    small = np.random.default_rng(3).normal(0, 1, (1, 3, h, w)).astype(np.float32)
    n0 = gnet.range_fallbacks
    go, oo = H.run_both(gnet, onet, small, np.array([[h, w, 1]], np.float32))
    assert gnet.range_fallbacks == n0 and H.rel_err(go["c2"], oo["c2"]) < 2e-5


def test_weights_beyond_fp16_range_refuse_the_mode():
    h, w = 16, 16
    txt = H.single_layer_net(conv_layer("c0", "data", 64, 3, 1) + conv_layer("c1", "c0", 128, 3, 1), 3, h, w)
    gnet, onet = H.make_pair(P.parse(txt), seed=12)
    onet.params["c1"][0][0, 0, 0, 0] = 1.0e5
    H.load_params(gnet, onet.params)
    with pytest.raises(Exception, match="outside the fp16 range"):
        gnet.set_conv_mode("f16x3")
    # still a working fp32 net
    data = np.random.default_rng(2).normal(0, 1, (1, 3, h, w)).astype(np.float32)
    go, oo = H.run_both(gnet, onet, data, np.array([[h, w, 1]], np.float32))
    assert H.rel_err(go["c1"], oo["c1"]) < 2e-5


def test_fused_detector_redoes_out_of_range_images_in_fp32():
    """The device-resident path: an image whose convolutions leave the fp16 range is redone on the exact kernels --
    same detections as the fp32 mode gives, neighbours in the two-image pipeline untouched."""
    from smallhardface_amd import test as T
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = True
    cfg.TEST.SCALES = [100, 300]
    gnet, _ = H.make_pair(H.detector_msg(True), cls_bias=1.0)
    ims = [np.random.default_rng(40 + i).integers(0, 256, (96, 128, 3)).astype(np.uint8) for i in range(3)]
    units = [list(T.pyramid_units(im)) for im in ims]
    # image 1: blobs x 2e5 -> conv1_1 outputs far beyond 65504 (the synthetic first layer has a gain of ~0.1)
    units[1] = [(u[0] * np.float32(2.0e5),) + tuple(u[1:]) for u in units[1]]
    gnet.set_conv_mode("fp32")
    ref = [T.FusedDetector(gnet, n_lanes=4, mode="group").detect(u, thresh=0.05)[0] for u in units]
    gnet.set_conv_mode("f16x3")
    fd = T.FusedDetector(gnet, n_lanes=4, mode="group")
    one = fd.detect(units[1], thresh=0.05)[0]                     # un-pipelined form
    assert fd.range_fallbacks == 1
    np.testing.assert_array_equal(one, ref[1])
    got = []
    for u in units:                                               # two images in flight
        fd.submit(u, thresh=0.05)
        if fd.pending() > 1:
            got.append(fd.collect()[0])
    while fd.pending():
        got.append(fd.collect()[0])
    assert fd.range_fallbacks == 2 and len(got) == 3
    np.testing.assert_array_equal(got[1], ref[1])                 # redone exactly
    for k in (0, 2):                                              # neighbours: split-fp16 results, within tolerance of fp32
        assert abs(len(got[k]) - len(ref[k])) <= 2
        n = min(len(got[k]), len(ref[k]))
        assert np.abs(got[k][:n, 4] - ref[k][:n, 4]).max() < 1e-4
    assert np.isfinite(np.concatenate(got)).all()
