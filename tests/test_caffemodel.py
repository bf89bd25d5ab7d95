"""``.caffemodel`` wire format: python writer <-> the runtime's C++ reader."""
import numpy as np
import pytest

from smallhardface_amd import caffemodel, weights
from tests import helpers as H


def test_roundtrip_cpu(tmp_path):
    msg = H.detector_msg(True)
    params = weights.synth_params(msg, seed=3)
    p = caffemodel.write_caffemodel(str(tmp_path / "m.caffemodel"), params)
    for name in ("conv1_1", "conv4_3", "head_1", "cls_score_2", "conv5_256_up"):
        for i, arr in enumerate(params[name]):
            got = caffemodel.read_blob(p, name, i)
            assert got.shape == arr.shape
            np.testing.assert_array_equal(got, arr)
    with pytest.raises(Exception, match="no layer named"):
        caffemodel.read_blob(p, "nope", 0)


@pytest.mark.gpu
def test_net_loads_caffemodel(tmp_path):
    """caffe.Net(proto, weights, TEST) copies blobs by layer NAME (net.cpp:733-768)."""
    from smallhardface_amd import caffe, prototxt as P
    msg = H.detector_msg(True)
    params = weights.synth_params(msg, seed=3)
    mp = caffemodel.write_caffemodel(str(tmp_path / "m.caffemodel"), params)
    pp = str(tmp_path / "test.prototxt")
    open(pp, "w").write(P.dumps(msg))
    a = caffe.Net(pp, mp, caffe.TEST)
    b = caffe.Net(pp, None, caffe.TEST)
    H.load_params(b, params)
    np.testing.assert_array_equal(a.params["head_4"][0].data, params["head_1"][0])
    data = H.synth_image_blob(48, 64, seed=2)
    info = np.array([[48, 64, 1.0]], np.float32)
    outs = []
    for net in (a, b):
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(1, 3)
        o = net.forward(data=data, im_info=info)
        outs.append((o["boxes"].copy(), o["cls_prob"].copy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception, match="Could not open|could not open"):
        caffe.Net(pp, str(tmp_path / "missing.caffemodel"), caffe.TEST)
