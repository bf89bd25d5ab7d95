"""Shared builders for the parity tests: the same graph + the same seeded synthetic
weights in the numpy oracle (CPU checker) and in the HIP runtime (through Net.params)."""
import numpy as np

from oracle import oracle as O
from smallhardface_amd import prototxt as P
from smallhardface_amd.config import cfg


def detector_msg(different_dilation=True):
    old = cfg.MODEL.DIFFERENT_DILATION.ENABLE
    cfg.MODEL.DIFFERENT_DILATION.ENABLE = different_dilation
    try:
        return P._add_dimension_reduction(P.build_test_template(different_dilation))
    finally:
        cfg.MODEL.DIFFERENT_DILATION.ENABLE = old


def load_params(gpu_net, params):
    for name, blobs in params.items():
        for i, arr in enumerate(blobs):
            gpu_net.params[name][i].data[...] = arr
    gpu_net.commit_params()


def make_pair(msg, seed=1234, cls_bias=4.0):
    """(gpu_net, oracle_net) holding identical parameters."""
    from smallhardface_amd import caffe
    params = O.synth_params(msg, seed=seed, cls_bias=cls_bias)
    onet = O.OracleNet(msg, params=params)
    gnet = caffe.Net(None, prototxt_text=P.dumps(msg))
    load_params(gnet, params)
    return gnet, onet


def run_both(gnet, onet, data, im_info):
    for net in (gnet, onet):
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(*im_info.shape)
    go = gnet.forward(data=data, im_info=im_info)
    oo = onet.forward(data=data, im_info=im_info)
    return go, oo


def synth_image_blob(h, w, seed=0):
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 256, (h, w, 3)).astype(np.float32) - np.array(cfg.PIXEL_MEANS, dtype=np.float32)[0]
    return np.ascontiguousarray(im.transpose(2, 0, 1)[None], dtype=np.float32)


def single_layer_net(layers_txt, cin, h=8, w=8):
    return ('name: "t"\ninput: "data"\ninput_shape { dim: 1 dim: %d dim: %d dim: %d }\n'
            'input: "im_info"\ninput_shape { dim: 1 dim: 3 }\n' % (cin, h, w)) + layers_txt


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
