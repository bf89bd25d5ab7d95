"""The PRODUCT's host-side functions against the vectors produced by the reference's own code
(tests/golden/make_golden.py).  CPU only: nothing here launches a kernel -- the GPU halves of the same
fixtures (HIP tail on proposal.npz, device flip-fix/unscale on forward_net.npz, detect() with the HIP
bbox_vote on detect.npz) are in tests/test_gpu_golden.py.

  forward_net.npz           smallhardface_amd.test.forward_net          lib/test.py:21-106
  pyramid_scales.npz        test_utils._compute_scaling_factor / pyramid_scales   lib/utils/test_utils.py:8-26, lib/test.py:131-137
  config_smallhardface.json config.cfg_from_file / cfg_from_list        lib/utils/get_config.py:94-158
  anchors.npz               nms.generate_anchors -> C ABI shf_generate_anchors    lib/layers/generate_anchors.py:11-86
  template_digest.json      prototxt.build_test_template                models/test_*template.prototxt
"""
import hashlib
import json
import os

import numpy as np
import pytest

from smallhardface_amd.config import cfg, cfg_from_file, cfg_from_list
from tests.golden.make_golden import FakeNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("i", range(4))
def test_forward_net_flip_unscale_tile(golden, i):
    from smallhardface_amd import test as T
    g = golden("forward_net.npz")
    h, w, s, flip = g["c%d_args" % i]
    h, w, flip = int(h), int(w), bool(flip)
    net = FakeNet(int(g["c%d_seed" % i][0]))
    blob = {"data": np.random.default_rng(i).normal(0, 50, (1, 3, h, w)).astype(np.float32)}
    probs, boxes = T.forward_net(net, blob, float(s), pyramid=True, flip=flip)
    assert len(probs) == 1 and len(boxes) == 1
    np.testing.assert_array_equal(probs[0], g["c%d_probs" % i])
    assert boxes[0].dtype == g["c%d_boxes" % i].dtype
    np.testing.assert_array_equal(boxes[0], g["c%d_boxes" % i])          # (R,8): unscaled, tiled per class
    np.testing.assert_array_equal(np.array(net.calls[0][0]), g["c%d_fed_shape" % i])   # padded to x16
    np.testing.assert_array_equal(net.calls[0][1], g["c%d_fed_im_info" % i])           # UNPADDED dims + scale
    np.testing.assert_array_equal(net.blobs["boxes"].data, g["c%d_raw_boxes_after" % i])  # in-place flip fix
    with pytest.raises(NotImplementedError, match="Please complete this part!"):
        T.forward_net(FakeNet(1), {"data": blob["data"]}, 1.0, pyramid=False)      # test.py:84-88


def test_pyramid_scales(golden):
    from smallhardface_amd.test_utils import _compute_scaling_factor, pyramid_scales
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    g = golden("pyramid_scales.npz")
    assert len(g.files) == 6
    for key in g.files:
        hh, ww = [int(v) for v in key.split("x")]
        shape = (hh, ww, 3)
        base = _compute_scaling_factor(shape, cfg.TEST.PYRAMID_BASE_SIZE[0], cfg.TEST.PYRAMID_BASE_SIZE[1])
        got = np.array([base] + list(pyramid_scales(shape)), dtype=np.float64)
        np.testing.assert_array_equal(got, g[key])
    np.testing.assert_array_equal(g["1024x1024"][1:], [0.09765625, 0.29296875, 0.5859375, 0.9765625, 1.3671875])


def test_config_matches_reference_dump():
    want = json.load(open(os.path.join(GOLDEN, "config_smallhardface.json")))
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    cfg.TEST.NO_CACHE = True                                       # train_test.py:58
    cfg_from_list(["TEST.MODEL", "dummy.caffemodel", "TEST.GPU_ID", "[0]"])
    assert cfg.MAX_RESOLUTION == want["MAX_RESOLUTION"]
    assert cfg.PIXEL_MEANS == want["PIXEL_MEANS"]
    assert cfg.USE_GPU_NMS == want["USE_GPU_NMS"]
    assert cfg.MODEL.DIFFERENT_DILATION.ENABLE == want["MODEL.DIFFERENT_DILATION.ENABLE"]
    for k, v in want["TEST"].items():
        assert cfg.TEST[k] == v, k
        assert type(cfg.TEST[k]) is type(v), k
    # the reference's error behaviour (get_config.py:94-158)
    with pytest.raises(AssertionError, match="Please put NO_SUCH_KEY in default.toml"):
        cfg_from_list(["TEST.NO_SUCH_KEY", "1"])
    with pytest.raises(AssertionError):
        cfg_from_list(["TEST.MODEL"])


def test_generate_anchors_through_the_c_abi(golden):
    from smallhardface_amd.nms import generate_anchors
    g = golden("anchors.npz")
    a = generate_anchors(scales=np.array([1, 2, 4]), base_size=16, ratios=np.array([1]), shifts=np.array([0]),
                         strides=np.array([8, 8, 8]))
    assert a.dtype == np.float64
    np.testing.assert_array_equal(a, g["default_param_str"])
    np.testing.assert_array_equal(
        generate_anchors(scales=np.array((8, 16, 32)), base_size=16, ratios=np.array((0.5, 1, 2)),
                         shifts=np.array([0]), strides=np.array([16] * 3)), g["frcnn_defaults"])
    np.testing.assert_array_equal(
        generate_anchors(scales=np.array([2, 3]), base_size=8, ratios=np.array([0.5, 2]), shifts=np.array([0]),
                         strides=np.array([8, 8])), g["two_ratios_base8"])


@pytest.mark.parametrize("key,dd", [("plain", False), ("different_dilation", True)])
def test_generated_template_equals_the_reference_template(key, dd):
    """build_test_template() == the reference's models/test_*template.prototxt, field for field (the digest is
    the SHA-256 of the reference file's canonical dump through the same parser/printer)."""
    from smallhardface_amd import prototxt as P
    want = json.load(open(os.path.join(GOLDEN, "template_digest.json")))[key]
    msg = P.build_test_template(dd)
    canon = P.dumps(msg)
    assert len(msg.getall("layer")) == want["n_layers"]
    assert len(canon) == want["n_chars"]
    assert hashlib.sha256(canon.encode()).hexdigest() == want["sha256"]
    assert P.parse(canon) == msg                       # printer/parser round trip


def test_c_abi_exports_every_declared_symbol_and_the_allocation_counters_answer_without_a_gpu():
    """include/shf_hip.h vs the built library (no compute call: CPU containers have no GPU), and the round-4 measurement
    entry point shf_alloc_counts, which only reads two process-wide counters."""
    from smallhardface_amd import _lib, caffe
    lib = _lib.load(require_gpu=False)
    names = _lib.declared_symbols()
    assert len(names) > 50 and "shf_alloc_counts" in names and "shf_prof_only" in names
    missing = [s for s in names if not hasattr(lib, s)]
    assert not missing, missing
    d, h = caffe.alloc_counts()
    assert isinstance(d, int) and isinstance(h, int) and d >= 0 and h >= 0
