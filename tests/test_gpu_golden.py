"""The HIP kernels against the vectors produced by the REFERENCE's own Python (tests/golden/make_golden.py),
through the C ABI -- the GPU halves of tests/test_product_golden.py:

  proposal.npz     tail decode -> select -> sort -> gather (csrc/tail.hip) via shf_debug_proposal
                   lib/layers/proposal_layer.py:60-220, lib/utils/bbox_transform.py:33-93 (incl. the overflow clamp)
  forward_net.npz  device flip fix + unscale + >thresh cut (append_dets_kernel) via shf_debug_append
                   lib/test.py:52-54,59-66,163-167
  detect.npz       the product's detect() (host driver + HIP bbox_vote) with the generator's canned net
                   lib/test.py:109-178,181-217
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from smallhardface_amd.config import cfg, cfg_from_file
from tests import helpers as H
from tests.golden.make_golden import FakeNet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BOX_TOL = 1e-3   # px: device expf vs numpy's float32 exp differ in the last ulp; everything else is op-for-op


@pytest.fixture(scope="module")
def net():
    from smallhardface_amd import caffe, prototxt as P
    return caffe.Net(None, prototxt_text=P.dumps(H.detector_msg(True)))


def rows_sorted(a):
    a = np.asarray(a)
    return a if a.shape[0] == 0 else a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("case", ["small", "unpadded", "wide", "all_below", "over_10000", "c1_512", "overflow", "ties"])
def test_hip_tail_on_reference_proposal_vectors(golden, net, case):
    g = golden("proposal.npz")
    sc, dl, ii = g[case + "_scores"], g[case + "_deltas"], g[case + "_im_info"]
    gb, gp = g[case + "_boxes"], g[case + "_probs"]
    net._apply_cfg()
    boxes, probs, overflow = net.debug_proposal(sc, dl, ii)
    assert boxes.shape == gb.shape and probs.shape == gp.shape
    assert overflow == (case == "overflow")          # np.seterr(over='raise') -> clamp branch, bbox_transform.py:52-65
    if case == "over_10000":
        assert len(boxes) == 10000
    if case == "all_below":
        assert len(boxes) == 1 and probs[0, 1] < cfg.TEST.SCORE_THRESH   # the single best anchor is kept
    # scores come out in the same (descending) order, bit for bit
    np.testing.assert_array_equal(probs, gp) if case != "ties" else np.testing.assert_array_equal(probs[:, 1], gp[:, 1])
    assert np.all(boxes[:, 0] == 0)
    if len(np.unique(gp[:, 1])) == gp.shape[0]:
        # no tied scores: row order is defined -> row-for-row against the reference
        assert np.abs(boxes - gb).max() < BOX_TOL
    # tie order of argsort()[::-1] is implementation-defined: row-for-row against the oracle's canonical order
    # (the oracle equals the fixture as a set, tests/test_oracle_golden.py) ...
    ob, op = O.proposal_forward(sc, dl, ii)
    np.testing.assert_array_equal(probs, op)
    assert np.abs(boxes - ob).max() < BOX_TOL
    # ... and as a set of (score, box) records against the reference itself
    a = rows_sorted(np.hstack([probs, np.round(boxes, 2)]))
    b = rows_sorted(np.hstack([gp, np.round(gb, 2)]))
    assert np.abs(a - b).max() < 0.011
    # every box is clipped to the UNPADDED image (im_info), proposal_layer.py:158
    assert boxes[:, [1, 3]].max() <= ii[0, 1] - 1 and boxes[:, [2, 4]].max() <= ii[0, 0] - 1 and boxes[:, 1:].min() >= 0


def test_hip_tail_empty_and_config_edges(net):
    """R == 0 (min-size filter removes everything) -> dummy roi [[0,0,0,16,16]] and an empty cls_prob
    (proposal_layer.py:207-215); pre_nms_topN and score_thresh are read from cfg like the reference's layer."""
    rng = np.random.default_rng(3)
    h, w = 7, 9
    fg = rng.uniform(0.01, 0.9, (3, h, w)).astype(np.float32)
    sc = np.concatenate([1 - fg, fg], 0)[None].astype(np.float32)
    dl = rng.normal(0, 0.2, (1, 12, h, w)).astype(np.float32)
    ii = np.array([[56, 72, 1.0]], np.float32)
    net.set_proposal_cfg(10000, 0.002, 5000.0)             # ANCHOR_MIN_SIZE larger than any box
    boxes, probs, _ = net.debug_proposal(sc, dl, ii)
    np.testing.assert_array_equal(boxes, [[0, 0, 0, 16, 16]])
    assert probs.shape == (0, 2)
    for topn, thr in ((17, 0.002), (10000, 0.5), (5, 0.95)):
        net.set_proposal_cfg(topn, thr, 0.0)
        boxes, probs, _ = net.debug_proposal(sc, dl, ii)
        ob, op = O.proposal_forward(sc, dl, ii, O.ProposalParams(pre_nms_topN=topn, score_thresh=thr))
        assert boxes.shape == ob.shape, (topn, thr)
        np.testing.assert_array_equal(probs, op)
        assert np.abs(boxes - ob).max() < BOX_TOL
    net._apply_cfg()


@pytest.mark.parametrize("i", range(4))
@pytest.mark.parametrize("thresh", [0.05, 0.5])
def test_device_flip_fix_and_unscale_on_reference_vectors(golden, net, i, thresh):
    import torch
    g = golden("forward_net.npz")
    h, w, s, flip = g["c%d_args" % i]
    h, w, flip = int(h), int(w), bool(flip)
    fake = FakeNet(int(g["c%d_seed" % i][0]))
    out = fake.forward(data=np.zeros((1, 3, h, w), np.float32), im_info=np.array([[h, w, s]], np.float32))
    raw_boxes, raw_probs = out["boxes"].copy(), out["cls_prob"].copy()       # what the net handed forward_net
    np.testing.assert_array_equal(raw_probs, g["c%d_probs" % i])
    want_boxes, want_probs = g["c%d_boxes" % i], g["c%d_probs" % i]          # the reference's forward_net outputs
    keep = want_probs[:, 1] > thresh                                          # detect(): test.py:163-167
    want = np.hstack([want_boxes[keep, :4], want_probs[keep, 1:2]]).astype(np.float32)
    net.detect_begin()
    net.debug_append(raw_boxes, raw_probs, w, float(s), flip, thresh)
    buf = torch.zeros((64, 5), dtype=torch.float32, device="cuda")
    n = net.detect_export(buf.data_ptr(), 64)
    got = buf[:n].cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)                                  # bit-exact: same fp32 ops
    assert net.detect_count() == len(want)
    # two units of one image land one after the other (test.py:141-158 concatenation order)
    net.detect_begin()
    net.debug_append(raw_boxes, raw_probs, w, float(s), flip, thresh)
    net.debug_append(raw_boxes, raw_probs, w, float(s), not flip, thresh)
    n = net.detect_export(buf.data_ptr(), 64)
    got2 = buf[:n].cpu().numpy()
    assert n == 2 * len(want)
    np.testing.assert_array_equal(got2[:len(want)], want)
    np.testing.assert_array_equal(got2[len(want):, [1, 3, 4]], want[:, [1, 3, 4]])


@pytest.mark.parametrize("i", range(2))
def test_detect_driver_on_reference_vectors(golden, i):
    """The product's detect(): pyramid scales, blob building, 10 forward_net calls in the reference's order,
    concat, > 0.05, HIP bbox_vote -- against what the reference's detect() returned for the same canned net."""
    from smallhardface_amd import test as T
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    g = golden("detect.npz")
    im = g["i%d_image" % i]
    fake = FakeNet(int(g["i%d_seed" % i][0]))
    cls_dets, timers = T.detect(fake, None, thresh=0.05, pyramid=True, im=im)
    assert len(cls_dets) == 1
    want = g["i%d_dets" % i]
    got = np.asarray(cls_dets[0])
    assert got.dtype == np.float64 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(np.array([c[0] for c in fake.calls]), g["i%d_fed_shapes" % i])
    np.testing.assert_array_equal(np.concatenate([c[1] for c in fake.calls]), g["i%d_fed_im_info" % i])
    np.testing.assert_array_equal(np.array([c[3] for c in fake.calls]), g["i%d_fed_first" % i])
    np.testing.assert_allclose(np.array([c[2] for c in fake.calls]), g["i%d_fed_sum" % i], rtol=1e-12)
    assert timers['detect'].calls == 1 and timers['misc'].calls == 1
