"""Pin the numpy oracle's Python half against vectors produced by the
reference's own code (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import oracle as O


def rows_sorted(a):
    a = np.asarray(a)
    if a.shape[0] == 0:
        return a
    return a[np.lexsort(a.T[::-1])]


def test_anchors(golden):
    g = golden("anchors.npz")
    np.testing.assert_array_equal(
        O.generate_anchors(16, [1], [1, 2, 4], [0], [8, 8, 8]), g["default_param_str"])
    np.testing.assert_array_equal(g["default_param_str"],
                                  [[0, 0, 15, 15], [-8, -8, 23, 23], [-24, -24, 39, 39]])
    np.testing.assert_array_equal(
        O.generate_anchors(16, (0.5, 1, 2), (8, 16, 32), [0], [16] * 3), g["frcnn_defaults"])
    np.testing.assert_array_equal(
        O.generate_anchors(8, [0.5, 2], [2, 3], [0], [8, 8]), g["two_ratios_base8"])


def test_bbox_transform(golden):
    g = golden("bbox_transform.npz")
    p = O.bbox_transform_inv(g["boxes"], g["deltas"])
    np.testing.assert_array_equal(p, g["pred"])
    assert p.dtype == np.float32
    np.testing.assert_array_equal(O.clip_boxes(p.copy(), g["im_shape"]), g["clipped"])
    np.testing.assert_array_equal(O.bbox_transform_inv(g["boxes"], g["deltas_overflow"]), g["pred_overflow"])
    np.testing.assert_array_equal(O.bbox_transform_inv(g["boxes"], g["deltas_big"]), g["pred_big"])


@pytest.mark.parametrize("case", ["small", "unpadded", "wide", "all_below", "over_10000", "c1_512",
                                  "overflow", "ties"])
def test_proposal(golden, case):
    g = golden("proposal.npz")
    boxes, probs = O.proposal_forward(g[case + "_scores"], g[case + "_deltas"], g[case + "_im_info"])
    gb, gp = g[case + "_boxes"], g[case + "_probs"]
    assert boxes.shape == gb.shape and probs.shape == gp.shape
    if case == "over_10000":
        assert gb.shape[0] == 10000
    if case == "all_below":
        assert gb.shape[0] == 1  # nothing >= SCORE_THRESH: the best one is kept
    # scores are produced in descending order by both
    np.testing.assert_array_equal(probs[:, 1], gp[:, 1])
    # rows compared as a set of (score, box) records: tie order is implementation-defined
    a = rows_sorted(np.hstack([probs, boxes]))
    b = rows_sorted(np.hstack([gp, gb]))
    np.testing.assert_array_equal(a, b)
    if len(np.unique(gp[:, 1])) == gp.shape[0]:  # no tied scores: order is defined
        np.testing.assert_array_equal(boxes, gb)


VOTE_SETS = ["empty", "single", "two_overlap", "singletons", "last_singleton", "clusters_small",
             "clusters_mid", "clusters_big", "ties", "dense", "iou_exact_0p4"]


@pytest.mark.parametrize("name", VOTE_SETS)
def test_bbox_vote(golden, name):
    g = golden("vote_nms.npz")
    # replay the reference's own (tie-order-undefined) permutation -> exact equality
    out = O.bbox_vote(g[name + "_dets"], order=g[name + "_order"])
    ref = g[name + "_vote"]
    assert out.shape == ref.shape
    np.testing.assert_array_equal(np.asarray(out, dtype=np.float64), ref)
    if name != "ties":  # without tied scores the canonical order is the same order
        np.testing.assert_array_equal(np.asarray(O.bbox_vote(g[name + "_dets"]), dtype=np.float64), ref)


@pytest.mark.parametrize("name", VOTE_SETS)
@pytest.mark.parametrize("thr", [0.4, 0.3, 0.7])
def test_nms(golden, name, thr):
    g = golden("vote_nms.npz")
    d = g[name + "_dets"]
    keep = O.nms(d, thr, order=g[name + "_order"])
    ref = g[name + "_nms_%02d" % int(thr * 100)]
    np.testing.assert_array_equal(keep, ref)
    if name != "ties":
        np.testing.assert_array_equal(O.nms(d, thr), ref)
    if name == "iou_exact_0p4" and thr == 0.4:
        # IoU == thr is NOT suppressed by the canonical '>' predicate ...
        assert len(keep) == 4
        # ... but is by the Cython '>=' variant (cpu_nms.pyx:65)
        assert len(O.nms_ge(d, thr)) == 2
