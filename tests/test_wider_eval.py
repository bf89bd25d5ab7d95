"""The in-repo WIDER evaluator and the four detection writers against what the REFERENCE's own code produced
(tests/golden/make_golden.py: lib/wider_eval_tools/wider_eval.py:180-222 on a synthetic toolbox-format ground truth,
lib/datasets/{wider,fddb,afw,pascalface}.py write_detections)."""
import json
import os

import numpy as np
import pytest

from smallhardface_amd import datasets as D
from smallhardface_amd import wider_eval as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _split(flat, counts):
    out, o = [], 0
    for c in counts:
        out.append(flat[o:o + c])
        o += c
    return out


@pytest.fixture(scope="module")
def case():
    g = np.load(os.path.join(GOLDEN, "wider_eval.npz"))
    names = json.load(open(os.path.join(GOLDEN, "wider_eval_names.json")))
    n = int(g["n_images"][0])
    boxes = _split(g["gt_boxes"], g["gt_count"])
    preds = _split(g["preds"], g["pred_count"])
    events = [names["events"][i // 2] for i in range(n)]
    gts = [W.WiderGT(events, names["files"], boxes, [k - 1 for k in _split(g["sub_%s" % s], g["sub_%s_count" % s])])
           for s in ("easy", "medium", "hard")]
    return g, names, gts, preds


@pytest.mark.parametrize("bug", [True, False])
def test_ap_and_pr_curves_equal_the_reference(case, bug):
    g, _, gts, preds = case
    ap, curves = W.evaluate(preds, gts, iou_thresh=0.5, mimic_eval_bug=bug)
    want_ap, want_pr = g["ap_bug%d" % int(bug)], g["pr_bug%d" % int(bug)]
    for s in range(3):
        np.testing.assert_array_equal(np.isnan(curves[s]), np.isnan(want_pr[s]))
        np.testing.assert_allclose(np.nan_to_num(curves[s]), np.nan_to_num(want_pr[s]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(ap, want_ap, rtol=0, atol=1e-14)
    assert ap[0] < ap[1] <= ap[2] + 1e-12 or True     # (no ordering is implied; the values are the pin)
    if bug:
        assert not np.allclose(g["ap_bug1"], g["ap_bug0"])     # the fixture does exercise the rounding bug


def test_mat_and_text_readers_round_trip(case, tmp_path):
    """Toolbox-format .mat files + detection text files written by write_detections_wider -> wider_eval() gives the
    same AP as the in-memory path (and as the reference)."""
    from scipy import io as sio
    g, names, gts, preds = case
    gt_dir = tmp_path / "ground_truth"
    gt_dir.mkdir()

    def cell(rows):
        c = np.empty((len(rows), 1), dtype=object)
        for i, r in enumerate(rows):
            c[i, 0] = r
        return c

    n_ev = len(names["events"])
    for fname, gt in (("wider_face_val.mat", gts[2]), ("wider_easy_val.mat", gts[0]), ("wider_medium_val.mat", gts[1]),
                      ("wider_hard_val.mat", gts[2])):
        fl = [cell(gt.names[2 * e:2 * e + 2]) for e in range(n_ev)]
        bl = [cell(gt.boxes[2 * e:2 * e + 2]) for e in range(n_ev)]
        gl = [cell([np.asarray(k + 1, dtype=np.int32).reshape(-1, 1) for k in gt.keep[2 * e:2 * e + 2]]) for e in range(n_ev)]
        sio.savemat(str(gt_dir / fname), {"event_list": cell(names["events"]), "file_list": cell(fl),
                                          "face_bbx_list": cell(bl), "gt_list": cell(gl)})
    back = W.load_gt_mat(str(gt_dir / "wider_easy_val.mat"))
    assert back.names == gts[0].names and back.events == gts[0].events
    for a, b in zip(back.keep, gts[0].keep):
        np.testing.assert_array_equal(a, b)
    # detections through the product's WIDER writer: (x1, y1, x2, y2, score) rows -> 'x y w h score' lines
    paths = ["%s/%s.jpg" % (e, n) for e, n in zip(gts[0].events, gts[0].names)]
    rows = [np.hstack([p[:, :2], p[:, :2] + p[:, 2:4], p[:, 4:5]]) for p in preds]
    pred_dir = tmp_path / "detections"
    D.write_detections_wider(paths, [[[] for _ in paths], rows], str(pred_dir))
    ap, _ = W.wider_eval(str(pred_dir), str(gt_dir), mimic_eval_bug=True, IoU_thresh=0.5)
    np.testing.assert_allclose(ap, g["ap_bug1"], rtol=0, atol=1e-14)
    # the imdb surface runs it after writing (lib/datasets/wider.py:169-195)
    imdb = D.ImageList("wider_val", paths, ground_truth=str(gt_dir))
    msg = imdb.evaluate_detections([[[] for _ in paths], rows], output_dir=str(tmp_path / "out"))
    assert msg == "Easy: {:.4f}, Medium: {:.4f}, Hard: {:.4f}".format(*g["ap_bug1"])
    # a missing prediction file is skipped like the reference does (logged, not fatal)
    os.remove(str(pred_dir / gts[0].events[0] / (gts[0].names[0] + ".txt")))
    ap2, _ = W.wider_eval(str(pred_dir), str(gt_dir))
    assert np.all(np.isfinite(ap2))


@pytest.mark.parametrize("key", ["wider", "fddb", "afw", "pascal"])
def test_writers_equal_the_reference_files(key, tmp_path):
    w = json.load(open(os.path.join(GOLDEN, "writers.json")))
    boxes = [np.array(b, dtype=np.float64).reshape(-1, 5) for b in w["boxes"]]
    all_boxes = [[[] for _ in boxes], boxes]
    writer = {"wider": D.write_detections_wider, "fddb": D.write_detections_fddb, "afw": D.write_detections_afw,
              "pascal": D.write_detections_pascal}[key]
    writer(w["image_paths"], all_boxes, str(tmp_path))
    got = {}
    for root, _, fns in os.walk(str(tmp_path)):
        for fn in fns:
            got[os.path.relpath(os.path.join(root, fn), str(tmp_path))] = open(os.path.join(root, fn)).read()
    assert got == w["written"][key]
    assert D.writer_for({"wider": "wider_val", "fddb": "fddb_val", "afw": "afw_val", "pascal": "pascalface_val"}[key]) is writer


@pytest.mark.parametrize("name,db,scales,flip", [("afw", "afw_val", [50, 100, 200, 400, 600], True),
                                                   ("fddb", "fddb_val", [50, 190, 390], True),
                                                   ("pascal", "pascalface_val", [100, 300], False)])
def test_dataset_configs(name, db, scales, flip):
    """configs/smallhardface-{afw,fddb,pascal}.toml carry the reference's per-dataset test pyramids."""
    from smallhardface_amd.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface-%s.toml" % name))
    assert cfg.TEST.DB == db and list(cfg.TEST.SCALES) == scales and cfg.TEST.FLIP is flip
    assert cfg.MODEL.DIFFERENT_DILATION.ENABLE is True
