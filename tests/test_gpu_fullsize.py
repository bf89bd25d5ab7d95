"""Full-size parity: BASELINE.json's configs C2..C5 as stated, HIP path vs the CPU oracle (``-m gpu``).

  C2  one 1024x1024 level, fp32 and split-fp16, vs oracle.OracleNet            (~7 s of host time)
  C3  8 x 1024x1024 units as ONE grouped launch + on-device NMS (IoU > 0.4)     (oracle nms on the GPU's candidates)
  C4  WIDER-shaped 768x1024 (HxW) source, 4 levels: two small levels vs the oracle net, the whole image through
      properties, and 4 ranks (strict one-scale-per-rank, gloo, all on this GPU) == 1 rank, bit for bit
  C5  one whole image of the bench workload (10 units, 5 TFLOP) through FusedDetector vs the reference driver
      control flow over the oracle net (lib/test.py:109-178), both conv arithmetics               (~45 s host)

Bars (north star): every anchor score within 1e-4, regression deltas 1e-3, same proposal count (+-boundary
flips), voted boxes' scores within 1e-4 and the written pixel coordinates (int truncation,
lib/datasets/wider.py:160-167) equal except where a coordinate sits within float noise of an integer --
those are counted and bounded.  A summary goes to gpurun_out/fullsize_parity.json.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from smallhardface_amd import prototxt as P
from smallhardface_amd.config import cfg, cfg_from_file
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_TOL = 1e-4
REPORT = {}


def _report(key, **kw):
    REPORT[key] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()}
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(REPORT, open(os.path.join(ROOT, "gpurun_out", "fullsize_parity.json"), "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def iou(a, b):
    """(n,4) x (m,4) IoU, +1 convention."""
    ix = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1
    iy = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1
    inter = np.clip(ix, 0, None) * np.clip(iy, 0, None)
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (aa[:, None] + ab[None, :] - inter)


def match_detections(got, want, score_tol=SCORE_TOL):
    """Pair every row of ``want`` with the row of ``got`` that is the same detection (score within tol, IoU > 0.98).
    Returns (pairs, unmatched_want, unmatched_got)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    used = np.zeros(len(got), bool)
    pairs, miss = [], []
    order = np.argsort(-got[:, 4], kind="stable")
    gs = got[order, 4]
    for j, w in enumerate(want):
        lo = np.searchsorted(-gs, -(w[4] + score_tol), side="left")
        hi = np.searchsorted(-gs, -(w[4] - score_tol), side="right")
        cand = [order[k] for k in range(lo, hi) if not used[order[k]]]
        best = None
        if cand:
            ious = iou(got[cand, :4], w[None, :4])[:, 0]
            k = int(np.argmax(ious))
            if ious[k] > 0.98:
                best = cand[k]
        if best is None:
            miss.append(j)
        else:
            used[best] = True
            pairs.append((best, j))
    return pairs, miss, list(np.where(~used)[0])


def written(d):
    """The integers wider.write_detections puts in the file (lib/datasets/wider.py:160-167)."""
    d = np.asarray(d, np.float64)
    x1, y1, x2, y2 = [d[:, k].astype(np.int64) for k in range(4)]
    return np.stack([x1, y1, x2 - x1, y2 - y1], 1)


def compare_detection_lists(key, got, want, max_unmatched=0, max_pixel_rows=0):
    """The north star's bar: every box matched, 0-pixel index difference (lib/datasets/wider.py:160-167 writes the
    truncated integers).  Round 4 measured 0 unmatched / 0 written-pixel rows at EVERY config (C3, C4, C5, the afw /
    fddb / pascal scale sets; gpurun_out/fullsize_parity.json), and the inputs are seeded, so 0 / 0 is the default;
    a caller may only pass a looser bar together with the measured non-zero it documents (none does today)."""
    pairs, miss, extra = match_detections(got, want)
    assert len(pairs) > 0
    gi = np.array([p[0] for p in pairs])
    wi = np.array([p[1] for p in pairs])
    ds = np.abs(np.asarray(got)[gi, 4] - np.asarray(want)[wi, 4]).max()
    dc = np.abs(np.asarray(got)[gi, :4] - np.asarray(want)[wi, :4]).max()
    wg, ww = written(np.asarray(got)[gi]), written(np.asarray(want)[wi])
    px_rows = int(np.any(wg != ww, axis=1).sum())
    # a written-pixel difference is only legitimate where the coordinate sits within float noise of an integer
    g4, w4 = np.asarray(got)[gi, :4], np.asarray(want)[wi, :4]
    crossing = np.floor(g4) != np.floor(w4)
    assert np.all(np.abs(g4 - w4)[crossing] < 2e-2), "pixel differences that are not integer-boundary crossings"
    _report(key, n_got=len(got), n_want=len(want), matched=len(pairs), unmatched_oracle=len(miss),
            unmatched_gpu=len(extra), max_abs_dscore=ds, max_abs_dcoord=dc, rows_with_written_pixel_diff=px_rows)
    assert ds < SCORE_TOL, ds
    assert len(miss) <= max_unmatched, (len(miss), len(want))
    assert len(extra) <= max_unmatched, (len(extra), len(want))
    assert px_rows <= max_pixel_rows, px_rows
    return pairs


# ------------------------------------------------------------------------------------------------------------
# C2: one 1024 x 1024 level
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2():
    msg = H.detector_msg(True)
    gnet, onet = H.make_pair(msg)
    data = H.synth_image_blob(1024, 1024, seed=22)
    info = np.array([[1024, 1024, 1.0]], np.float32)
    onet.blobs['data'].reshape(*data.shape)
    onet.blobs['im_info'].reshape(1, 3)
    oo = onet.forward(data=data, im_info=info)
    ref = {k: onet.blobs[k].data.copy() for k in ("cls_prob_reshape_output", "bbox_pred_output", "conv5_3", "conv1_2",
                                                    "conv4_fuse_final")}
    ref["boxes"], ref["cls_prob"] = oo["boxes"].copy(), oo["cls_prob"].copy()
    return gnet, data, info, ref


@pytest.mark.parametrize("mode", ["fp32", "f16x3"])
def test_c2_1024_level_vs_oracle(c2, mode):
    gnet, data, info, ref = c2
    gnet.set_conv_mode(mode)
    gnet.blobs['data'].reshape(*data.shape)
    gnet.blobs['im_info'].reshape(1, 3)
    go = gnet.forward(data=data, im_info=info)
    gp = gnet.blobs["cls_prob_reshape_output"].data
    assert gp.shape == (1, 6, 128, 128)                       # 49 152 anchors
    err = float(np.abs(gp - ref["cls_prob_reshape_output"]).max())
    derr = float(np.abs(gnet.blobs["bbox_pred_output"].data - ref["bbox_pred_output"]).max())
    e53 = H.rel_err(gnet.blobs["conv5_3"].data, ref["conv5_3"])
    e12 = H.rel_err(gnet.blobs["conv1_2"].data, ref["conv1_2"])
    eff = H.rel_err(gnet.blobs["conv4_fuse_final"].data, ref["conv4_fuse_final"])
    n = min(len(go["boxes"]), len(ref["boxes"]))
    srt = float(np.abs(go["cls_prob"][:n, 1] - ref["cls_prob"][:n, 1]).max())
    _report("C2_1024_" + mode, max_abs_dscore_all_anchors=err, max_abs_ddelta=derr, conv5_3_rel=e53, conv1_2_rel=e12,
            conv4_fuse_final_rel=eff, proposals_gpu=len(go["boxes"]), proposals_oracle=len(ref["boxes"]),
            max_abs_dscore_ranked=srt)
    assert err < SCORE_TOL, (mode, err)
    assert derr < 1e-3
    assert e53 < 5e-5 and e12 < 5e-5 and eff < 5e-5
    assert len(go["boxes"]) == len(ref["boxes"])         # (both sides: the 10 000 best of 49 152)
    assert srt < SCORE_TOL
    # the proposal stage on IDENTICAL inputs: order / indices exact at 49 152 anchors
    pb, pp = O.proposal_forward(gp, gnet.blobs["bbox_pred_output"].data, info)
    assert go["boxes"].shape == pb.shape
    np.testing.assert_array_equal(go["cls_prob"], pp)
    assert np.abs(go["boxes"] - pb).max() < 1e-3


# ------------------------------------------------------------------------------------------------------------
# C3: 8 x 1024 x 1024 as one grouped launch + device NMS
# ------------------------------------------------------------------------------------------------------------
def test_c3_batch8_grouped_launch_and_device_nms(c2):
    import torch
    from smallhardface_amd import test as T
    gnet, data0, info, ref = c2
    gnet.set_conv_mode("f16x3")
    cfg.TEST.NMS_METHOD = "NMS"
    blobs = [data0] + [H.synth_image_blob(1024, 1024, seed=300 + i) for i in range(7)]
    units = [(b, 1024, 1024, 1024, 1024, 1.0, False) for b in blobs]
    fd = T.FusedDetector(gnet, n_lanes=8, mode="group")
    head = fd.lanes[0]
    head.detect_add_levels(fd.lanes[:8], units, 0.05, per_member_lists=True)    # ONE grid per conv layer over the batch
    head.sync()
    buf = torch.empty((10000, 5), dtype=torch.float32, device="cuda")
    total = 0
    for m in range(8):
        n = fd.lanes[m].detect_export(buf.data_ptr(), 10000)
        cand = buf[:n].cpu().numpy()
        kept = fd.lanes[m].detect_finish("NMS", cfg.TEST.NMS_THRESH)              # greedy NMS on the device
        keep = O.nms(cand, cfg.TEST.NMS_THRESH)                                   # oracle on the GPU's own candidates
        assert len(kept) == len(keep)
        np.testing.assert_array_equal(kept, cand[keep].astype(np.float64))        # same boxes, same (score) order
        assert n > 0 and np.all(cand[:, 4] > 0.05)
        total += n
        if m == 0:
            # unit 0 is the C2 image: its candidates against the oracle NET's (>0.05, scale 1: no unscale)
            ob, op = ref["boxes"], ref["cls_prob"]
            k = op[:, 1] > 0.05
            want = np.hstack([ob[k, 1:5], op[k, 1:2]])
            compare_detection_lists("C3_unit0_candidates_vs_oracle_net", cand, want)
    # same units one at a time == the grouped batch (bit-exact: the group is a scheduling choice)
    gnet.detect_begin()
    gnet.detect_add_level(*units[3], 0.05)
    n1 = gnet.detect_export(buf.data_ptr(), 10000)
    single = buf[:n1].cpu().numpy()
    n3 = fd.lanes[3].detect_export(buf.data_ptr(), 10000)
    np.testing.assert_array_equal(single, buf[:n3].cpu().numpy())
    _report("C3_batch8", candidates_total=total)


# ------------------------------------------------------------------------------------------------------------
# C5: one whole image of the bench workload vs the reference control flow over the oracle net
# ------------------------------------------------------------------------------------------------------------
def test_c5_whole_image_vs_oracle_driver():
    from smallhardface_amd import caffe, test as T, weights
    import smallhardface_amd.test as tm
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    msg = P._add_dimension_reduction(P.build_test_template(True))
    params = weights.synth_params(msg, seed=1234)
    gnet = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(gnet, params)
    onet = O.OracleNet(msg, params=params)
    im = np.random.default_rng(1000).integers(0, 256, (1024, 1024, 3)).astype(np.uint8)   # bench.py's image 0
    units = list(T.pyramid_units(im))
    assert [u[1] for u in units] == [112, 112, 304, 304, 608, 608, 1008, 1008, 1408, 1408]
    # reference control flow (detect -> forward_net x 10 -> concat -> >0.05 -> bbox_vote) over the ORACLE net,
    # merging with the oracle's bbox_vote
    old = tm.bbox_vote
    tm.bbox_vote = lambda d, thresh=None: np.asarray(O.bbox_vote(d, cfg.TEST.NMS_THRESH), dtype=np.float64)
    try:
        od, _ = T.detect(onet, None, thresh=0.05, pyramid=True, im=im)
    finally:
        tm.bbox_vote = old
    want = np.asarray(od[0], np.float64)
    assert len(want) > 50
    for mode in ("f16x3", "fp32"):
        gnet.set_conv_mode(mode)
        fd = T.FusedDetector(gnet, n_lanes=10, mode="group")
        got = fd.detect(units, thresh=0.05)[0]
        compare_detection_lists("C5_image_" + mode, got, want, max_unmatched=0, max_pixel_rows=0)
        del fd


# ------------------------------------------------------------------------------------------------------------
# C4: WIDER-shaped source, 4 levels; strict one-scale-per-rank on 4 ranks == 1 rank
# ------------------------------------------------------------------------------------------------------------
def _c4_cfg():
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    cfg.TEST.SCALES = [300, 600, 1000, 1400]
    cfg.TEST.FLIP = False


def test_c4_wider_shaped_pyramid_vs_oracle():
    from smallhardface_amd import caffe, test as T, weights
    _c4_cfg()
    msg = P._add_dimension_reduction(P.build_test_template(True))
    params = weights.synth_params(msg, seed=1234)
    gnet = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(gnet, params)
    gnet.set_conv_mode("f16x3")
    onet = O.OracleNet(msg, params=params)
    im = np.random.default_rng(1000).integers(0, 256, (768, 1024, 3)).astype(np.uint8)
    units = list(T.pyramid_units(im))
    assert [(u[1], u[2]) for u in units] == [(304, 400), (608, 800), (1008, 1344), (1408, 1872)]
    assert [(u[3], u[4]) for u in units] == [(300, 400), (600, 800), (1000, 1333), (1400, 1867)]
    # the two small levels against the oracle net, unit by unit
    import torch
    buf = torch.empty((10000, 5), dtype=torch.float32, device="cuda")
    for u in units[:2]:
        data, Hh, Ww, ih, iw, s, flip = u
        info = np.array([[ih, iw, s]], np.float32)
        onet.blobs['data'].reshape(*data.shape)
        onet.blobs['im_info'].reshape(1, 3)
        oo = onet.forward(data=data, im_info=info)
        k = oo["cls_prob"][:, 1] > 0.05
        want = np.hstack([oo["boxes"][k, 1:5] / np.float32(s), oo["cls_prob"][k, 1:2]])
        gnet.detect_begin()
        gnet.detect_add_level(data, Hh, Ww, ih, iw, s, flip, 0.05)
        n = gnet.detect_export(buf.data_ptr(), 10000)
        compare_detection_lists("C4_level_%dx%d" % (Hh, Ww), buf[:n].cpu().numpy(), want)
    # the whole image: grouped == one unit at a time, plus range/order properties
    fd = T.FusedDetector(gnet, n_lanes=4, mode="group")
    a = fd.detect(units, thresh=0.05)[0]
    b = T.detect_fused(gnet, units, thresh=0.05)[0]
    np.testing.assert_array_equal(a, b)
    assert len(a) > 0 and np.all(np.diff(a[:, 4]) <= 0) and np.all(a[:, 4] > 0.05)
    assert a[:, [0, 2]].max() <= 1024 and a[:, [1, 3]].max() <= 768 and a[:, :4].min() >= -1e-3


def _run_bench(tmp_path, world, extra, tag, more_env=None, bare=False):
    out = str(tmp_path / ("dets_%s.npy" % tag))
    env = dict(os.environ, PYTHONPATH=ROOT, SHF_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(more_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
            "--no-latency", "--no-calib", "--no-reduced", "--no-mixed", "--no-forward-path", "--overlap-seconds", "0",
            "--sustain-seconds", "0", "--dump-dets", out] + extra
    if world == 1:
        cmd = [sys.executable] + base
    elif bare:
        # the way the driver starts every run: ONE bare command, bench.py starts its ranks itself
        cmd = [sys.executable] + base + ["--backend", "gloo"]
    else:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + base + ["--backend", "gloo"]
    # fresh child processes: nothing in them has touched the GPU before torchrun forks its workers
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (tag, r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return np.load(out), json.loads(line)


@pytest.mark.timeout(1800)
def test_c4_four_ranks_strict_one_scale_per_rank_equals_one_rank(tmp_path):
    """C4 as stated: the 4-level pyramid of WIDER-shaped sources sharded one scale per rank over 4 ranks with a
    detection gather (gloo here, all ranks on this one GPU; RCCL on a real node), == the 1-rank result."""
    c4 = ["--source", "768x1024", "--scales", "300,600,1000,1400", "--no-flip"]
    d1, j1 = _run_bench(tmp_path, 1, c4, "c4_n1")
    d4, j4 = _run_bench(tmp_path, 4, c4 + ["--shard", "strict"], "c4_n4_strict")
    assert j4["n_gpus"] == 4 and j4["config"]["shard"] == "strict" and j4["config"]["images_per_step"] == 4
    assert j4["collectives_issued_rank0"] > 0
    assert len(d1) > 0
    np.testing.assert_array_equal(d1, d4)
    dw, jw = _run_bench(tmp_path, 2, c4 + ["--shard", "window"], "c4_n2_window")
    np.testing.assert_array_equal(d1, dw)


@pytest.mark.timeout(1800)
def test_c5_two_and_four_ranks_equal_one_rank(tmp_path):
    """The bench workload itself (10 units/image): window schedule on 2 ranks and strict level->rank on 4 ranks
    give the 1-rank detections bit for bit."""
    d1, _ = _run_bench(tmp_path, 1, [], "c5_n1")
    d2, j2 = _run_bench(tmp_path, 2, ["--shard", "window"], "c5_n2")
    np.testing.assert_array_equal(d1, d2)
    d4, j4 = _run_bench(tmp_path, 4, ["--shard", "strict"], "c5_n4_strict")
    np.testing.assert_array_equal(d1, d4)
    assert j4["config"]["shard"] == "strict"
    # the bare command (no launcher around it: bench.py starts torch.distributed.run itself) is the same run
    db, jb = _run_bench(tmp_path, 2, ["--shard", "window"], "c5_n2_bare", bare=True)
    np.testing.assert_array_equal(d1, db)
    assert jb["n_gpus"] == 2 and jb["collective_ranks"] == 2 and jb["collectives_issued_rank0"] > 0


@pytest.mark.timeout(1800)
def test_c5_eight_ranks_like_the_scaling_run(tmp_path):
    """The command the driver's scaling run ends with -- `python bench.py --gpus 8` -- on the one GPU there is (gloo, all
    ranks on it): the balanced window schedule (every rank one unit of each kind per window of 8 images) and the north
    star's strict one-scale-per-GPU form (ranks 0-4 hold 2 flips x 8 images = 16 units = exactly one grouped pass, ranks
    5-7 only take part in the exchange) both reproduce the 1-rank detections bit for bit."""
    d1, _ = _run_bench(tmp_path, 1, [], "c5_n1_for8")
    d8, j8 = _run_bench(tmp_path, 8, ["--shard", "window"], "c5_n8_window", bare=True)
    np.testing.assert_array_equal(d1, d8)
    assert j8["n_gpus"] == 8 and j8["collective_ranks"] == 8 and j8["config"]["images_per_step"] == 8
    d8s, j8s = _run_bench(tmp_path, 8, ["--shard", "strict"], "c5_n8_strict", bare=True)
    np.testing.assert_array_equal(d1, d8s)
    assert j8s["config"]["shard"] == "strict"


@pytest.mark.timeout(1800)
def test_more_than_sixteen_units_per_share(tmp_path):
    """A share larger than one grouped pass (16 units: one kernel-argument member table): 9 scales x flip = 18 units per
    image on one rank (FusedDetector.submit splits the image into two passes into the same list), and strict level->rank
    sharding on 2 ranks (rank 0: 5 levels x 2 flips x 2 images = 20 units per window = two passes on two head lanes)
    -- the same detections bit for bit.  (the north star's strict mode at 8 ranks sits exactly AT 16.)"""
    many = ["--source", "240x320", "--scales", "100,150,200,250,300,350,400,450,500"]
    d1, j1 = _run_bench(tmp_path, 1, many, "u18_n1")
    assert j1["config"]["units_per_image"] == 18 and len(d1) > 0
    d2, j2 = _run_bench(tmp_path, 2, many + ["--shard", "strict"], "u18_n2_strict")
    np.testing.assert_array_equal(d1, d2)
    d1s, _ = _run_bench(tmp_path, 1, many + ["--mode", "streams", "--lanes", "3"], "u18_n1_streams")
    np.testing.assert_array_equal(d1, d1s)


@pytest.mark.timeout(1800)
def test_rccl_one_rank_group_runs_the_collective_path(tmp_path):
    """The RCCL branch executed for real on the one GPU there is: a fresh process initialises a ONE-rank
    `nccl` process group (device_id given, before any other GPU call), and bench.py --force-dist runs the N>1 schedule
    on it -- per-member lists, shf_detect_export_many, gather_window's device-tensor all_to_all_single over RCCL,
    shf_detect_import of the buffer RCCL produced, merge on the owner -- and must reproduce the plain 1-GPU detections
    bit for bit.  (What 8 GPUs add is more peers in the same collective; the reference's analogue is the Queue gather of
    lib/test.py:327-344.)"""
    d1, _ = _run_bench(tmp_path, 1, [], "c5_n1_plain")
    dr, jr = _run_bench(tmp_path, 1, ["--force-dist", "--backend", "nccl"], "c5_n1_rccl")
    assert jr["rccl_ranks"] == 1 and jr["collective_backend"] == "nccl" and jr["collectives_issued_rank0"] >= 2
    assert len(d1) > 0
    np.testing.assert_array_equal(d1, dr)
    # the persistent first pair's walk without its per-block tile table (the path launches with more than 300 tiles per
    # block take): ~106 tiles per block here, each decoded a tile ahead on a producer wave -- the same bits
    dt, _ = _run_bench(tmp_path, 1, [], "c5_n1_no_tile_table", more_env={"SHF_F16X3_PC_TAB": "0"})
    np.testing.assert_array_equal(d1, dt)


# ------------------------------------------------------------------------------------------------------------
# the other datasets' test pyramids (configs/smallhardface-{afw,fddb,pascal}.toml)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n_units", [("afw", 10), ("fddb", 6), ("pascal", 2)])
def test_dataset_scale_sets(name, n_units):
    """Each dataset's scale set x flip through the grouped device path: == one unit at a time, the small levels
    against the oracle net, range / order properties for the whole image."""
    import torch
    from smallhardface_amd import caffe, test as T, weights
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface-%s.toml" % name))
    msg = P._add_dimension_reduction(P.build_test_template(True))
    params = weights.synth_params(msg, seed=1234)
    gnet = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(gnet, params)
    gnet.set_conv_mode("f16x3")
    onet = O.OracleNet(msg, params=params)
    im = np.random.default_rng(55).integers(0, 256, (300, 400, 3)).astype(np.uint8)
    units = list(T.pyramid_units(im))
    assert len(units) == n_units
    fd = T.FusedDetector(gnet, n_lanes=n_units, mode="group")
    a = fd.detect(units, thresh=0.05)[0]
    b = T.detect_fused(gnet, units, thresh=0.05)[0]
    np.testing.assert_array_equal(a, b)
    assert len(a) > 0 and np.all(np.diff(a[:, 4]) <= 0) and np.all(a[:, 4] > 0.05)
    # (clipped at the scaled image, then divided by the scale in fp32: a last-ulp overshoot is the reference's too)
    assert a[:, [0, 2]].max() <= 400 + 1e-3 and a[:, [1, 3]].max() <= 300 + 1e-3 and a[:, :4].min() >= -1e-3
    buf = torch.empty((10000, 5), dtype=torch.float32, device="cuda")
    checked = 0
    for u in units:
        data, Hh, Ww, ih, iw, s, flip = u
        if Hh * Ww > 160 * 224 or checked >= 3:
            continue
        info = np.array([[ih, iw, s]], np.float32)
        onet.blobs['data'].reshape(*data.shape)
        onet.blobs['im_info'].reshape(1, 3)
        oo = onet.forward(data=data, im_info=info)
        ob = oo["boxes"].copy()
        if flip:
            ob[:, [1, 3]] = iw - ob[:, [3, 1]]
        k = oo["cls_prob"][:, 1] > 0.05
        want = np.hstack([ob[k, 1:5] / np.float32(s), oo["cls_prob"][k, 1:2]])
        gnet.detect_begin()
        gnet.detect_add_level(data, Hh, Ww, ih, iw, s, flip, 0.05)
        n = gnet.detect_export(buf.data_ptr(), 10000)
        if len(want):
            compare_detection_lists("%s_unit_%dx%d_flip%d" % (name, Hh, Ww, int(flip)), buf[:n].cpu().numpy(), want)
        else:
            assert n == 0
        checked += 1
    assert checked >= 1
