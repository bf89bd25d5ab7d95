"""``train_test.py --train false`` end to end on the GPU: TOML config, --amend, manipulate_test,
synthetic .caffemodel, image list, pyramid + flip, bbox_vote, WIDER-format detection files."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stderr_of(r, exp):
    """What the run wrote to stderr: its own pipe until train_test.py redirects fd 2 into <output_dir>/stderr.log
    (reference train_test.py:122-124), that file from then on."""
    text = r.stderr[-1500:]
    for root, _, files in os.walk(str(exp)):
        if "stderr.log" in files:
            text += "\n--- stderr.log ---\n" + open(os.path.join(root, "stderr.log")).read()[-3000:]
    return text


def _run_outputs(exp):
    """(stderr.log paths, cfgs.txt paths) under an experiment directory."""
    logs, cfgs = [], []
    for root, _, files in os.walk(str(exp)):
        logs += [os.path.join(root, f) for f in files if f == "stderr.log"]
        cfgs += [os.path.join(root, f) for f in files if f == "cfgs.txt"]
    return logs, cfgs


@pytest.mark.gpu
def test_cli_writes_wider_detections(tmp_path):
    from PIL import Image
    from smallhardface_amd import caffemodel, weights
    from tests import helpers as H
    data = tmp_path / "data"
    (data / "images" / "0--Parade").mkdir(parents=True)
    rng = np.random.default_rng(0)
    names = []
    for i, (h, w) in enumerate([(120, 160), (90, 140)]):
        p = data / "images" / "0--Parade" / ("img%d.jpg" % i)
        Image.fromarray(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).save(p)
        names.append("images/0--Parade/img%d.jpg" % i)
    (data / "wider_val.txt").write_text("\n".join(names) + "\n")
    model = str(tmp_path / "synthetic.caffemodel")
    caffemodel.write_caffemodel(model, weights.synth_params(H.detector_msg(True), cls_bias=1.0))
    runs = {}
    for fused in ("1", "0"):
        env = dict(os.environ, PYTHONPATH=ROOT, SHF_FUSED_DETECT=fused)
        exp = tmp_path / ("exp" + fused)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train_test.py"), "--train", "false", "--conf",
                            os.path.join(ROOT, "configs", "smallhardface.toml"), "--amend", "TEST.MODEL", model,
                            "DATA_DIR", str(data), "TEST.GPU_ID", "[0]", "TEST.SCALES", "[100, 300]", "EXP_DIR",
                            str(exp)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, _stderr_of(r, exp)
        found = []
        for root, _, files in os.walk(str(exp)):
            found += [os.path.join(root, f) for f in files if f.endswith(".txt") and "img" in f]
        assert len(found) == 2, (found, _stderr_of(r, exp))
        # train_test.py:122-132: stderr captured beside the results, the non-TRAIN configuration dumped as TOML
        import tomli
        logs, cfgs = _run_outputs(exp)
        assert len(logs) == 1 and len(cfgs) == 1
        dumped = tomli.loads(open(cfgs[0]).read())
        assert "TRAIN" not in dumped and dumped["TEST"]["SCALES"] == [100, 300] and dumped["TEST"]["MODEL"] == model
        assert dumped["EXP_DIR"] == str(exp) and dumped["TEST"]["NO_CACHE"] is True
        runs[fused] = [open(f).read().splitlines() for f in sorted(found)]
    lines = runs["1"][0]
    assert lines[0] == names[0] and int(lines[1]) == len(lines) - 2 and int(lines[1]) >= 1
    x, y, w, h, s = lines[2].split()
    assert int(w) >= 0 and int(h) >= 0 and 0.0 < float(s) <= 1.0
    # device-resident path (default) == one Net.forward() per unit (the reference's call pattern)
    for a, b in zip(runs["1"], runs["0"]):
        assert a[:2] == b[:2]
        for la, lb in zip(a[2:], b[2:]):
            fa, fb = la.split(), lb.split()
            assert fa[:4] == fb[:4] and abs(float(fa[4]) - float(fb[4])) <= 1e-4


def _run_cli(tmp_path, data, model, gpu_id, exp_name, scales="[100, 300]"):
    env = dict(os.environ, PYTHONPATH=ROOT)
    exp = tmp_path / exp_name
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_test.py"), "--train", "false", "--conf",
                        os.path.join(ROOT, "configs", "smallhardface.toml"), "--amend", "TEST.MODEL", model,
                        "DATA_DIR", str(data), "TEST.GPU_ID", gpu_id, "TEST.SCALES", scales, "EXP_DIR", str(exp)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, _stderr_of(r, exp)
    found = {}
    for root, _, files in os.walk(str(exp)):
        for f in files:
            if f.endswith(".txt") and "img" in f:
                found[f] = open(os.path.join(root, f)).read()
    return found, r


@pytest.mark.gpu
@pytest.mark.parametrize("n_images,gpu_id", [(4, "[0,0]"), (3, "[0,0,0,0]")])
def test_cli_one_process_per_gpu_id(tmp_path, n_images, gpu_id):
    """The reference's own multi-GPU mode (lib/test.py:327-345): one process per TEST.GPU_ID entry, contiguous image
    ranges of ceil(N / workers), results gathered through a Queue in rank order, `len(dets[0]) == len(imdb)` asserted.
    Two (resp. four: the last one gets an EMPTY range) spawn workers on the one test GPU must write the same detection
    files as the single-process run."""
    from PIL import Image
    from smallhardface_amd import caffemodel, weights
    from tests import helpers as H
    data = tmp_path / "data"
    (data / "images" / "0--Parade").mkdir(parents=True)
    rng = np.random.default_rng(3)
    names = []
    for i in range(n_images):
        h, w = 80 + 12 * i, 150 - 9 * i
        Image.fromarray(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).save(
            data / "images" / "0--Parade" / ("img%d.jpg" % i))
        names.append("images/0--Parade/img%d.jpg" % i)
    (data / "wider_val.txt").write_text("\n".join(names) + "\n")
    model = str(tmp_path / "synthetic.caffemodel")
    caffemodel.write_caffemodel(model, weights.synth_params(H.detector_msg(True), cls_bias=1.0))
    one, _ = _run_cli(tmp_path, data, model, "[0]", "exp_one")
    many, r = _run_cli(tmp_path, data, model, gpu_id, "exp_many")
    assert len(one) == n_images and sorted(one) == sorted(many), (sorted(one), sorted(many), _stderr_of(r, tmp_path / "exp_many"))
    for f in one:
        assert one[f] == many[f], f           # same process-independent arithmetic: byte-identical files
        assert int(one[f].splitlines()[1]) >= 1


@pytest.mark.gpu
@pytest.mark.timeout(1500)
@pytest.mark.parametrize("shard,world,n_images", [("pyramid", 2, 3), ("pyramid_strict", 3, 4), ("pyramid", 1, 2)])
def test_cli_pyramid_sharded_over_torchrun_ranks(tmp_path, shard, world, n_images):
    """TEST.SHARD pyramid (not in the reference, whose multi-GPU mode is the image-range split above): ``train_test.py
    --train false`` under ``torch.distributed.run`` shards every image's PYRAMID over the ranks (pyramid.ShardedDetector:
    windows of `world` images, this rank's units as grouped passes, ONE all_to_all of detections per window to the owner
    ranks, merge there) -- here 2 ranks (window schedule; 3 images = a full and a partly filled window) and 3 ranks (the
    north star's strict one-scale-per-rank form on 2 scales: rank 2 only takes part in the exchange), all on the one test
    GPU over gloo.  Rank 0 writes the detection files: byte-identical to the single-process run."""
    import socket
    from PIL import Image
    from smallhardface_amd import caffemodel, weights
    from tests import helpers as H
    data = tmp_path / "data"
    (data / "images" / "0--Parade").mkdir(parents=True)
    rng = np.random.default_rng(5)
    names = []
    for i in range(n_images):
        h, w = 84 + 10 * i, 146 - 7 * i
        Image.fromarray(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).save(
            data / "images" / "0--Parade" / ("img%d.jpg" % i))
        names.append("images/0--Parade/img%d.jpg" % i)
    (data / "wider_val.txt").write_text("\n".join(names) + "\n")
    model = str(tmp_path / "synthetic.caffemodel")
    caffemodel.write_caffemodel(model, weights.synth_params(H.detector_msg(True), cls_bias=1.0))
    one, _ = _run_cli(tmp_path, data, model, "[0]", "exp_one")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, SHF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SHF_DIST_TIMEOUT="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    exp = tmp_path / "exp_sharded"
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                "--master-addr", "127.0.0.1", "--master-port", str(port)] if world > 1 else [sys.executable]   # (1: no launcher, no group)
    r = subprocess.run(launcher + [os.path.join(ROOT, "train_test.py"), "--train", "false", "--conf",
                        os.path.join(ROOT, "configs", "smallhardface.toml"), "--amend", "TEST.MODEL", model,
                        "DATA_DIR", str(data), "TEST.GPU_ID", "[" + ",".join(["0"] * world) + "]", "TEST.SCALES", "[100, 300]",
                        "TEST.SHARD", shard, "EXP_DIR", str(exp)],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, _stderr_of(r, exp)
    many = {}
    for root, _, files in os.walk(str(exp)):
        for f in files:
            if f.endswith(".txt") and "img" in f:
                assert f not in many, "only rank 0 writes detection files"
                many[f] = open(os.path.join(root, f)).read()
    assert len(one) == n_images and sorted(one) == sorted(many), (sorted(one), sorted(many), _stderr_of(r, exp))
    for f in one:
        assert one[f] == many[f], f
    logs, cfgs = _run_outputs(exp)
    assert len(logs) == world and len(cfgs) == world          # every rank keeps its own stderr.log / cfgs.txt


def test_cli_refuses_training():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_test.py"), "--train", "true"],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "outside the scope" in (r.stderr + r.stdout)
