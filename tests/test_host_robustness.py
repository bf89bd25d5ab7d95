"""Host-side behaviour around the hot path that the reference gets wrong or leaves to a launcher:
a dead inference worker (lib/test.py:339 waits forever), and ``python bench.py --gpus N`` started as ONE command
(the reference starts its N workers from one command, lib/test.py:327-344)."""
import multiprocessing as mp
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, result_queue, mode):
    if mode == "die":
        os._exit(7)            # what an out-of-memory kill or a native crash looks like from the parent
    if mode == "silent":
        return                 # exits 0 without delivering
    if mode == "slow":
        time.sleep(1.5)
    result_queue.put((rank, ["dets of rank %d" % rank]))


def _start(modes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = []
    for rank, mode in enumerate(modes):
        p = ctx.Process(target=_worker, args=(rank, q, mode))
        p.daemon = True
        p.start()
        procs.append(p)
    return q, procs


def test_gather_results_orders_by_rank():
    from smallhardface_amd.test import _gather_results
    q, procs = _start(["slow", "ok", "ok"])
    got = _gather_results(q, procs)
    assert [g[0] for g in got] == [0, 1, 2] and got[2][1] == ["dets of rank 2"]
    for p in procs:
        p.join(10)


@pytest.mark.parametrize("mode,code", [("die", "7"), ("silent", "0")])
def test_dead_worker_raises_instead_of_hanging(mode, code, monkeypatch):
    from smallhardface_amd.config import cfg
    from smallhardface_amd.test import _gather_results
    monkeypatch.setattr(cfg.TEST, "GPU_ID", [0, 0, 0])
    q, procs = _start(["ok", mode, "slow"])
    t0 = time.time()
    with pytest.raises(RuntimeError) as e:
        _gather_results(q, procs, poll_seconds=0.2, grace_polls=5)
    assert time.time() - t0 < 30
    assert "inference worker 1" in str(e.value) and "code %s" % code in str(e.value)
    for p in procs:
        p.join(10)
        assert p.exitcode is not None      # the survivors were stopped


def test_bench_multi_gpu_bare_command_reports_missing_gpus():
    """`python bench.py --gpus 2` with no launcher around it: on a box with fewer than 2 GPUs the message is about the
    GPUs, not about how to launch (the bare command used to exit with "launch with torch.distributed.run")."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SHF_BENCH_ONE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 visible MI355X" in r.stderr and "torch.distributed.run" not in r.stderr


def test_bench_self_launch_starts_the_ranks(tmp_path):
    """The bare command really starts N ranks through torch.distributed.run: with SHF_BENCH_ONE_GPU=1 (no device-count
    gate in the parent) and no GPU here, every RANK reaches its own device check and says so."""
    import torch
    if torch.cuda.device_count() >= 1:
        pytest.skip("GPU box: covered by tests/test_gpu_fullsize.py (bare-command case)")
    env = dict(os.environ, PYTHONPATH=ROOT, SHF_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert r.stderr.count("needs 2 visible MI355X, found 0") >= 1
