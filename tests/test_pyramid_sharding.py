"""Host logic of the multi-GPU path on CPU: the unit schedule and the detection gather
(gloo, world_size 2 -- the same code runs over RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest

from smallhardface_amd import pyramid


def test_schedule_is_balanced_and_complete():
    for world in (1, 2, 4, 8):
        seen = set()
        for r in range(world):
            mine = pyramid.my_units(r, world, world, 10)
            assert len(mine) == 10                      # every rank: 10 units per window
            assert sorted(u for _, u in mine) == list(range(10))  # ... one of each (level, flip) kind
            seen.update(mine)
        assert len(seen) == world * 10                  # every unit of every image exactly once
    assert [pyramid.strict_level_rank(l, 8) for l in range(5)] == [0, 1, 2, 3, 4]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_images = world
        local = {}
        for i in range(n_images):
            rng = np.random.default_rng(100 * rank + i)
            n = int(rng.integers(0, 7)) if not (rank == 1 and i == 0) else 0  # an empty contribution too
            local[i] = torch.from_numpy(rng.normal(size=(n, 5)).astype(np.float32))
        got = pyramid.gather_window(local, n_images, rank, world)
        q.put((rank, {i: t.numpy() for i, t in got.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_window_gloo_world2():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for i in range(world):
        owner = pyramid.image_owner(i, world)
        assert i in res[owner] and i not in res[1 - owner]
        exp = []
        for r in range(world):
            rng = np.random.default_rng(100 * r + i)
            n = int(rng.integers(0, 7)) if not (r == 1 and i == 0) else 0
            exp.append(rng.normal(size=(n, 5)).astype(np.float32))
        np.testing.assert_array_equal(res[owner][i], np.concatenate(exp, 0))


def test_gather_window_single():
    import torch
    local = {0: torch.zeros((3, 5))}
    assert pyramid.gather_window(local, 1, 0, 1)[0] is local[0]


def test_pyramid_level_shape_matches_host_rounding():
    """C ABI shf_pyramid_level_shape == the host mirror's dsize / pad arithmetic (no GPU needed)."""
    from smallhardface_amd import caffe
    rng = np.random.default_rng(3)
    cases = [(600, 800, 0.5), (5, 7, 0.5), (3, 3, 2.5), (101, 203, 1.0), (1000, 1500, 0.8533333)]
    cases += [(int(h), int(w), float(s)) for h, w, s in
              zip(rng.integers(8, 2000, 200), rng.integers(8, 2000, 200), rng.uniform(0.05, 3.0, 200))]
    for h, w, s in cases:
        lh = h if s == 1.0 else int(np.round(h * s))
        lw = w if s == 1.0 else int(np.round(w * s))
        m = 16
        want = (lh, lw, int(np.ceil(1.0 * lh / m) * m), int(np.ceil(1.0 * lw / m) * m))
        assert caffe.pyramid_level_shape(h, w, s, m) == want, (h, w, s)
    with pytest.raises(ValueError):
        caffe.pyramid_level_shape(0, 5, 1.0, 16)
