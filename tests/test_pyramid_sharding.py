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


def _contribution(rank, image, scenario):
    """Rows rank `rank` holds for window image `image` (seeded: the checker regenerates them)."""
    rng = np.random.default_rng(100 * rank + image)
    n = int(rng.integers(0, 7))
    if scenario == "overflow":
        n = int(rng.integers(0, 40))                      # more rows than the (shrunk) block capacity
    if rank == 1 and image == 0:
        n = 0                                             # an empty contribution
    if scenario == "strict8" and (rank >= 5 or image == 2):
        n = 0                                             # ranks beyond the 5 levels run nothing; image 2: no detections anywhere
    return rng.normal(size=(n, 5)).astype(np.float32)


def _worker(rank, world, port, q, n_images, scenario):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if scenario == "overflow":
            pyramid._GATHER_CAP["rows"] = 4               # forces the agreed second exchange with a larger capacity
        got = {}
        for window in range(2):                           # twice: the grown capacity persists, results stay the same
            local = {i: torch.from_numpy(_contribution(rank, i, scenario)) for i in range(n_images)}
            if scenario == "parts":                       # the per-unit export buffers as they are: lists of pieces
                local = {i: [t[:2], t[2:2], t[2:]] for i, t in local.items()}
            got = pyramid.gather_window(local, n_images, rank, world, force_collective=(world == 1))
        q.put((rank, {i: t.numpy() for i, t in got.items()}, pyramid._GATHER_CAP["rows"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n_images,scenario", [(2, 2, "plain"), (3, 5, "plain"), (2, 2, "overflow"),
                                                     (8, 8, "strict8"), (2, 3, "parts"), (1, 2, "parts"), (1, 1, "overflow")])
def test_gather_window_gloo(world, n_images, scenario):
    """One all_to_all per window, rows to the image's owner only: uneven ownership (5 images on 3 ranks), contributions
    larger than the block capacity (agreed re-exchange), and the strict one-scale-per-GPU form on 8 ranks with 5
    levels -- three ranks contribute nothing, one image has no detection on any rank; contributions handed over as
    lists of pieces; and a ONE-rank group with the collective forced (how the RCCL branch runs on a single-GPU box)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_images, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    res, caps = {}, set()
    for _ in procs:
        r, got, cap = q.get(timeout=240)
        res[r] = got
        caps.add(cap)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert len(caps) == 1                                  # every rank ended with the same capacity
    if scenario == "overflow":
        assert caps.pop() >= 32
    for i in range(n_images):
        owner = pyramid.image_owner(i, world)
        for r in range(world):
            assert (i in res[r]) == (r == owner)
        exp = np.concatenate([_contribution(r, i, scenario) for r in range(world)], 0)
        assert res[owner][i].shape == exp.shape and res[owner][i].dtype == np.float32
        np.testing.assert_array_equal(res[owner][i], exp)


def test_strict_sharding_with_more_ranks_than_levels():
    """8 ranks, 5 pyramid levels x 2 flips: levels land on ranks 0..4 (both flips together, every image of the window),
    ranks 5..7 hold nothing and only take part in the gather; every unit is run exactly once."""
    world, n_units = 8, 10
    seen = []
    for r in range(world):
        mine = pyramid.my_units(r, world, world, n_units, shard="strict")
        if r >= 5:
            assert mine == []
        else:
            assert sorted(set(u for _, u in mine)) == [2 * r, 2 * r + 1] and len(mine) == 2 * world
        seen += mine
    assert sorted(seen) == [(i, u) for i in range(world) for u in range(n_units)]
    with pytest.raises(ValueError):
        pyramid.my_units(0, 2, 2, 10, shard="columns")


def test_gather_window_single():
    import torch
    local = {0: torch.zeros((3, 5))}
    assert pyramid.gather_window(local, 1, 0, 1)[0] is local[0]
    parts = {0: [torch.ones((2, 5)), torch.zeros((0, 5)), 2 * torch.ones((1, 5))], 1: []}
    got = pyramid.gather_window(parts, 2, 0, 1)
    assert got[0].shape == (3, 5) and float(got[0][2, 0]) == 2.0 and got[1].shape == (0, 5)


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
