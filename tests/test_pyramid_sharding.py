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


# ------------------------------------------------------------------------------------------------------------
# pyramid.ShardedDetector's schedule on CPU: a FAKE net (host tensors, rows that encode which unit produced them) under
# the real class, the real gather_window and a real gloo group -- lane sets, chunking into passes of 16, export ->
# all_to_all -> import -> merge on the owner, window k + 1 enqueued before window k is finished, a partly filled last window,
# a rank without a share.  (The same class runs the GPU net in bench.py and in test.pyramid_sharded_inference.)
# ------------------------------------------------------------------------------------------------------------
class _FakeNet(object):
    """The slice of caffe.Net that ShardedDetector touches.  A unit's `data` field carries its id (window, image, unit);
    the rows a unit "detects" are a function of that id alone, so the checker can regenerate them."""

    def __init__(self, log=None):
        self.log = log if log is not None else []
        self.rows = None
        self.imported = []

    def clone(self):
        return _FakeNet(self.log)

    def set_pipeline(self, on=True):
        pass

    def sync(self):
        pass

    @staticmethod
    def rows_of(uid):
        rng = np.random.default_rng(uid)
        n = int(rng.integers(0, 6))
        r = rng.uniform(1, 100, (n, 5)).astype(np.float32)
        r[:, 4] = uid + np.arange(n) / 16.0            # scores: unique, exact in fp32, identify the unit
        return r

    def detect_add_levels(self, members, units, thresh, on_device=False, per_member_lists=False):
        assert on_device and per_member_lists and 1 <= len(units) <= 16 and len(members) == len(units)
        assert len(set(id(m) for m in members)) == len(members) and members[0] is self
        self.log.append(("pass", len(units)))
        for m, u in zip(members, units):
            m.rows = self.rows_of(int(u[0]))

    def detect_export_many(self, members, dst_ptrs, cap_rows):
        import ctypes
        counts = []
        for m, p in zip(members, dst_ptrs):
            r = np.ascontiguousarray(m.rows[:cap_rows])
            if len(r):
                ctypes.memmove(int(p), r.ctypes.data, r.nbytes)
            counts.append(len(m.rows))
        self.log.append(("export", len(members)))
        return counts

    def detect_begin(self):
        self.imported = []

    def detect_import(self, src_ptr, n_rows):
        import ctypes
        a = np.empty((n_rows, 5), np.float32)
        if n_rows:
            ctypes.memmove(a.ctypes.data, int(src_ptr), a.nbytes)
        self.imported.append(a)

    def detect_finish(self, method="BBOX_VOTE", nms_thresh=0.4, cap=None):
        a = np.concatenate(self.imported, 0) if self.imported else np.zeros((0, 5), np.float32)
        return a[np.argsort(-a[:, 4], kind="stable")].astype(np.float64)      # a stand-in "merge": score order


def _uid(window, image, unit):
    return 1000 * window + 20 * image + unit + 1


def _sharded_worker(rank, world, port, q, shard, n_units, n_windows, last_valid):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _FakeNet()
        sd = pyramid.ShardedDetector(net, rank, world, n_units, units_per_level=2, shard=shard, thresh=0.05,
                                     device=torch.device("cpu"), cap_rows=64)
        assert len(sd.lane_sets) == 2 and all(len(ls) == len(sd.mine) for ls in sd.lane_sets)
        out = {}
        prev = None
        for w in range(n_windows):
            n_valid = last_valid if w == n_windows - 1 else world
            picks = [(i, u) for (i, u) in sd.mine if i < n_valid]
            units = [(_uid(w, i, u), 16, 16, 16, 16, 1.0, False) for (i, u) in picks]
            done = sd.submit(units, picks=picks, n_valid=n_valid)
            if w == 0:
                assert done == {}                              # nothing to finish behind the first window
            for i, d in done.items():
                out[(prev, i)] = d
            prev = w
        for i, d in sd.flush().items():
            out[(prev, i)] = d
        assert sd.flush() == {}
        q.put((rank, {k: v for k, v in out.items()}, sd.collectives, [e for e in net.log if e[0] == "pass"]))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,shard,n_units,n_windows,last_valid", [
    (2, "window", 10, 3, 1),        # the bench's shape on two ranks; the last window holds ONE image
    (2, "strict", 18, 2, 2),        # rank 0: 5 levels x 2 flips x 2 images = 20 units = two passes (16 + 4) on two head lanes
    (3, "strict", 4, 2, 3),         # 2 levels on 3 ranks: rank 2 runs nothing and only takes part in the exchange
    (1, "window", 10, 2, 1),        # one rank, no process group: the same class without a collective
])
def test_sharded_detector_schedule_with_a_fake_net(world, shard, n_units, n_windows, last_valid):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, shard, n_units, n_windows, last_valid))
             for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, out, collectives, passes = q.get(timeout=240)
        res[r] = (out, collectives, passes)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        out, collectives, passes = res[r]
        assert collectives == n_windows                     # ONE exchange per window, on every rank alike
        mine = pyramid.my_units(r, world, world, n_units, shard=shard, units_per_level=2)
        if len(mine) > 16:                                  # a share above one member table runs as several passes
            assert ("pass", 16) in passes and ("pass", len(mine) - 16) in passes
        # this rank holds exactly the images it owns, of exactly the windows' valid images
        want_keys = {(w, i) for w in range(n_windows) for i in range(last_valid if w == n_windows - 1 else world)
                     if pyramid.image_owner(i, world) == r}
        assert set(out) == want_keys, (r, sorted(out), sorted(want_keys))
        for (w, i), got in out.items():
            rows = [_FakeNet.rows_of(_uid(w, i, u)) for u in range(n_units)]
            exp = np.concatenate(rows, 0)
            exp = exp[np.argsort(-exp[:, 4], kind="stable")].astype(np.float64)
            assert got.dtype == np.float64 and got.shape == exp.shape, (w, i, got.shape, exp.shape)
            np.testing.assert_array_equal(got, exp)


def test_sharded_detector_refuses_a_mismatched_share():
    import torch
    sd = pyramid.ShardedDetector(_FakeNet(), 0, 1, 4, units_per_level=2, device=torch.device("cpu"), cap_rows=8)
    with pytest.raises(ValueError):
        sd.submit([(1, 16, 16, 16, 16, 1.0, False)] * 3, picks=[(0, 0), (0, 1)])
    with pytest.raises(ValueError):
        sd.submit([(1, 16, 16, 16, 16, 1.0, False)] * 5, picks=[(0, u) for u in range(5)])
