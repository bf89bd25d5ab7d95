"""Sharding the test pyramid across GPUs (one process per GPU, torch.distributed).

The reference shards by IMAGE RANGE, one forked worker per GPU, and gathers pickled
results through a multiprocessing.Queue (lib/test.py:327-344) -- no data-path exchange.
The north star asks for the pyramid itself to be sharded: the unit of work is one
(image, scale level, flip) forward (lib/test.py:141-155); a window of ``world`` images
is scheduled so that every rank runs exactly one unit of each (level, flip) kind
(perfect balance although unit costs span 9 .. 1433 GFLOP), and the >thresh detections
of an image are gathered on its owner rank, which runs bbox_vote / NMS.  The gather is
the only collective: ONE all_to_all of fixed-capacity blocks per window, each image's rows
going to its owner only (RCCL over xGMI on the GPUs, gloo in the CPU tests).  Payloads
are a few hundred KB: latency-bound, far from the per-link bandwidth.
"""
import numpy as np


def unit_rank(image, unit, world):
    """Rank that runs unit ``unit`` (level*2+flip order of detect()) of window image ``image``."""
    return (unit + image) % world


def image_owner(image, world):
    """Rank that merges (bbox_vote / NMS) window image ``image``."""
    return image % world


def strict_level_rank(level, world):
    """The north star's plain mode: pyramid level -> GPU (both flips of a level together)."""
    return level % world


def my_units(rank, world, n_images, n_units, shard="window", units_per_level=2):
    """[(image, unit)] this rank runs for a window of ``n_images`` images, image-major.

    shard "window": unit u of window image i on rank (u + i) mod world -- one unit of every kind per rank.
    shard "strict": the north star's one-scale-per-GPU form -- every unit of pyramid level l (its
    ``units_per_level`` flips, of every image of the window) on rank l mod world; with fewer ranks than levels a
    rank holds several levels, with more ranks than levels the surplus ranks only take part in the gather."""
    if shard == "strict":
        return [(i, u) for i in range(n_images) for u in range(n_units)
                if strict_level_rank(u // max(1, units_per_level), world) == rank]
    if shard != "window":
        raise ValueError("shard must be 'window' or 'strict'")
    return [(i, u) for i in range(n_images) for u in range(n_units) if unit_rank(i, u, world) == rank]


# rows a (sender, image) block can carry; grow-only, the same on every rank (see gather_window)
_GATHER_CAP = {"rows": 4096}
# the exchange's blocks, allocated once per (device, world, slots, cap) and reused by every window (two sets, used in
# turn: a caller that pipelines windows may still hold views of the previous window's receive block)
_GATHER_BUFS = {}


def _gather_buffers(dev, world, per_owner, cap):
    import torch
    key = (str(dev), world, per_owner, cap)
    ent = _GATHER_BUFS.get(key)
    if ent is None:
        for k in [k for k in _GATHER_BUFS if k[:3] == key[:3]]:       # a smaller cap of the same geometry: drop it
            del _GATHER_BUFS[k]
        pin = torch.device(dev).type == "cuda"
        ent = {"turn": 0, "sets": [
            {"send": torch.zeros((world, per_owner, 1 + cap, 5), dtype=torch.float32, device=dev),
             "recv": torch.zeros((world, per_owner, 1 + cap, 5), dtype=torch.float32, device=dev),
             "hdr": torch.zeros((world, per_owner, 5), dtype=torch.float32, pin_memory=pin),
             "head": torch.zeros((world, per_owner, 2), dtype=torch.float32, pin_memory=pin)} for _ in range(2)]}
        _GATHER_BUFS[key] = ent
    ent["turn"] ^= 1
    return ent["sets"][ent["turn"]]


def gather_window(local, n_images, rank, world, device=None, group=None, force_collective=False):
    """Exchange the window's detections: every rank sends each image's rows to the image's OWNER only.

    ``local[i]`` is this rank's (n_i, 5) float32 tensor of detections for window image i (possibly empty), or a LIST
    of such tensors (the per-unit export buffers: copied straight into the send block, no concatenation first).
    Returns {i: (N_i, 5) tensor} for the images this rank owns, rows concatenated in rank order (deterministic).
    Uses torch.distributed when world > 1 -- or with ``force_collective`` on a 1-rank group: the same code path, which
    is how the RCCL branch is executed on a single-GPU box (tests/test_gpu_fullsize.py, bench.py --force-dist).

    ONE collective per window (all_to_all_single; RCCL on GPUs, gloo in the CPU tests) and one host
    synchronisation, on its result: the block rank s sends for an image is (1 + cap) rows of 5 floats, row 0 =
    [rows in this block, the sender's largest block of the window, 0, 0, 0] (exact in fp32 below 2^24), then the
    rows.  ``cap`` is a grow-only module constant; a sender with more rows than cap sends the first cap of them and
    everybody learns it from row 0 (every rank receives a block from every sender), so all ranks agree -- without
    another collective -- to repeat the exchange once with cap = the next power of two that fits.  An owner thus
    receives world x images-it-owns blocks instead of every rank receiving everything (the round-2 all_gather pair).

    Host side (round 4): the send / receive blocks are allocated once and reused; all header rows of a window go up
    as ONE host-built tensor copy (they were ~3 scalar device writes per image), rows beyond a block's count are never
    read, so nothing is cleared; the header rows come back through one pinned copy.

    LIFETIME of the result: an image that exactly one sender contributed to is returned as a VIEW into the cached
    receive block (no copy); two buffer sets alternate, so a returned tensor stays valid through the NEXT call and is
    overwritten by the one after it.  A caller that keeps results longer must ``.clone()`` them (bench.py imports them
    into the owner's image list before its next-but-one window).  With world == 1 and no collective the inputs
    themselves are returned.

    ``device``: where the exchange's blocks live; when None it is taken from the first non-empty part of ``local`` --
    a rank whose parts are ALL empty lists must therefore pass it (ValueError otherwise).
    """
    import torch

    def parts_of(i):
        t = local[i]
        return [p for p in (t if isinstance(t, (list, tuple)) else [t]) if p.shape[0] > 0]

    if world == 1 and not force_collective:
        out = {}
        for i in range(n_images):
            ps = parts_of(i)
            if len(ps) == 1:
                out[i] = ps[0]
                continue
            if device is not None:
                d0 = device
            else:
                seen = [p for j in range(n_images)
                        for p in (local[j] if isinstance(local[j], (list, tuple)) else [local[j]])]
                d0 = seen[0].device if seen else "cpu"
            out[i] = torch.cat(ps, 0) if ps else torch.zeros((0, 5), dtype=torch.float32, device=d0)
        return out
    import torch.distributed as dist
    def first_tensor():
        for i in range(n_images):
            t = local[i]
            for p in (t if isinstance(t, (list, tuple)) else [t]):
                return p
        return None

    if device is not None:
        dev = device
    else:
        t0 = first_tensor()
        if t0 is None:
            raise ValueError("gather_window: every local part is an empty list; pass device=")
        dev = t0.device
    out_dev = dev
    to_host = dist.get_backend(group) == "gloo" and torch.device(dev).type != "cpu"
    if to_host:
        # gloo (CPU tests / one-GPU validation) exchanges host tensors; RCCL exchanges device tensors
        dev = torch.device("cpu")
    per_owner = (n_images + world - 1) // world          # image slots per destination rank (image i -> slot i // world)
    parts = [parts_of(i) for i in range(n_images)]
    counts = [sum(int(p.shape[0]) for p in ps) for ps in parts]   # host-known: shapes of this rank's own tensors
    biggest = max(counts) if counts else 0
    while True:
        cap = _GATHER_CAP["rows"]
        buf = _gather_buffers(dev, world, per_owner, cap)
        send, recv, hdr = buf["send"], buf["recv"], buf["hdr"]
        hdr.zero_()
        hdr[:, :, 1] = float(biggest)
        for i in range(n_images):
            n = min(counts[i], cap)
            hdr[image_owner(i, world), i // world, 0] = float(n)
            blk, at = send[image_owner(i, world), i // world], 1
            for p_ in parts[i]:
                take = min(int(p_.shape[0]), 1 + n - at)
                if take <= 0:
                    break
                blk[at:at + take].copy_(p_[:take], non_blocking=True)     # (device -> host for gloo-on-GPU validation)
                at += take
        send[:, :, 0, :].copy_(hdr, non_blocking=True)   # every header row of the window in one copy
        if to_host:
            # (validation path: the rows came down from the GPU on torch's current stream -- wait for THAT stream, not for the
            # device: the next window's convolutions are already running on the runtime's own streams)
            torch.cuda.current_stream().synchronize()
        dist.all_to_all_single(recv.view(world, -1), send.view(world, -1), group=group)
        buf["head"].copy_(recv[:, :, 0, :2])             # the one host synchronisation: [sender][slot] -> (rows, sender's max)
        if torch.device(dev).type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        head = buf["head"].numpy()
        need = int(head[:, :, 1].max())
        if need <= cap:
            break
        while _GATHER_CAP["rows"] < need:                # every rank sees the same `need`: the same new cap everywhere
            _GATHER_CAP["rows"] *= 2
    out = {}
    for i in range(n_images):
        if image_owner(i, world) != rank:
            continue
        got = [recv[s, i // world, 1:1 + int(head[s, i // world, 0])] for s in range(world) if head[s, i // world, 0] > 0]
        out[i] = (got[0] if len(got) == 1 else
                  (torch.cat(got, 0) if got else torch.zeros((0, 5), dtype=torch.float32, device=dev))).to(out_dev)
    return out


GROUP_UNITS = 16   # units per grouped pass (one kernel-argument member table: csrc/conv_common.h MAX_GROUP)


class ShardedDetector(object):
    """One rank's half of the pyramid-sharded schedule (north star / SURVEY.md §8e; the reference's analogue is the
    image-range split with a Queue gather, lib/test.py:327-344).

    A WINDOW is ``world`` images.  ``self.mine`` = the (image-in-window, unit) pairs this rank runs (``my_units``:
    shard "window" or "strict").  ``submit(units)`` enqueues this rank's units of the next window as grouped passes of
    at most GROUP_UNITS units -- every lane keeps the detections of ITS unit (per-member lists) -- on one of TWO lane
    sets, and then finishes the window submitted before it: export of every unit's rows, ONE all_to_all to the images'
    owner ranks (``gather_window``), import into the root net's image list and bbox_vote / NMS there.  So window k + 1's
    convolutions are already queued while window k's detections travel and merge: the exchange hides under compute.
    ``flush()`` finishes what is still pending.  Both return {image-in-window: (n, 5) float64 detections} for the
    images this rank OWNS (``image_owner``).

    Streams: both windows' convolutions share ONE in-order stream, tails and exports run on the set heads' own
    high-priority streams, the merges on the root net's (shf_net_set_pipeline) -- independent of how the runtime maps
    streams to hardware queues.  The root net is in neither lane set: its image list is where gathered rows merge.

    Every rank must make the same sequence of submit / flush calls (each finished window is one collective), also a
    rank whose share is empty (strict sharding with more ranks than levels)."""

    def __init__(self, net, rank, world, n_units, units_per_level=2, shard="window", thresh=0.05, device=None,
                 group=None, force_collective=False, cap_rows=None):
        import torch
        from .config import cfg
        self.net, self.rank, self.world = net, int(rank), int(world)
        self.shard, self.thresh, self.group = shard, float(thresh), group
        self.force_collective = bool(force_collective)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.cap_rows = int(cap_rows or cfg.TEST.N_DETS_PER_MODULE)
        self.mine = my_units(self.rank, self.world, self.world, n_units, shard=shard, units_per_level=units_per_level)
        n = len(self.mine)
        self.chunks = [(a, min(a + GROUP_UNITS, n)) for a in range(0, n, GROUP_UNITS)]
        self.lane_sets = [[net.clone() for _ in range(n)] for _ in range(2)]
        for h in [ls[a] for ls in self.lane_sets for (a, _) in self.chunks] + [net]:
            h.set_pipeline(True)
        self.export_sets = [[torch.empty((self.cap_rows, 5), dtype=torch.float32, device=self.device) for _ in range(n)]
                            for _ in range(2)]
        self._k = 0
        self._pending = None      # (lane set, [(image, unit)] of that window, n_valid images)
        self.collectives = 0
        self.host_seconds = {"enqueue": 0.0, "export": 0.0, "gather": 0.0, "merge": 0.0}

    @property
    def nets(self):
        """Every net / lane that launches kernels for this detector (profiling, synchronisation)."""
        return [self.net] + self.lane_sets[0] + self.lane_sets[1]

    def submit(self, units, picks=None, n_valid=None):
        """Enqueue this rank's units of the next window; finish and return the previous one ({} for the first).

        ``units``: (data, H, W, im_h, im_w, scale, flip) tuples with DEVICE pointers, aligned with ``picks`` (default
        ``self.mine``; a last, partly filled window passes the pairs it has, in ``self.mine`` order).  ``n_valid``: images
        the window really holds (default ``world``); owners of the others get nothing back."""
        import time
        picks = self.mine if picks is None else list(picks)
        units = list(units)
        if len(units) != len(picks) or len(picks) > len(self.mine):
            raise ValueError("ShardedDetector.submit: %d units for %d picks (this rank's share is %d)"
                             % (len(units), len(picks), len(self.mine)))
        w = self._k & 1
        self._k += 1
        ls = self.lane_sets[w]
        t0 = time.perf_counter()
        for a in range(0, len(picks), GROUP_UNITS):
            b = min(a + GROUP_UNITS, len(picks))
            ls[a].detect_add_levels(ls[a:b], units[a:b], self.thresh, on_device=True, per_member_lists=True)
        self.host_seconds["enqueue"] += time.perf_counter() - t0
        done = self._finish() if self._pending is not None else {}
        self._pending = (w, picks, self.world if n_valid is None else int(n_valid))
        return done

    def flush(self):
        """Finish the pending window (one collective) and return its detections; {} when nothing is pending."""
        if self._pending is None:
            return {}
        done = self._finish()
        self._pending = None
        return done

    def _finish(self):
        import time
        from .config import cfg
        w, picks, n_valid = self._pending
        ls, ex = self.lane_sets[w], self.export_sets[w]
        t_a = time.perf_counter()
        counts = []
        for a in range(0, len(picks), GROUP_UNITS):       # (each pass was enqueued on its own head lane ls[a])
            b = min(a + GROUP_UNITS, len(picks))
            counts += ls[a].detect_export_many(ls[a:b], [e.data_ptr() for e in ex[a:b]], self.cap_rows)
        t_b = time.perf_counter()
        local = {i: [] for i in range(self.world)}        # per image: the units' export buffers as they are
        for m, (i, _) in enumerate(picks):
            if counts[m]:
                local[i].append(ex[m][:min(counts[m], self.cap_rows)])
        got = gather_window(local, self.world, self.rank, self.world, device=self.device, group=self.group,
                            force_collective=self.force_collective)
        self.collectives += 1
        t_c = time.perf_counter()
        out = {}
        for i, t in got.items():                          # (gather_window has synchronised on the received header rows)
            if i >= n_valid:
                continue
            self.net.detect_begin()
            t = t.contiguous()
            self.net.detect_import(t.data_ptr(), int(t.shape[0]))
            out[i] = self.net.detect_finish(cfg.TEST.NMS_METHOD, cfg.TEST.NMS_THRESH)
        t_d = time.perf_counter()
        self.host_seconds["export"] += t_b - t_a
        self.host_seconds["gather"] += t_c - t_b
        self.host_seconds["merge"] += t_d - t_c
        return out

    def sync(self):
        for ln in self.nets:
            ln.sync()


def level_flops(H, W):
    """Algorithmic conv FLOPs of one unit: 2 * 361460 MAC per input pixel."""
    return 2.0 * 361460.0 * H * W
