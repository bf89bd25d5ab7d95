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


def gather_window(local, n_images, rank, world, device=None, group=None):
    """Exchange the window's detections: every rank sends each image's rows to the image's OWNER only.

    ``local[i]`` is this rank's (n_i, 5) float32 tensor of detections for window image i
    (possibly empty).  Returns {i: (N_i, 5) tensor} for the images this rank owns, rows
    concatenated in rank order (deterministic).  Uses torch.distributed when world > 1.

    ONE collective per window (all_to_all_single; RCCL on GPUs, gloo in the CPU tests) and one host
    synchronisation, on its result: the block rank s sends for an image is (1 + cap) rows of 5 floats, row 0 =
    [rows in this block, the sender's largest block of the window, 0, 0, 0] (exact in fp32 below 2^24), then the
    rows.  ``cap`` is a grow-only module constant; a sender with more rows than cap sends the first cap of them and
    everybody learns it from row 0 (every rank receives a block from every sender), so all ranks agree -- without
    another collective -- to repeat the exchange once with cap = the next power of two that fits.  An owner thus
    receives world x images-it-owns blocks instead of every rank receiving everything (the round-2 all_gather pair).
    """
    import torch
    if world == 1:
        return {i: local[i] for i in range(n_images)}
    import torch.distributed as dist
    dev = device if device is not None else local[0].device
    out_dev = dev
    if dist.get_backend(group) == "gloo" and torch.device(dev).type != "cpu":
        # gloo (CPU tests / one-GPU validation) exchanges host tensors; RCCL exchanges device tensors
        local = {i: t.cpu() for i, t in local.items()}
        dev = torch.device("cpu")
    per_owner = (n_images + world - 1) // world          # image slots per destination rank (image i -> slot i // world)
    counts = [int(local[i].shape[0]) for i in range(n_images)]   # host-known: shapes of this rank's own tensors
    biggest = max(counts) if counts else 0
    while True:
        cap = _GATHER_CAP["rows"]
        send = torch.zeros((world, per_owner, 1 + cap, 5), dtype=torch.float32, device=dev)
        for i in range(n_images):
            n = min(counts[i], cap)
            blk = send[image_owner(i, world), i // world]
            blk[0, 0] = float(n)
            if n:
                blk[1:1 + n] = local[i][:n]
        send[:, :, 0, 1] = float(biggest)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(world, -1), send.view(world, -1), group=group)
        head = recv[:, :, 0, :2].cpu().numpy()           # the one host synchronisation: [sender][slot] -> (rows, sender's max)
        need = int(head[:, :, 1].max())
        if need <= cap:
            break
        while _GATHER_CAP["rows"] < need:                # every rank sees the same `need`: the same new cap everywhere
            _GATHER_CAP["rows"] *= 2
    out = {}
    for i in range(n_images):
        if image_owner(i, world) != rank:
            continue
        parts = [recv[s, i // world, 1:1 + int(head[s, i // world, 0])] for s in range(world) if head[s, i // world, 0] > 0]
        out[i] = (torch.cat(parts, 0) if parts else torch.zeros((0, 5), dtype=torch.float32, device=dev)).to(out_dev)
    return out


def level_flops(H, W):
    """Algorithmic conv FLOPs of one unit: 2 * 361460 MAC per input pixel."""
    return 2.0 * 361460.0 * H * W
