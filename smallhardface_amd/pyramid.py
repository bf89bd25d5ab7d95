"""Sharding the test pyramid across GPUs (one process per GPU, torch.distributed).

The reference shards by IMAGE RANGE, one forked worker per GPU, and gathers pickled
results through a multiprocessing.Queue (lib/test.py:327-344) -- no data-path exchange.
The north star asks for the pyramid itself to be sharded: the unit of work is one
(image, scale level, flip) forward (lib/test.py:141-155); a window of ``world`` images
is scheduled so that every rank runs exactly one unit of each (level, flip) kind
(perfect balance although unit costs span 9 .. 1433 GFLOP), and the >thresh detections
of an image are gathered on its owner rank, which runs bbox_vote / NMS.  The gather is
the only collective: an all_gather of a count vector followed by an all_gather of
buffers padded to the window's largest contribution (RCCL over xGMI on the GPUs, gloo in
the CPU tests).  Payloads are a few hundred KB: latency-bound, far from the per-link
bandwidth.
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


def gather_window(local, n_images, rank, world, device=None, group=None):
    """Exchange the window's detections.

    ``local[i]`` is this rank's (n_i, 5) float32 tensor of detections for window image i
    (possibly empty).  Returns {i: (N_i, 5) tensor} for the images this rank owns, rows
    concatenated in rank order (deterministic).  Uses torch.distributed when world > 1.
    """
    import torch
    if world == 1:
        return {i: local[i] for i in range(n_images)}
    import torch.distributed as dist
    dev = device if device is not None else local[0].device
    out_dev = dev
    if dist.get_backend(group) == "gloo" and torch.device(dev).type != "cpu":
        # gloo (CPU tests / one-GPU validation) gathers host tensors; RCCL gathers device tensors
        local = {i: t.cpu() for i, t in local.items()}
        dev = torch.device("cpu")
    counts = torch.tensor([int(local[i].shape[0]) for i in range(n_images)], dtype=torch.int64, device=dev)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = torch.stack(all_counts).cpu().numpy()  # [rank][image]
    cap = int(max(1, all_counts.max()))
    buf = torch.zeros((n_images, cap, 5), dtype=torch.float32, device=dev)
    for i in range(n_images):
        n = int(local[i].shape[0])
        if n:
            buf[i, :n] = local[i]
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = {}
    for i in range(n_images):
        if image_owner(i, world) != rank:
            continue
        parts = [bufs[r][i, :int(all_counts[r][i])] for r in range(world) if all_counts[r][i] > 0]
        out[i] = (torch.cat(parts, 0) if parts else torch.zeros((0, 5), dtype=torch.float32, device=dev)).to(out_dev)
    return out


def level_flops(H, W):
    """Algorithmic conv FLOPs of one unit: 2 * 361460 MAC per input pixel."""
    return 2.0 * 361460.0 * H * W
