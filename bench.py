#!/usr/bin/env python3
"""Headline benchmark: images/sec over the full multi-scale test pyramid.

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its N ranks itself through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the same, pre-launched)

One "step" = one window of N images (N = number of GPUs, weak scaling), each image being
the reference's whole test pyramid (configs/smallhardface.toml: scales
[100,300,600,1000,1400] x flip = 10 forward units, lib/test.py:141-155) of a synthetic
1024x1024 source, followed by the >0.05 cut and bbox_vote (lib/test.py:161-175).  The
pyramid blobs are resident in HBM before the timed region.  Units are sharded over the
ranks so that each rank runs one unit of every (level, flip) kind per window; the
detections of an image are sent to its owner rank with ONE RCCL all_to_all per window
(smallhardface_amd/pyramid.py).

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      dominant kernel (split-fp16 MFMA implicit-GEMM conv, or the exact fp32 MFMA one with
                --conv-mode fp32) measured live with HIP events on the runtime's own stream over the timed region
  cpu_baseline  the numpy/OpenBLAS oracle (Caffe's im2col+SGEMM algorithm) timed on the
                host cores on the five levels of one image (flips counted twice) + bbox_vote
  reduced_precision  (N=1) a short second run in bf16 / f16 after the timed region: value + measured drift
  sustained     (N=1) back-to-back steps for --sustain-seconds after the timed region (the timed region is a fraction of a second)
  mixed_shapes  (N=1) 32 uint8 images of 8 WIDER-like shapes through upload + device pyramid + re-plan + two images in flight
                (test.inference_worker's loop), with the runtime's allocation counts after the first pass
  from_files    (N=1) the same shapes as JPEG FILES: decode (reader threads) -> the loop above -> WIDER detection files
HIP events: an untimed survey pass brackets every launch (per-class table); the timed region brackets only the dominant
kernel's launches (an event pair costs the stream a few microseconds; --events-all restores full bracketing).
--force-dist runs the N>1 schedule on a ONE-rank process group (with --backend nccl: a 1-rank RCCL communicator).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix-core peaks (256 CU x 2.4 GHz)
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 (dense; the 5 PF marketing figure is 2:1 sparse)
SRC_H = SRC_W = 1024


def build_units(image_index, src_hw=(SRC_H, SRC_W)):
    """The (data, H, W, im_h, im_w, scale, flip) units of one synthetic source image (10 for the default workload)."""
    from smallhardface_amd.test import pyramid_units
    rng = np.random.default_rng(1000 + image_index)
    im = rng.integers(0, 256, (src_hw[0], src_hw[1], 3)).astype(np.uint8)
    return list(pyramid_units(im))


PMC_FILE = "profiles/r06_pmc.json"   # written by tools/make_profiles.sh (separate rocprofv3 --pmc passes of this command)
LADDER_FILE = "profiles/r06_precision_ladder.json"   # tools/precision_ladder.py: measured drift of the reduced modes


def ladder_drift(mode):
    """'max |dscore| X vs the oracle at C1' for a reduced mode, read from the committed ladder (never a remembered figure)."""
    try:
        v = json.load(open(os.path.join(ROOT, LADDER_FILE)))["modes"][mode]["c1_max_abs_dscore_vs_oracle"]
        return "max |dscore| %.1e vs the oracle at C1 (%s)" % (v, LADDER_FILE)
    except Exception:
        return "drift: see tools/precision_ladder.py"


def committed_pmc(kernel_name):
    """Counter-based figures for ``kernel_name`` from the committed PMC passes (tools/make_profiles.sh): they are NOT
    measured in this run -- the entry carries the kernel-source hash it was taken with and is reported only while
    that still matches the sources (None otherwise)."""
    try:
        tab = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return None
    from tools.kernel_hash import kernel_source_hash
    if tab.get("kernel_source_hash") != kernel_source_hash():
        return {"stale": True}
    return tab.get("kernels", {}).get(kernel_name)


def cpu_baseline(msg, params, seconds_budget=45.0):
    """Oracle (port of Caffe's CPU algorithm: per-image im2col + OpenBLAS SGEMM + separate bias / ReLU / pool passes,
    numpy ProposalLayer, Python-loop bbox_vote) on the host cores, on the bench image's own pyramid: the five levels
    of the workload once each (the flipped units repeat the same shapes and the same work: counted twice), then the
    >0.05 cut and bbox_vote over what they found.  A slow host stops after the levels that fit the budget and scales the
    rest by algorithmic FLOPs (said in 'sample')."""
    from oracle import oracle as O
    from smallhardface_amd.pyramid import level_flops
    from smallhardface_amd.config import cfg
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    onet = O.OracleNet(msg, params=params)
    sides = (112, 304, 608, 1008, 1408)
    img_flops = sum(2 * level_flops(s, s) for s in sides)
    done, tot_fl, tot_dt, dets = [], 0.0, 0.0, []
    for k, side in enumerate(sides):
        rng = np.random.default_rng(7 + side)
        data = (rng.integers(0, 256, (1, 3, side, side)).astype(np.float32) - 115.0)
        scale = side / 1024.0
        onet.blobs['data'].reshape(*data.shape)
        onet.blobs['im_info'].reshape(1, 3)
        t0 = time.perf_counter()
        o = onet.forward(data=data, im_info=np.array([[side - 4, side - 4, scale]], np.float32))
        dt = time.perf_counter() - t0
        d = np.hstack([o["boxes"][:, 1:5] / scale, o["cls_prob"][:, 1:2]]).astype(np.float32)
        dets.append(d[d[:, 4] > 0.05])
        fl = level_flops(side, side)
        done.append("%dx%d" % (side, side))
        tot_fl += fl
        tot_dt += dt
        nxt = sides[k + 1] if k + 1 < len(sides) else None
        if nxt is not None and tot_dt + dt * (level_flops(nxt, nxt) / fl) > seconds_budget:
            break
    t0 = time.perf_counter()
    alld = np.vstack(dets + dets)                       # the flipped units find (mirrored) as much again
    O.bbox_vote(alld, cfg.TEST.NMS_THRESH)
    vote_dt = time.perf_counter() - t0
    whole = len(done) == len(sides)
    img_seconds = (2.0 * tot_dt if whole else tot_dt * img_flops / tot_fl) + vote_dt
    return {"value": 1.0 / img_seconds, "unit": "images/s", "cores": int(threads), "kind": "port", "cpu_model": cpu_model,
            "sample": ("all five pyramid levels of the workload (%s), one forward each through the numpy/OpenBLAS oracle in "
                       "%.2f s -- counted twice for the flipped units, which repeat the same shapes -- plus bbox_vote over "
                       "%d boxes in %.2f s: one whole 10-unit image = %.2f s" % (" + ".join(done), tot_dt, len(alld), vote_dt, img_seconds))
                      if whole else
                      ("pyramid levels %s (%.1f GFLOP of the %.1f GFLOP image) in %.2f s, scaled by algorithmic FLOPs, plus "
                       "bbox_vote %.2f s" % (" + ".join(done), tot_fl / 1e9, img_flops / 1e9, tot_dt, vote_dt)),
            "sample_seconds": tot_dt + vote_dt, "sample_gflops_per_s": tot_fl / tot_dt / 1e9}


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching the HIP runtime (the launcher parent only starts children): the KFD
    topology's nodes with SIMDs (CPU nodes have simd_count 0), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when one is set to a plain index list.  None when the topology cannot be read (the ranks then
    report a shortage themselves: "needs N visible MI355X, found M")."""
    import glob
    n = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0                       # no amdgpu / KFD driver on this host at all
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for p in nodes:
        try:
            for line in open(p):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(",") if t.strip() != ""]
            if all(t.strip().lstrip("-").isdigit() for t in ids):
                n = min(n, len([t for t in ids if int(t) >= 0]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as children of ONE fresh
    `python -m torch.distributed.run` (the reference starts its N workers from one command as well,
    lib/test.py:327-344) and return its exit code.  The parent counts the devices from the KFD topology in sysfs
    (visible_gpu_count: no HIP call, no KFD handle held for the run) and never touches the GPU; no exec.  `--standalone` keeps the rendezvous on a TCPStore
    the agent binds itself (port 0) and hands to the workers: no bind-close-reuse of a port."""
    import subprocess
    if os.environ.get("SHF_BENCH_ONE_GPU") != "1":
        n_dev = visible_gpu_count()
        if n_dev is not None and n_dev < n:
            raise SystemExit("bench.py --gpus %d needs %d visible MI355X, found %d (SHF_BENCH_ONE_GPU=1 --backend gloo runs "
                             "all ranks on one GPU for validation)" % (n, n, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL among processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    for k in ("RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    # the children inherit stdout / stderr: rank 0's JSON line and every rank's diagnostics appear as they are written
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~14 ms: 50 timed steps keep the pipeline fill / drain of the two-image software pipeline below 1 %
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-launch HIP events")
    ap.add_argument("--events-all", action="store_true", help="bracket EVERY launch inside the timed region (default: only the "
                    "dominant kernel's; the per-class table then comes from the untimed survey pass before it)")
    ap.add_argument("--method", default=None, help="BBOX_VOTE (default, the reference's) or NMS")
    ap.add_argument("--conv-mode", default="f16x3", choices=["f16x3", "fp32", "f16x2", "f16", "bf16"],
                    help="f16x3 (headline): split-fp16 MFMA, 3 fp16 products per fp32 product, fp32 accumulate: fp32-class "
                         "accuracy, passes every 1e-4 parity test; fp32: exact v_mfma_f32_32x32x2_f32 everywhere; f16x2 / "
                         "f16 / bf16: the reduced ladder (2 / 1 fp16 products, 1 bf16 product, in EVERY conv kernel) -- NOT "
                         "parity modes, their measured score drift is in " + LADDER_FILE + " (f16x2: %s; f16: %s; bf16: %s)"
                         % (ladder_drift("f16x2"), ladder_drift("f16"), ladder_drift("bf16")))
    ap.add_argument("--no-reduced", action="store_true", help="skip the reduced-precision leg (N=1, headline mode f16x3 only): a "
                    "short second run in bf16 and f16 AFTER and OUTSIDE the timed region, reported as 'reduced_precision'")
    ap.add_argument("--host-input", nargs="?", const="blobs", default=None, choices=["blobs", "image"],
                    help="N=1 only, PCIe-inclusive rates (never the headline value): 'blobs' hands the 10 HOST fp32 "
                         "blobs to the C ABI each step; 'image' uploads the raw uint8 image and builds the pyramid on "
                         "the device (shf_make_pyramid_level)")
    ap.add_argument("--dump-dets", default=None, help="rank 0 writes the detections of window image 0 to this .npy")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "validating the N>1 code path with several ranks on ONE GPU: SHF_BENCH_ONE_GPU=1)")
    ap.add_argument("--lanes", type=int, default=5, help="execution lanes (HIP streams) per GPU in --mode streams")
    ap.add_argument("--mode", default="group", choices=["group", "streams"],
                    help="group: one grid per conv layer over all units of the image; streams: units on HIP streams")
    ap.add_argument("--shard", default="window", choices=["window", "strict"],
                    help="N>1: 'window' = one unit of every (level, flip) kind per rank per window (balanced); "
                         "'strict' = the north star's one-scale-per-GPU form, level l on rank l mod N (SURVEY.md 8e)")
    ap.add_argument("--source", default="1024x1024", help="HxW of the synthetic source images (C4: 768x1024)")
    ap.add_argument("--scales", default=None, help="comma list overriding TEST.SCALES (C4: 300,600,1000,1400)")
    ap.add_argument("--no-flip", action="store_true", help="TEST.FLIP = false")
    ap.add_argument("--no-calib", action="store_true", help="skip the matrix-pipe calibration (N=1, ~0.3 s after the timed region)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-image (un-pipelined) latency leg")
    ap.add_argument("--force-dist", action="store_true",
                    help="N=1: run the N>1 code path on a ONE-rank process group (with --backend nccl: a 1-rank RCCL "
                         "communicator) -- init_process_group, the device-tensor all_to_all_single and detect_import of the "
                         "buffer RCCL produced all execute on a single-GPU box")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="N=1: after the timed region, back-to-back steps for this long in the headline mode -> 'sustained' "
                         "(0 to skip); --steps still defines 'value'")
    ap.add_argument("--overlap-seconds", type=float, default=3.0,
                    help="N=1: a measurement leg after `sustained` -- the same kernels with consecutive images' convolutions "
                         "overlapped on two lane sets / two streams -> 'overlapped_pipeline' (0 to skip; never `value`)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-shape image stream leg ('mixed_shapes', N=1)")
    ap.add_argument("--no-forward-path", action="store_true", help="skip the literal Net.forward() drop-in leg ('net_forward_path', N=1: "
                    "lib/test.py's detect() on the bench image -- host pre-processing, ten Net.forward() with host blobs, C-ABI bbox_vote)")
    ap.add_argument("--no-files", action="store_true", help="skip the file-to-detections leg ('from_files', N=1; part of the "
                    "mixed-shape leg: the same shapes as JPEG files through test.fused_image_loop into the WIDER writer)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (the reference starts its N workers from one
        # command too, lib/test.py:327-344).  This parent never initialises the GPU.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a launcher with WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    one_gpu = os.environ.get("SHF_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0  # validation only: every rank on device 0 (needs --backend gloo)
    n_dev = torch.cuda.device_count()   # (does not initialise the GPU)
    if n_dev < (1 if one_gpu else max(world, 1)):
        raise SystemExit("bench.py --gpus %d needs %d visible MI355X, found %d (SHF_BENCH_ONE_GPU=1 --backend gloo runs "
                         "all ranks on one GPU for validation)" % (args.gpus, world, n_dev))
    torch.cuda.set_device(local_rank)
    dist = None
    dist_path = world > 1 or args.force_dist      # the N>1 schedule (lane sets, export, gather, import); N=1 only with --force-dist
    if dist_path:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store_dir = None
        pg = dict(rank=rank, world_size=world,
                  timeout=datetime.timedelta(seconds=float(os.environ.get("SHF_BENCH_DIST_TIMEOUT", "600"))))
        if world == 1 and "MASTER_PORT" not in os.environ:   # (--force-dist without a launcher: a file store, no port at all)
            import tempfile
            store_dir = tempfile.mkdtemp(prefix="shf_bench_store_")
            pg["init_method"] = "file://" + os.path.join(store_dir, "store")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **pg)
        else:
            dist.init_process_group(args.backend, **pg)
    dev = torch.device("cuda", local_rank)
    ranks_seen = None
    if dist is not None:
        # a 1-element all_reduce of ones: its sum is the number of ranks the collective backend really connected
        one = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(round(float(one.item())))
        if ranks_seen != world:
            raise SystemExit("the %s group connected %d ranks, expected %d" % (args.backend, ranks_seen, world))

    from smallhardface_amd import caffe, prototxt as P, pyramid, weights
    from smallhardface_amd.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    if args.method:
        cfg.TEST.NMS_METHOD = args.method
    if args.scales:
        cfg.TEST.SCALES = [int(v) for v in args.scales.split(",")]
    if args.no_flip:
        cfg.TEST.FLIP = False
    src_hw = tuple(int(v) for v in args.source.lower().split("x"))
    caffe.set_mode_gpu()
    caffe.set_device(local_rank)

    msg = P._add_dimension_reduction(P.build_test_template(True))
    params = weights.synth_params(msg, seed=1234)
    net = caffe.Net(None, prototxt_text=P.dumps(msg))
    for name, blobs in params.items():
        for i, arr in enumerate(blobs):
            net.params[name][i].data[...] = arr
    net.commit_params()
    net.set_conv_mode(args.conv_mode)

    # ---- the window: `world` images, this rank's share of their units resident in HBM
    n_flip = 2 if cfg.TEST.FLIP else 1
    n_units = len(cfg.TEST.SCALES) * n_flip
    mine = pyramid.my_units(rank, world, world, n_units, shard=args.shard, units_per_level=n_flip)
    units = {}
    cache = {}
    for (i, u) in mine:
        if i not in cache:
            cache = {i: build_units(i, src_hw)}
        data, H, W, im_h, im_w, s, flip = cache[i][u]
        t = torch.from_numpy(data).to(dev)
        units[(i, u)] = (t, H, W, im_h, im_w, s, flip)
    del cache
    thresh = 0.05
    last = {}

    fd = sd = None
    if world == 1:
        unit_list = [(units[(0, u)][0].data_ptr(),) + units[(0, u)][1:] for u in range(n_units)]
    if not dist_path:
        from smallhardface_amd.test import FusedDetector
        fd = FusedDetector(net, n_lanes=(n_units if args.mode == 'group' else args.lanes), mode=args.mode)
        lanes = fd.lanes
        if args.host_input == "blobs":
            host_list = [(units[(0, u)][0].cpu().numpy(),) + units[(0, u)][1:] for u in range(n_units)]
        elif args.host_input == "image":
            from smallhardface_amd.test import DevicePyramid
            dp = DevicePyramid(net, n_slots=2)
            host_im = np.random.default_rng(1000).integers(0, 256, (src_hw[0], src_hw[1], 3)).astype(np.uint8)
    else:
        # the N>1 schedule lives in the package (smallhardface_amd/pyramid.py ShardedDetector: two lane sets, grouped
        # passes with per-member lists, export -> ONE all_to_all per window -> import -> merge on the owner, window k + 1's
        # convolutions enqueued before window k's detections travel); this file only feeds it the resident window
        sd = pyramid.ShardedDetector(net, rank, world, n_units, units_per_level=n_flip, shard=args.shard, thresh=thresh,
                                     device=dev, force_collective=args.force_dist)
        lanes = sd.nets
        mine_units = [(units[k][0].data_ptr(),) + units[k][1:] for k in mine]
    host_trace = [] if os.environ.get("SHF_BENCH_HOST_TRACE") == "1" else None   # diagnostics: where the host waits

    def step():
        if not dist_path and args.mode == "streams":  # the older per-unit-streams schedule: one image at a time
            last[0] = fd.detect(unit_list, thresh, on_device=True)[0]
            return
        if not dist_path:
            # two images in flight: image k's box merging / read-back overlaps image k+1's convolutions
            if args.host_input == "blobs":
                fd.submit(host_list, thresh, on_device=False)
            elif args.host_input == "image":
                fd.submit(dp.units(host_im, net=fd.next_head()), thresh, on_device=True)
            else:
                if host_trace is not None:
                    host_trace.append(("submit>", time.perf_counter()))
                fd.submit(unit_list, thresh, on_device=True)
                if host_trace is not None:
                    host_trace.append(("submit<", time.perf_counter()))
            if fd.pending() > 1:
                last[0] = fd.collect()[0]
                if host_trace is not None:
                    host_trace.append(("collect<", time.perf_counter()))
            return
        # this rank's units (from different images) as grouped passes; the window submitted before this one is finished
        # behind them (ShardedDetector.submit)
        last.update(sd.submit(mine_units))

    def fence():
        while fd is not None and fd.pending() > 0:
            last[0] = fd.collect()[0]
        if sd is not None:
            last.update(sd.flush())
        for ln in lanes + (getattr(fd, "_heads", []) if fd is not None else []):
            ln.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed set-up: both pipeline halves (two head lanes / two lane sets) allocate their buffers on first use
    for _ in range(2):
        step()
    fence()
    prof_nets = lanes + (getattr(fd, "_heads", []) if fd is not None else [])

    def read_prof():
        acc = {}
        for ln in prof_nets:
            for k, v in ln.prof_read().items():
                a = acc.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
                for f in a:
                    a[f] += v[f]
        return acc
    # HIP events: an event pair around a launch costs the stream a few microseconds (measured on this workload: 79.5
    # images/s with all ~45 launches of an image bracketed, 82.3 with none).  So: an untimed SURVEY pass with every launch
    # bracketed (the per-class table, and which kernel is the dominant one), then the timed region with only the dominant
    # kernel's launches bracketed -- that is what the roofline needs (--events-all: every launch in the timed region too)
    survey, survey_steps, dominant = None, 0, None
    if not args.no_events:
        for ln in prof_nets:
            ln.prof_enable(True)
        step()                      # (the event pool is created outside everything that is timed)
        fence()
        for ln in prof_nets:
            ln.prof_reset()
        survey_steps = max(4, min(args.steps, 10))
        for _ in range(survey_steps):
            step()
        fence()
        survey = read_prof()
        convs_ = {k: v for k, v in survey.items() if k.startswith("conv_mfma") and v["launches"] > 0}
        if convs_ and not args.events_all:
            dominant = max(convs_.items(), key=lambda kv: kv[1]["ms"])[0]
            for ln in prof_nets:
                ln.prof_only(dominant)
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_events:
        for ln in prof_nets:
            ln.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    headline_dets = np.array(last[0], copy=True) if (world == 1 and 0 in last) else None   # (later legs overwrite `last`)
    if host_trace is not None and dist_path:
        print("host time per window (ms): " + ", ".join("%s %.2f" % (k, 1e3 * v / max(1, sd.collectives)) for k, v in sd.host_seconds.items()), file=sys.stderr)
    if host_trace:
        base = [t for k, t in host_trace if t >= t0][0]
        print(" ".join("%s%.2f" % (k, 1000 * (t - base)) for k, t in host_trace if t >= t0)[:4000], file=sys.stderr)
    prof = {}
    if not args.no_events:
        prof = read_prof()
        for ln in prof_nets:
            ln.prof_enable(False)
            ln.prof_only(None)
    if dist is not None and world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- sustained rate (N=1, headline mode): back-to-back steps for several seconds right after the timed region --
    #      the timed region itself is a quarter of a second on a chip that is power-limited and drifts; `value` stays
    #      what --steps defines.  HIP events are off here (they bracket every launch of the timed region).
    sustained = None
    hwmon = None
    if world == 1 and rank == 0:
        # the card's hwmon files (sclk, socket power), found through the runtime's PCI bus id: sampled from the host side
        # under `sustained` and under the matrix-pipe calibration (smallhardface_amd/telemetry.py); no GPU call
        from smallhardface_amd import _lib, telemetry
        import ctypes as C_
        buf = C_.create_string_buffer(64)
        if _lib.load().shf_device_pci_bus_id(buf, 64) == 0:
            hwmon = telemetry.find_card(buf.value.decode())
    if not dist_path and args.mode == "group" and args.sustain_seconds > 0 and not args.host_input:
        from smallhardface_amd import telemetry
        marks = []
        with telemetry.Sampler(hwmon) as tel:
            t1 = time.perf_counter()
            while True:
                step()
                marks.append(time.perf_counter())
                if marks[-1] - t1 >= args.sustain_seconds:
                    break
            fence()
            t2 = time.perf_counter()
        sustained_tel = tel.summary()
        marks = np.asarray(marks) - t1
        sustained = {"seconds": t2 - t1, "steps": len(marks), "value": len(marks) / (t2 - t1), "unit": "images/s",
                     "first_second": float(np.sum(marks <= 1.0)),
                     "last_second": float(np.sum(marks > marks[-1] - 1.0)),
                     "note": "same resident pyramid, same pipeline as the timed region, per-launch HIP events off; "
                             "first / last second = steps completed in that second",
                     "telemetry": sustained_tel}

    # ---- what overlapping consecutive images' convolutions would add (N=1; a MEASUREMENT leg, never `value`): the same
    #      kernels on two lane sets and two streams (FusedDetector lane_sets=2) fill each other's partial last rounds; every
    #      kernel's duration then contains its neighbour's blocks, which is why the shipped pipeline does not do it
    overlapped = None
    if (not dist_path and args.mode == "group" and args.overlap_seconds > 0 and not args.host_input and
            args.conv_mode == "f16x3"):
        from smallhardface_amd.test import FusedDetector as _FD
        fdo = _FD(net, n_lanes=n_units, mode="group", lane_sets=2)
        last_o = {}
        for _ in range(4):
            fdo.submit(unit_list, thresh, on_device=True)
            if fdo.pending() > 1:
                last_o[0] = fdo.collect()[0]
        while fdo.pending():
            last_o[0] = fdo.collect()[0]
        fence()
        n_o = 0
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < args.overlap_seconds:
            fdo.submit(unit_list, thresh, on_device=True)
            n_o += 1
            if fdo.pending() > 1:
                last_o[0] = fdo.collect()[0]
        while fdo.pending():
            last_o[0] = fdo.collect()[0]
        for ln in fdo.lanes + (fdo._lanes_b or []) + fdo._heads:
            ln.sync()
        torch.cuda.synchronize()
        dt_o = time.perf_counter() - t1
        overlapped = {"value": n_o / dt_o, "unit": "images/s", "seconds": dt_o, "steps": n_o,
                      "vs_value": (n_o / dt_o) / (args.steps / elapsed),
                      "identical_to_headline": bool(headline_dets is not None and 0 in last_o and
                                                    np.array_equal(np.asarray(last_o[0]), headline_dets)),
                      "note": "NOT the shipped pipeline and never `value`: two lane sets, two streams, consecutive images' "
                              "convolutions overlap on the GPU (same kernels, same bits) and fill each other's partial last "
                              "rounds; per-kernel durations then contain the neighbour's blocks, so no roofline is read from it"}
        del fdo

    # ---- single-image latency: ONE image, nothing else in flight, submit -> merged detections on the host
    latency_ms = None
    if not dist_path and args.mode == "group" and not args.no_latency and not args.host_input:
        lat = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            last[0] = fd.detect(unit_list, thresh, on_device=True)[0]
            lat.append(1000.0 * (time.perf_counter() - t1))
        latency_ms = float(np.median(lat[2:]))

    # ---- matrix-pipe calibration (N=1): what a pure MFMA stream of the conv kernels' shape sustains on THIS box with
    # random operands (the chip is power-limited under matrix load and the limit depends on operand toggling), measured
    # right after the timed region while the chip is warm; reported beside the nominal peak, never instead of it
    pipe = None
    if world == 1 and rank == 0 and not args.no_calib and args.conv_mode != "fp32":
        import ctypes as C
        from smallhardface_amd import _lib
        L = _lib.load()
        bf = 1 if args.conv_mode == "bf16" else 0

        from smallhardface_amd import telemetry
        pipe_tel = {}

        def pipe_rate(zero8, const, key):
            # the figure: the first call, as before; then the same stream for ~0.6 s more under the clock / power sampler
            v = C.c_double(0.0)
            _lib.check(L.shf_calib_matrix_pipe(bf, zero8, const, 20000, 6, C.byref(v)))
            first = v.value
            with telemetry.Sampler(hwmon, period_s=0.01) as tel:
                t_end = time.perf_counter() + 0.6
                while time.perf_counter() < t_end:
                    _lib.check(L.shf_calib_matrix_pipe(bf, zero8, const, 20000, 6, C.byref(v)))
            pipe_tel[key] = tel.summary()
            return first
        pipe = {"operands_constant": pipe_rate(0, 1, "operands_constant"), "operands_random": pipe_rate(0, 0, "operands_random"),
                "operands_random_half_of_activations_zero": pipe_rate(4, 0, "operands_random_half_of_activations_zero")}
    # ---- mixed-shape image stream (BASELINE config 4 "WIDER-val-shaped" as a RATE; the reference's hot loop runs images
    #      of different sizes, lib/test.py:239-244): uint8 images of 8 WIDER-like shapes, consecutive images never of the
    #      same shape, through DevicePyramid + FusedDetector.submit / collect exactly as test.inference_worker does --
    #      upload, pyramid on the device, re-plan per shape, two images in flight.  Pass 1 lets every grow-only buffer
    #      reach its largest shape; pass 2 is timed and must not allocate.
    mixed = None
    if not dist_path and args.mode == "group" and not args.no_mixed and not args.host_input and rank == 0:
        from smallhardface_amd.test import DevicePyramid
        from smallhardface_amd.test_utils import pyramid_scales
        shapes = [(768, 1024), (683, 1024), (1024, 732), (1365, 1024), (576, 1024), (1024, 819), (1536, 1024), (1024, 1024)]
        rngm = np.random.default_rng(4242)
        stream = [rngm.integers(0, 256, shapes[k % len(shapes)] + (3,)).astype(np.uint8) for k in range(32)]
        gflop = 0.0
        for im_ in stream:
            for sc_ in pyramid_scales(im_.shape):
                _, _, H_, W_ = caffe.pyramid_level_shape(im_.shape[0], im_.shape[1], sc_, cfg.MAX_RESOLUTION)
                gflop += n_flip * pyramid.level_flops(H_, W_) / 1e9
        dpm = DevicePyramid(net, n_slots=2)

        def run_stream(stream=stream):
            n_boxes = 0
            for im_ in stream:
                fd.submit(dpm.units(im_, net=fd.next_head()), thresh, on_device=True)
                if fd.pending() > 1:
                    n_boxes += len(fd.collect()[0])
            while fd.pending():
                n_boxes += len(fd.collect()[0])
            return n_boxes
        fence()
        a0 = caffe.alloc_counts()
        run_stream()
        fence()
        a1 = caffe.alloc_counts()
        t1 = time.perf_counter()
        n_boxes = run_stream()
        fence()
        dtm = time.perf_counter() - t1
        a2 = caffe.alloc_counts()
        resident_tflops = (sum(pyramid.level_flops(v[1], v[2]) for v in unit_list) / 1e12) * (args.steps / elapsed)
        mixed = {"value": len(stream) / dtm, "unit": "images/s", "images": len(stream),
                 "shapes": ["%dx%d" % hw for hw in shapes], "gflop_per_image_mean": gflop / len(stream),
                 "algorithmic_tflops": gflop / 1e3 / dtm, "resident_algorithmic_tflops": resident_tflops,
                 "flop_normalised_vs_resident": (gflop / 1e3 / dtm) / resident_tflops,
                 "hip_mallocs_first_pass": a1[0] - a0[0], "hip_mallocs_after_first_pass": a2[0] - a1[0],
                 "pinned_host_allocs_after_first_pass": a2[1] - a1[1], "voted_boxes": int(n_boxes),
                 "path": "host uint8 image -> upload (pageable, 2-3 MB) -> shf_make_pyramid_level x 10 -> grouped pass, two images "
                         "in flight (test.inference_worker's loop); consecutive images always differ in shape"}
        # ---- the same stream from image FILES (the reference's hot loop starts at cv2.imread, lib/test.py:113,239-244):
        #      JPEGs of the same 8 shapes in a temp dir -> test.fused_image_loop (decode on reader threads, upload, device
        #      pyramid, grouped pass, two images in flight -- what test.inference_worker runs) -> the WIDER writer
        try:
            from PIL import Image
        except ImportError:       # (no decoder in this interpreter: the leg is skipped, the line says nothing about files)
            Image = None
        if not args.no_files and Image is not None:
            import shutil
            import tempfile
            from smallhardface_amd.datasets import write_detections_wider
            from smallhardface_amd.test import fused_image_loop
            tdir = tempfile.mkdtemp(prefix="shf_bench_files_")
            try:
                os.makedirs(os.path.join(tdir, "images", "0--Bench"))
                rel, nbytes = [], 0
                for k in range(len(stream)):
                    h_, w_ = shapes[k % len(shapes)]
                    # photo-like content (a smooth random field + mild noise): uniform noise would make JPEGs of 1-2 MB that
                    # decode 3x slower than a WIDER photograph of the same size
                    low = rngm.integers(0, 256, (h_ // 16 + 2, w_ // 16 + 2, 3)).astype(np.uint8)
                    im_ = np.asarray(Image.fromarray(low).resize((w_, h_), Image.BICUBIC)).astype(np.int16)
                    im_ = np.clip(im_ + rngm.integers(-12, 13, im_.shape), 0, 255).astype(np.uint8)
                    rel.append("images/0--Bench/img%03d.jpg" % k)
                    Image.fromarray(im_).save(os.path.join(tdir, rel[-1]), quality=90)
                    nbytes += os.path.getsize(os.path.join(tdir, rel[-1]))
                paths = [os.path.join(tdir, r) for r in rel]
                fused_image_loop(net, paths, thresh, fd=fd, dp=dpm)        # untimed: page cache, buffers at these shapes
                fence()
                # the SAME decoded images from memory (the chip is power-limited and the rate depends on the operands: the
                # uniform-noise images of the leg above and these photo-like ones are not comparable with each other)
                from smallhardface_amd.test import _imread
                mem = [_imread(p_) for p_ in paths]
                run_stream(mem)
                fence()
                t1 = time.perf_counter()
                run_stream(mem)
                fence()
                mem_rate = len(mem) / (time.perf_counter() - t1)
                del mem
                # two timed passes, the second is the figure (the first pass with reader threads after the in-memory leg measured
                # 5-10 % lower on some boxes -- a transient of the host side: tools/scratch/files_probe.py); both are reported
                passes = []
                for _ in range(2):
                    st_ = {}
                    t1 = time.perf_counter()
                    dets_f = fused_image_loop(net, paths, thresh, fd=fd, dp=dpm, stats=st_)
                    fence()
                    t2 = time.perf_counter()
                    passes.append(len(paths) / (t2 - t1))
                write_detections_wider(rel, [[], dets_f], os.path.join(tdir, "detections"))
                t3 = time.perf_counter()
                files = {"value": len(paths) / (t3 - t1), "unit": "images/s", "images": len(paths),
                         "value_without_write": len(paths) / (t2 - t1), "first_pass_without_write": passes[0],
                         "same_images_from_memory": mem_rate, "vs_same_images_from_memory": (len(paths) / (t3 - t1)) / mem_rate,
                         "vs_mixed_shapes": (len(paths) / (t3 - t1)) / mixed["value"],
                         "mean_jpeg_kb": nbytes / len(paths) / 1024.0,
                         "decode_ms": st_["decode_ms"], "decode_wait_ms": st_["decode_wait_ms"], "submit_ms": st_["submit_ms"],
                         "collect_wait_ms": st_["collect_wait_ms"], "write_ms": 1000.0 * (t3 - t2) / len(paths),
                         "decode_threads": st_["decode_threads"], "decode_prefetch": st_["decode_prefetch"],
                         "decode_prefetch_adaptive": st_["decode_prefetch_adaptive"], "decode_over_step": st_["decode_over_step"],
                         "voted_boxes": int(sum(len(d) for d in dets_f)),
                         "path": "JPEG files (the mixed_shapes stream's 8 shapes, photo-like content, quality 90) -> PIL decode on "
                                 "reader threads -> test.fused_image_loop (what test.inference_worker runs) -> "
                                 "datasets.write_detections_wider; per-image means in ms: decode on the reader threads, and on "
                                 "the submitting thread decode_wait / submit / collect_wait / write"}
                mixed["from_files"] = files
            finally:
                shutil.rmtree(tdir, ignore_errors=True)
        del dpm

    # ---- reduced-precision leg (BASELINE configs C3 / C5 name bf16; the headline above stays the fp32-class mode): after
    #      and outside the timed region -- throughput of the same image pipeline in the mode, its score drift against the
    #      exact fp32 mode on one mid-size level (every anchor), and how many of the fp32 mode's boxes it reproduces
    reduced = None
    if not dist_path and args.mode == "group" and args.conv_mode == "f16x3" and not args.no_reduced and not args.host_input:
        def level_scores():
            d, H_, W_, im_h, im_w, sc_, _ = units[(0, 4 if n_units > 4 else 0)]
            net.blobs['data'].reshape(1, 3, H_, W_)
            net.blobs['im_info'].reshape(1, 3)
            net.forward(data=d.cpu().numpy(), im_info=np.array([[im_h, im_w, sc_]], np.float32))
            return net.blobs["cls_prob_reshape_output"].data.copy()

        def matched(got, ref, score_tol, box_tol=2.0):
            used, n = np.zeros(len(got), bool), 0
            for row in ref:
                ok = (~used) & (np.abs(got[:, 4] - row[4]) < score_tol) & (np.abs(got[:, :4] - row[:4]).max(axis=1) < box_tol) \
                    if len(got) else np.zeros(0, bool)
                if ok.any():
                    used[int(np.argmax(ok))] = True
                    n += 1
            return n

        headline_last = dict(last)   # (the leg's steps overwrite `last`: --dump-dets and the JSON line report the headline run)
        net.set_conv_mode("fp32")
        ref_scores = level_scores()
        ref_dets = np.asarray(fd.detect(unit_list, thresh, on_device=True)[0])
        legs = {}
        for mode in ("bf16", "f16"):
            net.set_conv_mode(mode)
            drift = float(np.abs(level_scores() - ref_scores).max())
            dets = np.asarray(fd.detect(unit_list, thresh, on_device=True)[0])
            for _ in range(3):
                step()
            fence()
            n_red = 15
            t1 = time.perf_counter()
            for _ in range(n_red):
                step()
            fence()
            dt = time.perf_counter() - t1
            legs[mode] = {"mode": mode, "value": n_red / dt, "unit": "images/s", "steps": n_red,
                          "max_abs_dscore_vs_fp32": drift, "boxes_fp32": int(len(ref_dets)), "boxes": int(len(dets)),
                          "boxes_matched": matched(dets, ref_dets, max(4 * drift, 1e-3))}
        net.set_conv_mode(args.conv_mode)
        last.clear()
        last.update(headline_last)
        reduced = dict(legs["bf16"])
        reduced["products"] = "one v_mfma_f32_32x32x16_bf16 per fp32 product in every conv kernel, fp32 accumulate, fp32 activations in HBM"
        reduced["drift_level"] = "%dx%d level of the workload, every anchor score against conv mode fp32; boxes matched within 2 px" % (
            units[(0, 4 if n_units > 4 else 0)][1], units[(0, 4 if n_units > 4 else 0)][2])
        reduced["also"] = legs["f16"]
        reduced["note"] = "drift-labelled throughput modes, NOT parity modes: the headline value is the f16x3 run above"

    # ---- the LITERAL drop-in path (N=1): what a maintainer gets who only swaps `import caffe` -- lib/test.py:109-178 detect()
    #      on the bench image: host pre-processing (_get_image_blob), ten forward_net() = ten Net.forward() with HOST blobs
    #      in and out (lib/test.py:21-106), the > 0.05 cut on the host, bbox_vote through the C ABI.  After and outside the
    #      timed region; every other leg times the device-resident loop (test.fused_image_loop / FusedDetector).
    nf_path = None
    if not dist_path and args.mode == "group" and not args.no_forward_path and not args.host_input and rank == 0:
        from smallhardface_amd import test as T
        from smallhardface_amd.test_utils import _get_image_blob, _get_image_blob_device, pyramid_scales
        im0 = np.random.default_rng(1000).integers(0, 256, (src_hw[0], src_hw[1], 3)).astype(np.uint8)   # = build_units(0)
        fence()
        T.detect(net, None, thresh, pyramid=True, im=im0)                  # untimed: buffers of the root net at these shapes
        t1 = time.perf_counter()
        _get_image_blob_device(im0, pyramid_scales(im0.shape))       # what detect() calls (C ABI shf_image_blobs)
        pre_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        _get_image_blob(im0, pyramid_scales(im0.shape))              # the numpy mirror of the same step (SHF_HOST_PREPROCESS=1)
        pre_host_s = time.perf_counter() - t1
        n_nf = 3
        net.timing = {}
        net.prof_enable(True)
        net.prof_reset()
        walls, det_t, misc_t = [], 0.0, 0.0
        for _ in range(n_nf):
            t1 = time.perf_counter()
            dets_nf, tm_nf = T.detect(net, None, thresh, pyramid=True, im=im0)
            walls.append(time.perf_counter() - t1)
            det_t += tm_nf['detect'].total_time
            misc_t += tm_nf['misc'].total_time
        pr = net.prof_read()
        net.prof_enable(False)
        tm = net.timing
        net.timing = None
        per = lambda v: 1000.0 * v / n_nf
        wall = float(np.median(walls))
        dev_ms = {k: v["ms"] / n_nf for k, v in pr.items() if v["launches"] > 0}
        fused_ref = np.asarray(headline_dets) if headline_dets is not None else None
        got = np.asarray(dets_nf[0], dtype=np.float64)
        nf_path = {
            "value": 1.0 / wall, "unit": "images/s", "ms_per_image": 1000.0 * wall, "images": n_nf,
            "vs_fused_rate": (1.0 / wall) / (args.steps / elapsed),
            "preprocess_ms": 1000.0 * pre_s, "preprocess_numpy_mirror_ms": 1000.0 * pre_host_s,
            "forward_calls_per_image": tm.get("calls", 0) / n_nf,
            "input_copy_ms": per(tm.get("input_copy_s", 0.0)), "forward_call_ms": per(tm.get("forward_call_s", 0.0)),
            "output_read_ms": per(tm.get("output_read_s", 0.0)),
            "h2d_ms": dev_ms.get("h2d_copy", 0.0), "d2h_ms": dev_ms.get("d2h_copy", 0.0),
            "forward_ms": sum(v for k, v in dev_ms.items() if k not in ("h2d_copy", "d2h_copy", "box_merge")),
            "merge_ms": 1000.0 * misc_t / n_nf, "detect_timer_ms": 1000.0 * det_t / n_nf,
            "boxes": int(len(got)),
            "identical_to_fused_path": bool(fused_ref is not None and fused_ref.shape == got.shape and np.array_equal(fused_ref, got)),
            "kernel_ms_per_image": {k: round(v, 3) for k, v in dev_ms.items()},
            "path": "test.detect(net, im=...) as lib/test.py:109-178: _get_image_blob with host blobs out (preprocess_ms: the five "
                    "levels through shf_image_blobs, where the reference calls cv2.resize; the bit-equal numpy mirror would take "
                    "preprocess_numpy_mirror_ms), ten forward_net() -> Net.forward() with host blobs "
                    "(input_copy_ms: np.pad + the copy into the blob's pinned mirror; forward_call_ms: shf_net_forward = H2D + "
                    "kernels + count read-back, on the device h2d_ms / forward_ms / d2h_ms by HIP events; output_read_ms: "
                    "Blob.data of boxes / cls_prob), > 0.05 cut on the host, bbox_vote through the C ABI (merge_ms); f16x3: "
                    "Net.forward() runs the fused path's kernels (fused first pair, pools in the epilogues)",
        }

    if rank == 0 and args.dump_dets and 0 in last:
        np.save(args.dump_dets, np.asarray(last[0]))
    if rank == 0:
        images = world * args.steps
        value = images / elapsed
        lv = [(v[1], v[2]) for v in (unit_list if world == 1 else build_units(0, src_hw))]
        gflop_image = sum(pyramid.level_flops(h, w) for h, w in lv) / 1e9
        default_wl = (src_hw == (1024, 1024) and list(cfg.TEST.SCALES) == [100, 300, 600, 1000, 1400] and cfg.TEST.FLIP)
        out = {
            "metric": "images_per_sec_full_multiscale_pyramid", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f16x3": "f32 via split-f16 MFMA (3x fp16 products, f32 accumulate)", "fp32": "f32",
                      "f16x2": "f16 activations x split-f16 weights (2 fp16 products, f32 accumulate): drift-labelled, " + ladder_drift("f16x2"),
                      "f16": "f16 operands (1 product, f32 accumulate): drift-labelled, " + ladder_drift("f16"),
                      "bf16": "bf16 operands (1 product, f32 accumulate): drift-labelled, " + ladder_drift("bf16")}[args.conv_mode],
            "data": "synthetic",
            "config": {
                "workload": ("C5: full smallhardface.toml test pyramid" if default_wl else "test pyramid") +
                            " of a %dx%d (HxW) source: scales %s -> padded levels %s%s = %d units/image (%.1f GFLOP), "
                            "VGG-16 + shared-weight dilated heads (different_dilation + dim_red), fp32 activations, conv "
                            "mode %s, proposal tail + >0.05 cut + %s on device"
                            % (src_hw[0], src_hw[1], list(cfg.TEST.SCALES),
                               "/".join("%dx%d" % hw for hw in lv[::n_flip]), " x flip" if cfg.TEST.FLIP else "",
                               n_units, gflop_image, args.conv_mode, cfg.TEST.NMS_METHOD),
                "images_per_step": world, "units_per_image": n_units, "lanes_per_gpu": len(lanes),
                "unit_execution": args.mode, "shard": args.shard if dist_path else None,
                "parallelism": ("single GPU" if not dist_path else
                                ("pyramid units sharded 1-of-each-kind per GPU per window" if args.shard == "window" else
                                 "strict one-scale-per-GPU: level l (all its flips, every image of the window) on rank "
                                 "l mod N") + "; one all_to_all of detections per window, each image's rows to its owner rank, over %s"
                                % ("RCCL" if args.backend == "nccl" else args.backend)),
                "weights": "seeded synthetic (no trained caffemodel exists in the reference tree)",
                "detections_last_image": int(len(next(iter(last.values())))) if last else 0,
            },
        }
        if dist_path:
            out["rccl_ranks"] = int(ranks_seen) if args.backend == "nccl" else 0   # (sum of a 1-element all_reduce of ones)
            out["collective_ranks"] = int(ranks_seen)
            out["collective_backend"] = str(dist.get_backend())
            out["collectives_issued_rank0"] = int(sd.collectives)
            out["all_ranks_on_one_gpu"] = bool(one_gpu)
        if latency_ms is not None:
            out["latency_ms"] = latency_ms
            out["latency_note"] = "one image, no pipelining: submit of the 10-unit grouped pass -> merged boxes on the host"
        # ---- roofline of the dominant kernel -----------------------------------------
        # the per-class table: from the timed region when every launch was bracketed there (--events-all), otherwise from the
        # survey pass; the DOMINANT kernel's figures always come from the timed region's own events
        table, table_steps = (prof, args.steps) if (dominant is None or survey is None) else (survey, survey_steps)
        convs = {k: v for k, v in table.items() if k.startswith("conv_mfma") and v["launches"] > 0}
        if convs and (dominant is None or prof.get(dominant, {}).get("launches", 0) > 0):
            name = dominant if dominant is not None else max(convs.items(), key=lambda kv: kv[1]["ms"])[0]
            dom = prof[name]
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            all_ms = sum(v["ms"] for v in table.values())
            split = "f16x3" in name
            nprod = {"f16x3": 3.0, "f16x2": 2.0, "f16": 1.0, "bf16": 1.0}.get(args.conv_mode, 1.0) if split else 1.0
            peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
            pmc = committed_pmc(name)
            stale = bool(pmc and pmc.get("stale"))
            pmc = None if (pmc is None or stale) else pmc
            out["roofline"] = {
                "bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak,
                "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                "traffic_measured_in_this_run": False,
                "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, bytes = "
                                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch: gfx950 FETCH_SIZE counts 128-B requests "
                                   "as 64 B, MI355X_MICROARCH.md HBM section); taken on the kernel sources of this tree "
                                   "(hash checked)" % PMC_FILE) if pmc else
                                  ("%s is stale (kernel sources changed since the PMC passes): not reported" % PMC_FILE
                                   if stale else None),
                "mfma_dtype": ({"f16x3": "fp16 x3 (split-fp16: 3 MFMA FLOPs issued per algorithmic FLOP)",
                                "f16x2": "fp16 x2 (2 MFMA FLOPs issued per algorithmic FLOP)", "f16": "fp16", "bf16": "bf16"}[args.conv_mode]
                               if split else "fp32"),
                "frac_issued": ach * nprod / peak,
                "issued_mfma_achieved": ach * nprod,
                "mfma_busy": pmc.get("mfma_busy") if pmc else None,
                "mfma_busy_source": ("%s: SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-derived kernel cycles x 4 SIMDs x CUs), "
                                     "not measured in this run" % PMC_FILE) if pmc and pmc.get("mfma_busy") is not None else None,
                "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                "algorithmic_gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                "all_conv_mfma_achieved": sum(v["flops"] for v in convs.values()) /
                                          (sum(v["ms"] for v in convs.values()) * 1e-3) / 1e12,
                "kernel_ms_share": {k: round(v["ms"] / all_ms, 4) for k, v in table.items() if v["ms"] > 0},
                "kernel_ms_per_image": {k: round(v["ms"] / table_steps, 3) for k, v in table.items() if v["ms"] > 0},
                "launches_per_image": {k: round(v["launches"] / table_steps, 2) for k, v in table.items() if v["launches"] > 0},
                "events": ("timed region: HIP events around the dominant kernel's launches only (what `achieved`, `avg_launch_ms` are "
                           "from); per-class table and all_conv_mfma_achieved: untimed survey pass of %d steps right before it, every "
                           "launch bracketed (dominant kernel there: %.4f ms per launch)"
                           % (survey_steps, survey[name]["ms"] / max(1, survey[name]["launches"])))
                          if table is survey else "timed region: HIP events around every launch",
            }
        if pipe is not None and "roofline" in out and out["roofline"]["peak"] == PEAK_F16_MFMA_TFLOPS:
            r = out["roofline"]
            ceil_ = pipe["operands_random_half_of_activations_zero"]
            r["matrix_pipe_sustained"] = {
                "unit": "TFLOP/s issued", **{k: round(v, 1) for k, v in pipe.items()},
                "frac_of_peak": {k: round(v / r["peak"], 4) for k, v in pipe.items()},
                "what": "shf_calib_matrix_pipe (csrc/calib.hip), measured in this run after the timed region: a pure stream of "
                        "v_mfma_f32_32x32x16_%s with the conv kernels' register diet (1 wave/SIMD, 8 accumulator tiles, 3 products "
                        "per fragment pair), no memory traffic; the clock drops with the operands' bit toggling (power limit), so "
                        "the nominal peak is only reached with constant operands" % ("bf16" if args.conv_mode == "bf16" else "f16"),
                "telemetry": {k: ({kk: v[kk] for kk in ("sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "power_w_max", "power_cap_w", "samples")}
                                  if v else None) for k, v in pipe_tel.items()},
                "frac_issued_of_sustained": round(r["issued_mfma_achieved"] / ceil_, 4),
                "frac_issued_of_sustained_note": "issued MFMA rate of the dominant kernel / the half-zero-activation row (the "
                                                 "activations this workload's convolutions read are 41-54 % zeros from conv3_2 on: "
                                                 "tools/diag_zero_fraction.py, profiles/r03_mfma_power.txt)",
            }
        if sustained is not None:
            out["sustained"] = sustained
        if overlapped is not None:
            out["overlapped_pipeline"] = overlapped
        if mixed is not None:
            from_files = mixed.pop("from_files", None)
            out["mixed_shapes"] = mixed
            if from_files is not None:
                out["from_files"] = from_files
        if reduced is not None:
            out["reduced_precision"] = reduced
        if nf_path is not None:
            out["net_forward_path"] = nf_path
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(msg, params)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if store_dir:
            import shutil
            shutil.rmtree(store_dir, ignore_errors=True)


if __name__ == "__main__":
    main()
