"""Does an image's result depend on what the buffers held before (previous image, allocator garbage)?
    python tools/scratch/poison_probe.py run <order> <out.npz>      (env SHF_POISON_ALLOC optional)
    python tools/scratch/poison_probe.py all  -> spawns the combinations and compares"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def images():
    rng = np.random.default_rng(3)
    out = []
    for i in range(3):
        h, w = 80 + 12 * i, 150 - 9 * i
        out.append(rng.integers(0, 256, (h, w, 3)).astype(np.uint8))
    return out


def run(order, out):
    from smallhardface_amd.config import cfg, cfg_from_file
    from smallhardface_amd import test as T, prototxt as P, caffe, weights
    from tests import helpers as H
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    cfg.TEST.SCALES = [100, 300]
    msg = H.detector_msg(True)
    net = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(net, weights.synth_params(msg, cls_bias=1.0))
    net.set_conv_mode("f16x3")
    ims = images()
    fd = T.FusedDetector(net, n_lanes=4, mode="group")
    dp = T.DevicePyramid(net, n_slots=2)
    res = {}
    queued = []
    for i in [int(c) for c in order] + [None]:
        if i is not None:
            fd.submit(dp.units(ims[i], net=fd.next_head()), 0.05, on_device=True)
            queued.append(i)
        if queued and (i is None or fd.pending() > 1):
            j = queued.pop(0)
            res.setdefault("im%d" % j, []).append(np.asarray(fd.collect()[0]))
    while queued:
        j = queued.pop(0)
        res.setdefault("im%d" % j, []).append(np.asarray(fd.collect()[0]))
    np.savez(out, **{k + "_%d" % n: a for k, v in res.items() for n, a in enumerate(v)})


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
        sys.exit(0)
    outdir = os.path.join(ROOT, "gpurun_out", "poison")
    os.makedirs(outdir, exist_ok=True)
    ref = None
    bad = 0
    for tag, order, env in [("o012", "012", {}), ("o210", "210", {}), ("o0", "0", {}), ("o1", "1", {}), ("o2", "2", {}),
                            ("o1_ff", "1", {"SHF_POISON_ALLOC": "0xff"}), ("o2_ff", "2", {"SHF_POISON_ALLOC": "0xff"}),
                            ("o0_ff", "0", {"SHF_POISON_ALLOC": "0xff"}), ("o012_7f", "012", {"SHF_POISON_ALLOC": "0x7f"}),
                            ("o2_7f", "2", {"SHF_POISON_ALLOC": "0x7f"}), ("o0120", "012012", {}), ("o2_47", "2", {"SHF_POISON_ALLOC": "0x47"})]:
        out = os.path.join(outdir, tag + ".npz")
        r = subprocess.run([sys.executable, __file__, "run", order, out], env=dict(os.environ, PYTHONPATH=ROOT, **env),
                           capture_output=True, text=True)
        if r.returncode:
            print(tag, "FAILED", r.stderr[-800:])
            bad += 1
            continue
        z = np.load(out)
        cur = {k: z[k] for k in z.files}
        if ref is None:
            ref = {k.rsplit("_", 1)[0]: v for k, v in cur.items()}
        for k, v in cur.items():
            base = k.rsplit("_", 1)[0]
            same = v.shape == ref[base].shape and np.array_equal(v, ref[base])
            if not same:
                bad += 1
                n = min(len(v), len(ref[base]))
                d = np.abs(v[:n] - ref[base][:n]).max() if n else -1
                print(tag, k, "DIFFERS: shapes", v.shape, ref[base].shape, "max abs diff of common rows", d)
            else:
                print(tag, k, "same", v.shape)
    print("differences:", bad)
