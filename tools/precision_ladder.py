"""Measured drift of the reduced-precision ladder (run on an MI355X; writes gpurun_out/precision_ladder.json).

    python tools/precision_ladder.py [--quick]

For each conv mode -- f16x3 (three fp16 products: the parity mode), f16x2 (activations act as fp16), f16 (plain fp16
operands), bf16 (plain bf16 operands) -- and for single layers switched to 2 / 1 products inside the f16x3 mode:
  * C1 (512x512 level): max |score - oracle| over the 12 288 anchors (the 1e-4 bar), max |delta - oracle|
  * 1008x1008 level: max |score - fp32 mode| over the 47 628 anchors
  * C5 image (10 units): voted detections against the fp32 mode: count, max |dscore|, rows whose written integer
    coordinates differ; images/s of the two-image pipeline
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    quick = "--quick" in sys.argv
    from oracle import oracle as O
    from smallhardface_amd import caffe, prototxt as P, test as T, weights
    from smallhardface_amd.config import cfg, cfg_from_file
    from tests import helpers as H
    from tests.test_gpu_fullsize import match_detections, written
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    msg = P._add_dimension_reduction(P.build_test_template(True))
    params = weights.synth_params(msg, seed=1234)
    net = caffe.Net(None, prototxt_text=P.dumps(msg))
    H.load_params(net, params)
    onet = O.OracleNet(msg, params=params)
    out = {"modes": {}, "single_layer": {}}

    def level_scores(data, info):
        net.blobs['data'].reshape(*data.shape)
        net.blobs['im_info'].reshape(1, 3)
        net.forward(data=data, im_info=info)
        return net.blobs["cls_prob_reshape_output"].data.copy(), net.blobs["bbox_pred_output"].data.copy()

    d512 = H.synth_image_blob(512, 512, seed=21)
    i512 = np.array([[512, 512, 1.0]], np.float32)
    onet.blobs['data'].reshape(*d512.shape)
    onet.blobs['im_info'].reshape(1, 3)
    onet.forward(data=d512, im_info=i512)
    o_sc, o_dl = onet.blobs["cls_prob_reshape_output"].data.copy(), onet.blobs["bbox_pred_output"].data.copy()
    d1008 = H.synth_image_blob(1008, 1008, seed=5)
    i1008 = np.array([[1000, 1000, 0.9765625]], np.float32)
    net.set_conv_mode("fp32")
    ref1008, _ = level_scores(d1008, i1008)

    im = np.random.default_rng(1000).integers(0, 256, (1024, 1024, 3)).astype(np.uint8)
    units = list(T.pyramid_units(im))
    fd = T.FusedDetector(net, n_lanes=10, mode="group")
    ref_dets = fd.detect(units, thresh=0.05)[0]

    def image_metrics():
        got = fd.detect(units, thresh=0.05)[0]
        pairs, miss, extra = match_detections(got, ref_dets, score_tol=0.05)
        gi = np.array([p[0] for p in pairs], int)
        wi = np.array([p[1] for p in pairs], int)
        ds = float(np.abs(got[gi, 4] - ref_dets[wi, 4]).max()) if len(pairs) else None
        px = int(np.any(written(got[gi]) != written(ref_dets[wi]), axis=1).sum()) if len(pairs) else 0
        return {"detections": int(len(got)), "detections_fp32": int(len(ref_dets)), "matched": len(pairs),
                "unmatched": len(miss) + len(extra), "max_abs_dscore_vs_fp32": ds, "rows_with_written_pixel_diff": px}

    def throughput(n=12):
        import torch
        dev = [(torch.from_numpy(u[0]).cuda(),) + tuple(u[1:]) for u in units]
        ul = [(d[0].data_ptr(),) + tuple(d[1:]) for d in dev]
        for _ in range(3):
            fd.submit(ul, 0.05, on_device=True)
            if fd.pending() > 1:
                fd.collect()
        while fd.pending():
            fd.collect()
        t0 = time.perf_counter()
        for _ in range(n):
            fd.submit(ul, 0.05, on_device=True)
            if fd.pending() > 1:
                fd.collect()
        while fd.pending():
            fd.collect()
        return n / (time.perf_counter() - t0)

    def measure(tag):
        sc, dl = level_scores(d512, i512)
        s1008, _ = level_scores(d1008, i1008)
        r = {"c1_max_abs_dscore_vs_oracle": float(np.abs(sc - o_sc).max()),
             "c1_max_abs_ddelta_vs_oracle": float(np.abs(dl - o_dl).max()),
             "l1008_max_abs_dscore_vs_fp32": float(np.abs(s1008 - ref1008).max())}
        print(tag, r, flush=True)
        return r

    for mode in ("f16x3", "f16x2", "f16", "bf16"):
        net.set_conv_mode(mode)
        r = measure(mode)
        r.update(image_metrics())
        r["images_per_s"] = throughput()
        print(mode, r, flush=True)
        out["modes"][mode] = r

    # one layer at a time inside the parity mode: which layers tolerate cheaper arithmetic?
    net.set_conv_mode("f16x3")
    layers = ["conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3",
              "conv4_fuse_final", "conv4_fuse_final_dim_red", "head_1"]
    for L in layers:
        for n in ((2,) if quick else (2, 1)):
            net.set_layer_products({L: n})
            out["single_layer"]["%s:%d" % (L, n)] = measure("%s:%d" % (L, n))
        net.set_layer_products({L: 0})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "precision_ladder.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
