"""Per-kernel and per-layer roofline evidence from the rocprofv3 passes of tools/make_profiles.sh.

    python tools/layer_table.py <prof_dir> <dst_dir> <tag>

Reads <prof_dir>/trace (kernel trace) and <prof_dir>/pmc_N (one counter set each), all runs of the same command
(bench.py --steps 2 --warmup 1 ...): the dispatch sequence is deterministic, so dispatch k of one run is dispatch k
of another.  Writes
  <tag>_pmc.json    per kernel name: launches, avg µs, counters per launch, derived HBM bytes / MFMA-busy
  <tag>_layers.csv  one row per convolution launch of ONE image (the last complete one), in graph order
Counter arithmetic (MI355X_MICROARCH.md): HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE counts
128-B requests as 64 B; both are reported in KiB); SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD-issued MFMA
(32 per v_mfma_f32_32x32x16_f16) summed over all SIMDs (checked: the value equals 32 x the MFMA count of the launch),
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, so the dispatch lasted GRBM_GUI_ACTIVE / 8 cycles and
mfma_busy = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 4 SIMDs x 256 CUs); effective clock = GRBM_GUI_ACTIVE / 8 / duration.
mfma_busy is utilisation AT THE SUSTAINED CLOCK; x effective_clock / 2.4 GHz gives the fraction of the nominal peak.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_CU, N_SIMD, N_XCD = 256, 4, 8
PEAK_F16 = 2500.0


def load_trace(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def load_counters(d):
    """{counter: [value per dispatch in start order]} (+ 'names'), summed / maxed over the XCD dimension rows."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    order = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            did = int(r["Dispatch_Id"])
            order[did] = (int(r.get("Start_Timestamp", 0) or 0), r["Kernel_Name"])
            c = r["Counter_Name"]
            v = float(r["Counter_Value"])
            a = per.setdefault(c, {}).setdefault(did, [0.0, 0.0])
            a[0] += v
            a[1] = max(a[1], v)
    dids = sorted(order, key=lambda k: (order[k][0], k))
    out = {"names": [order[k][1] for k in dids]}
    for c, dd in per.items():
        out[c] = [dd.get(k, [0.0, 0.0])[0] for k in dids]
        out[c + "#max"] = [dd.get(k, [0.0, 0.0])[1] for k in dids]
    return out


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "shf::"):
        n = n.replace(p, "")
    return n.strip()


def conv_layers_of_the_bench_image():
    """(name, cin, cout, k, pixels summed over the 10 units) for every MFMA conv launch of an image, in launch order."""
    from smallhardface_amd import prototxt as P
    from smallhardface_amd.config import cfg_from_file
    cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))   # different_dilation + dim_red, like bench.py
    msg = P._add_dimension_reduction(P.build_test_template(True))
    sides = [112, 304, 608, 1008, 1408]
    layers, down, cin_of = [], {"data": 1}, {"data": 3}
    for L in msg.getall("layer"):
        t, name = L.get("type"), L.get("name")
        bot = L.getall("bottom")
        top = L.getall("top")
        if t == "Convolution":
            cp = L.get("convolution_param")
            cout, k = int(cp.get("num_output")), int(cp.get("kernel_size"))
            d = down[bot[0]]
            px = 2 * sum((s // d) ** 2 for s in sides)
            layers.append((name, cin_of[bot[0]], cout, k, px))
            down[top[0]], cin_of[top[0]] = d, cout
        elif t == "Pooling":
            down[top[0]], cin_of[top[0]] = down[bot[0]] * 2, cin_of[bot[0]]
        elif t == "Deconvolution":
            down[top[0]], cin_of[top[0]] = down[bot[0]] // 2, cin_of[bot[0]]
        elif t == "Concat":
            down[top[0]], cin_of[top[0]] = down[bot[0]], sum(cin_of[b] for b in bot)
        else:
            for tp in top:
                if bot:
                    down[tp], cin_of[tp] = down.get(bot[0], 1), cin_of.get(bot[0], 0)
    # launches: conv1_1 rides inside conv1_2's launch; cls/bbox 1x1s are part of the tail (not MFMA launches)
    out = []
    for name, cin, cout, k, px in layers:
        if name.startswith("cls_score") or name.startswith("bbox_pred"):
            continue
        out.append([name, cin, cout, k, px, 2.0 * px * cin * cout * k * k])
    assert out[0][0] == "conv1_1" and out[1][0] == "conv1_2"
    out[1][0] = "conv1_1+conv1_2"
    out[1][5] += out[0][5]
    return out[1:]


def main():
    prof, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    from tools.kernel_hash import kernel_source_hash
    trace = load_trace(os.path.join(prof, "trace"))
    passes = [load_counters(p) for p in sorted(glob.glob(os.path.join(prof, "pmc_*"))) if os.path.isdir(p)]
    counters = {}
    for p in passes:
        for k, v in p.items():
            if k != "names":
                counters[k] = (p["names"], v)
    names = [r["Kernel_Name"] for r in trace]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace]
    # Dispatch k of kernel NAME in one run is dispatch k of that name in another (launches of one kernel are in
    # stream order; kernels of different streams may interleave differently from run to run).
    def occurrence_index(nm):
        seen, out = {}, []
        for n in nm:
            k = short(n)
            out.append((k, seen.get(k, 0)))
            seen[k] = seen.get(k, 0) + 1
        return out
    trace_occ = occurrence_index(names)
    cmap = {}
    for c, (nm, v) in counters.items():
        occ = occurrence_index(nm)
        cmap[c] = {o: v[i] for i, o in enumerate(occ)}
        if sorted(occ) != sorted(trace_occ):
            print("WARNING: counter pass %s saw a different set of dispatches than the trace (%d vs %d)" % (c, len(occ), len(names)))

    def cval(c, i):
        return cmap.get(c, {}).get(trace_occ[i])

    # ---- per kernel name
    per = {}
    for i, n in enumerate(names):
        k = short(n)
        if not ("conv" in k or "tail" in k or "bitonic" in k or "append" in k or "scan" in k or "vote" in k or "iou" in k):
            continue
        a = per.setdefault(k, {"launches": 0, "us": 0.0})
        a["launches"] += 1
        a["us"] += dur[i]
        for c in counters:
            if cval(c, i) is not None:
                a[c] = a.get(c, 0.0) + cval(c, i)
    kernels = {}
    for k, a in per.items():
        n = a["launches"]
        e = {"launches": n, "avg_us": a["us"] / n}
        for c in counters:
            if c in a:
                e[c + "_per_launch"] = a[c] / n
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            e["hbm_bytes_per_launch"] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0 / n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("GRBM_GUI_ACTIVE#max"):
            cyc = a["GRBM_GUI_ACTIVE"] / N_XCD
            e["mfma_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD * N_CU)
            clk = cyc / (a["us"] * 1e3)
            # GRBM_GUI_ACTIVE counts the chip's busy cycles while the dispatch is in flight: a small kernel that shares the
            # chip with another stream's grid (tails / merge beside the next image's convolutions) reads as tens of GHz
            if clk <= 2.6:
                e["effective_clock_ghz"] = clk
                e["mfma_busy_x_clock_over_2p4"] = e["mfma_busy"] * clk / 2.4
            else:
                e["effective_clock_ghz"] = None
                e["overlapped_with_another_stream"] = True
        if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]
        kernels[k] = e
    json.dump({"kernel_source_hash": kernel_source_hash(), "command": "bench.py --steps 2 --warmup 1 --no-events",
               "note": "profiled passes clock lower than un-profiled runs (DVFS): compare ratios, not absolute µs",
               "kernels": kernels}, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)

    # ---- per layer of the last complete image
    pcs = [i for i, n in enumerate(names) if "f16x3_pc_kernel" in n]
    a = pcs[-1]
    conv_idx = [i for i in range(a, len(names)) if "conv_mfma" in names[i]]
    layers = conv_layers_of_the_bench_image()
    if any("heads3_kernel" in names[i] for i in conv_idx):
        # the three shared-weight dilated heads are ONE launch (conv_f16x3_h3.h): one row, the input counted once
        hs = [q for q, l in enumerate(layers) if l[0] in ("head_1", "head_2", "head_4")]
        assert len(hs) == 3 and hs[2] == hs[0] + 2
        h0 = layers[hs[0]]
        merged = ["head_1+head_2+head_4", h0[1], h0[2], h0[3], h0[4], sum(layers[q][5] for q in hs), 4.0 * h0[4] * (h0[1] + 3 * h0[2])]
        layers = layers[:hs[0]] + [merged] + layers[hs[2] + 1:]
    # A layer of the dual-tile family may be TWO consecutive launches: two tiles per block for the whole rounds, then
    # single tiles for the rest (same <IN_SPLIT, rows>).  Merge such pairs while there are more dispatches than layers.
    import re
    def w4d(n):
        m = re.search(r"w4d_kernel<(\w+), (\d), (\d), (\d)(?:, \w+)?(?:, (\d))?>", n)
        # (the dilated heads -- last template argument 2 / 4 -- are layers of their own, never the second half of a pair)
        return None if m is None or (m.group(5) or "1") != "1" else (m.group(1), m.group(2), int(m.group(3)))
    groups, j = [], 0
    while j < len(conv_idx) and len(groups) < len(layers):
        i = conv_idx[j]
        left_d, left_l = len(conv_idx) - j, len(layers) - len(groups)
        a_, b_ = w4d(names[i]), w4d(names[conv_idx[j + 1]]) if j + 1 < len(conv_idx) else None
        if a_ and b_ and a_[2] == 2 and b_[2] == 1 and a_[:2] == b_[:2] and left_d > left_l:
            groups.append([i, conv_idx[j + 1]])
            j += 2
        else:
            groups.append([i])
            j += 1
    def gsum(c, g):
        v = [cval(c, i) for i in g]
        return None if any(x is None for x in v) else sum(v)
    rows = []
    for lrec, g in zip(layers, groups):
        lname, cin, cout, k, px, fl = lrec[:6]
        i = g[0]
        us = sum(dur[x] for x in g)
        tf = fl / (us * 1e-6) / 1e12
        alg_bytes = lrec[6] if len(lrec) > 6 else 4.0 * px * (cin + cout)   # input + output once, 4 B per element (weights: < 10 MB, L2-resident)
        hbm = None
        if gsum("FETCH_SIZE", g) is not None and gsum("WRITE_SIZE", g) is not None:
            hbm = (2.0 * gsum("FETCH_SIZE", g) + gsum("WRITE_SIZE", g)) * 1024.0
        busy = None
        if gsum("SQ_VALU_MFMA_BUSY_CYCLES", g) is not None and gsum("GRBM_GUI_ACTIVE", g):
            busy = gsum("SQ_VALU_MFMA_BUSY_CYCLES", g) / (gsum("GRBM_GUI_ACTIVE", g) / N_XCD * N_SIMD * N_CU)
        rows.append({"layer": lname, "kernel": " + ".join(short(names[x]) for x in g), "cin": cin, "cout": cout, "k": k, "pixels": px,
                     "us": round(us, 1), "algorithmic_gflop": round(fl / 1e9, 2), "algorithmic_tflops": round(tf, 1),
                     "frac_of_fp16_peak": round(tf / PEAK_F16, 4), "frac_issued": round(3.0 * tf / PEAK_F16, 4),
                     "mfma_busy": None if busy is None else round(busy, 4),
                     "hbm_bytes": None if hbm is None else int(hbm), "algorithmic_bytes": int(alg_bytes),
                     "hbm_over_algorithmic": None if hbm is None else round(hbm / alg_bytes, 2)})
    with open(os.path.join(dst, tag + "_layers.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        print(r)
    print("sum of conv µs of one image: %.1f; algorithmic GFLOP %.1f" % (sum(r["us"] for r in rows), sum(r["algorithmic_gflop"] for r in rows)))


if __name__ == "__main__":
    main()
