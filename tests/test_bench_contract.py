"""The committed bench line (profiles/r06_bench.json, written by bench.py on an MI355X) carries every field
the bench contract names, and the committed rocprofv3 summary names the same dominant kernel."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    if isinstance(base.get("metric"), str):
        assert d["unit"] in ("images/s",) and d["value"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["traffic_measured_in_this_run"] is False            # PMC passes are separate runs (tools/make_profiles.sh)
    assert abs(r["frac_issued"] - 3.0 * r["frac"]) < 1e-9 and "mfma_busy" in r
    if "matrix_pipe_sustained" in r:     # (lines written before the calibration leg existed do not carry it)
        m = r["matrix_pipe_sustained"]
        assert m["operands_random"] < m["operands_random_half_of_activations_zero"] < m["operands_constant"] <= 1.02 * r["peak"]
        assert abs(m["frac_issued_of_sustained"] - r["issued_mfma_achieved"] / m["operands_random_half_of_activations_zero"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert "all five pyramid levels" in c["sample"]               # the whole image's work, not a FLOP-scaled slice
    # the reduced-precision leg (BASELINE configs C3 / C5 name bf16): outside the timed region, drift-labelled
    rp = d["reduced_precision"]
    for k in ("mode", "value", "max_abs_dscore_vs_fp32", "boxes_matched"):
        assert k in rp and k in rp["also"], k
    assert rp["mode"] == "bf16" and rp["also"]["mode"] == "f16" and rp["value"] > d["value"]
    assert 1e-4 < rp["max_abs_dscore_vs_fp32"] < 0.1 and 1e-5 < rp["also"]["max_abs_dscore_vs_fp32"] < 0.02


def test_bench_json_round6_legs():
    """Round 6: sampled clock / socket power of the sustained leg and of the matrix-pipe calibration rows (host-side hwmon
    sampler), and the literal drop-in path -- lib/test.py's detect() with ten Net.forward() calls and host
    blobs -- as a rate of its own with its h2d / forward / d2h split, detections identical to the fused path (#3)."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    t = d["sustained"]["telemetry"]
    for k in ("sclk_mhz_mean", "sclk_mhz_min", "power_w_mean", "power_cap_w", "samples"):
        assert k in t, k
    assert t["samples"] > 50 and 500 < t["sclk_mhz_mean"] <= 2500 and 100 < t["power_w_mean"] <= t["power_cap_w"] * 1.05
    ct = d["roofline"]["matrix_pipe_sustained"]["telemetry"]
    assert set(ct) == {"operands_constant", "operands_random", "operands_random_half_of_activations_zero"}
    # toggling operands cost clock: the constant-operand stream runs faster than the random one, and draws less
    assert ct["operands_constant"]["sclk_mhz_mean"] > ct["operands_random"]["sclk_mhz_mean"]
    assert ct["operands_constant"]["power_w_mean"] < ct["operands_random"]["power_w_mean"]
    n = d["net_forward_path"]
    for k in ("value", "ms_per_image", "h2d_ms", "forward_ms", "d2h_ms", "preprocess_ms", "forward_calls_per_image",
              "identical_to_fused_path", "vs_fused_rate"):
        assert k in n, k
    assert n["forward_calls_per_image"] == 10 and n["identical_to_fused_path"] is True
    assert 0 < n["value"] < d["value"] and abs(n["vs_fused_rate"] - n["value"] / d["value"]) < 1e-6
    assert n["h2d_ms"] > 0 and n["forward_ms"] > n["h2d_ms"]
    f = d["from_files"]
    assert "decode_prefetch" in f and f["decode_prefetch"] >= 2
    o = d["overlapped_pipeline"]       # a measurement leg: what overlapping consecutive images' convolutions would add (nothing)
    assert o["identical_to_headline"] is True and 0.9 < o["vs_value"] < 1.1 and "never `value`" in o["note"]


def test_rocprof_summary_names_the_dominant_kernel():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_under_rocprof.json")))
    name = d["roofline"]["kernel"]
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_bench_kernel_stats.csv"))))
    hit = [r for r in rows if name in r["Name"]]
    assert hit, name
    avg_ms = float(hit[0]["AverageNs"]) / 1e6
    # HIP-event average of the timed steps vs rocprofv3 average over all launches of the run: within 5 %
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05


def test_committed_pmc_summary_and_layer_table():
    """profiles/r06_pmc.json (counter passes) and r06_layers.csv (one row per conv launch of an image) are what the
    roofline numbers can be recomputed from; the layer table covers the whole image's algorithmic work."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc.json")))
    dom = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))["roofline"]["kernel"]
    assert dom.startswith("conv_mfma_f16x3_w4d")    # the dual-tile 4-wave family
    k = d["kernels"][dom]
    for key in ("hbm_bytes_per_launch", "mfma_busy", "effective_clock_ghz", "lds_bank_conflict_frac", "avg_us"):
        assert key in k, key
    assert 0.0 < k["mfma_busy"] <= 1.0
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_layers.csv"))))
    assert len(rows) == 17 and rows[0]["layer"] == "conv1_1+conv1_2"     # (the three dilated heads are one launch: one row)
    gf = sum(float(r["algorithmic_gflop"]) for r in rows)
    assert abs(gf - 5021.6) < 2.0          # 5021.62 GFLOP per image (the deconv's 0.1 GFLOP aside)


def test_committed_counters_belong_to_the_committed_kernels():
    """profiles/r06_pmc.json carries the hash of the kernel sources it was measured on (tools/kernel_hash.py: comments and
    white space do not count); bench.py only quotes traffic / MFMA-busy from it while that matches -- so must the tree."""
    import sys
    sys.path.insert(0, ROOT)
    from tools.kernel_hash import kernel_source_hash
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc.json")))
    assert d["kernel_source_hash"] == kernel_source_hash()
    b = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    assert b["roofline"]["traffic"] is not None and b["roofline"]["mfma_busy"] is not None
