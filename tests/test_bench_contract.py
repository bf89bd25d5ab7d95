"""The committed bench line (profiles/r02_bench.json, written by bench.py on an MI355X) carries every field
the bench contract names, and the committed rocprofv3 summary names the same dominant kernel."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
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
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_rocprof_summary_names_the_dominant_kernel():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_under_rocprof.json")))
    name = d["roofline"]["kernel"]
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_bench_kernel_stats.csv"))))
    hit = [r for r in rows if name in r["Name"]]
    assert hit, name
    avg_ms = float(hit[0]["AverageNs"]) / 1e6
    # HIP-event average of the timed steps vs rocprofv3 average over all launches of the run: within 5 %
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05


def test_committed_pmc_summary_and_layer_table():
    """profiles/r02_pmc.json (counter passes) and r02_layers.csv (one row per conv launch of an image) are what the
    roofline numbers can be recomputed from; the layer table covers the whole image's algorithmic work."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc.json")))
    dom = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))["roofline"]["kernel"]
    assert dom.startswith("conv_mfma_f16x3_w4")     # a 4-wave split-fp16 kernel (the dual-tile family since round 2)
    k = d["kernels"][dom]
    for key in ("hbm_bytes_per_launch", "mfma_busy", "effective_clock_ghz", "lds_bank_conflict_frac", "avg_us"):
        assert key in k, key
    assert 0.0 < k["mfma_busy"] <= 1.0
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_layers.csv"))))
    assert len(rows) == 19 and rows[0]["layer"] == "conv1_1+conv1_2"
    gf = sum(float(r["algorithmic_gflop"]) for r in rows)
    assert abs(gf - 5021.6) < 2.0          # 5021.62 GFLOP per image (the deconv's 0.1 GFLOP aside)
