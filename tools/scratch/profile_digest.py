"""Print every figure DESIGN.md / BASELINE.md quote from a profiles_<tag> directory (default gpurun_out/profiles_r06)."""
import csv, json, sys
P = (sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_r06").rstrip("/") + "/"
tag = "r06"
d = json.load(open(P + tag + "_bench.json"))
r, t = d["roofline"], d["sustained"]["telemetry"]
print("value %.1f  ms %.2f  frac %.3f issued %.3f  dom ms %.3f  sustained %.1f  latency %.1f" % (d["value"], d["ms_per_step"], r["frac"], r["frac_issued"], r["avg_launch_ms"], d["sustained"]["value"], d["latency_ms"]))
print("telemetry sclk %d MHz power %d W cap %d" % (t["sclk_mhz_mean"], t["power_w_mean"], t["power_cap_w"]))
m = r["matrix_pipe_sustained"]
print("pipe", m["frac_of_peak"], "issued_of_sustained", m["frac_issued_of_sustained"], {k: (round(v["sclk_mhz_mean"]), round(v["power_w_mean"])) for k, v in m["telemetry"].items()}, {k: round(v) for k, v in m.items() if k.startswith("operands")})
print("mixed %.1f (%.3f)  files %.1f (%.3f of %.1f) prefetch %d decode/step %.2f decode %.1f ms" % (d["mixed_shapes"]["value"], d["mixed_shapes"]["flop_normalised_vs_resident"], d["from_files"]["value"], d["from_files"]["vs_same_images_from_memory"], d["from_files"]["same_images_from_memory"], d["from_files"]["decode_prefetch"], d["from_files"]["decode_over_step"], d["from_files"]["decode_ms"]))
n = d["net_forward_path"]
print("net_forward_path", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in n.items() if k not in ("path", "kernel_ms_per_image")})
print("reduced bf16 %.1f f16 %.1f  cpu %.4f (%.1f s)  busy %.3f traffic %.3e" % (d["reduced_precision"]["value"], d["reduced_precision"]["also"]["value"], d["cpu_baseline"]["value"], 1 / d["cpu_baseline"]["value"], r["mfma_busy"] or 0, r["traffic"] or 0))
rows = list(csv.DictReader(open(P + tag + "_layers.csv")))
us = sum(float(x["us"]) for x in rows); gf = sum(float(x["algorithmic_gflop"]) for x in rows)
print("stack %.1f us  alg %.3f issued %.3f" % (us, gf / us / 2500 * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e3, 3 * gf / (us * 1e-6) / 1e12 / 2500))
w4d = sum(float(x["us"]) for x in rows if "w4d" in x["kernel"]); print("w4d family %.1f us" % w4d)
for x in rows:
    print("%-26s %7.1f us  alg %.3f iss %.3f busy %.3f hbm %s" % (x["layer"], float(x["us"]), float(x["frac_of_fp16_peak"]), float(x["frac_issued"]), float(x["mfma_busy"]), x["hbm_over_algorithmic"]))
p = json.load(open(P + tag + "_pmc.json"))
for k in ("conv_mfma_f16x3_w4d_kernel<true, 4, 2, 3, false, 1>", "conv_mfma_f16x3_pc_kernel<3, false, true>", "conv_mfma_f16x3_heads3_kernel<true, 3>"):
    v = p["kernels"][k]; print(k, "%.1f us clk %.3f busy %.3f ldsconf %.3f" % (v["avg_us"], v["effective_clock_ghz"], v["mfma_busy"], v["lds_bank_conflict_frac"]))
u = json.load(open(P + tag + "_bench_under_rocprof.json")); print("under rocprof %.1f %.4f" % (u["value"], u["roofline"]["avg_launch_ms"]))
f = json.load(open(P + tag + "_bench_fp32_mode.json")); print("fp32 %.1f %.3f" % (f["value"], f["roofline"]["frac"]))
g = json.load(open(P + tag + "_bench_from_uint8_image.json")); print("uint8 %.1f" % g["value"])
hb = sum(float(x["hbm_bytes"]) for x in rows); al = sum(float(x["algorithmic_bytes"]) for x in rows); print("hbm %.2f GB alg %.2f GB" % (hb / 1e9, al / 1e9))
