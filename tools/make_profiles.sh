#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root, e.g. via gpurun):
#   tools/make_profiles.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> <tag>_bench_kernel_stats.csv
#    + the JSON line bench.py printed in that run                    -> <tag>_bench_under_rocprof.json
# 2. un-profiled bench (default and --conv-mode fp32)                -> <tag>_bench.json, <tag>_bench_fp32_mode.json
# 3. (run FIRST, so that the bench lines carry the traffic) two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE)
#    with --kernel-trace only -> <tag>_pmc_hbm_bytes.json
# Everything is first written under gpurun_out/prof_<tag>/ and the summaries copied to gpurun_out/profiles_<tag>/
# (gpurun merges gpurun_out/ back; copy from there into profiles/ and commit).
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles_$tag
mkdir -p $out $dst
cd /tmp && export TMPDIR=/tmp && cd $root
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-events --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
python3 - "$out" "$dst/${tag}_pmc_hbm_bytes.json" <<'PY'
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
tab = {}
for c, key in (("FETCH_SIZE", "fetch_size_kb_per_launch"), ("WRITE_SIZE", "write_size_kb_per_launch")):
    files = glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            name = r["Kernel_Name"].split("(")[0]
            if not ("conv" in name or "deconv" in name or "pyramid" in name):
                continue
            a = acc.setdefault(name, {})
            did = r["Dispatch_Id"]
            a[did] = a.get(did, 0.0) + float(r["Counter_Value"])   # summed over XCDs / dimensions
    for name, d in acc.items():
        t = tab.setdefault(name, {})
        t["launches"] = len(d)
        t[key] = sum(d.values()) / max(len(d), 1)
json.dump(tab, open(dst, "w"), indent=1)
print(json.dumps(tab, indent=1)[:1500])
PY
cp $dst/${tag}_pmc_hbm_bytes.json profiles/${tag}_pmc_hbm_bytes.json   # bench.py reads the per-launch traffic from here
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 > $out/bench_under_rocprof.out 2> $out/bench_under_rocprof.log
grep '^{"metric"' $out/bench_under_rocprof.out | tail -1 > $dst/${tag}_bench_under_rocprof.json
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $dst/${tag}_bench_kernel_stats.csv
python3 bench.py 2>/dev/null | tail -1 > $dst/${tag}_bench.json
python3 bench.py --steps 5 --warmup 2 --conv-mode fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $dst/${tag}_bench_fp32_mode.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-input image 2>/dev/null | tail -1 > $dst/${tag}_bench_from_uint8_image.json
ls -la $dst
