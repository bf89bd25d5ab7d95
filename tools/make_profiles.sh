#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root, e.g. via gpurun):
#   tools/make_profiles.sh r06
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> <tag>_bench_kernel_stats.csv
#    + the JSON line bench.py printed in that run                    -> <tag>_bench_under_rocprof.json
# 2. counter passes, each its own run with --kernel-trace only (never with --stats / --sys-trace):
#      FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE |
#      SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE | SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
#    -> <tag>_pmc.json (per kernel, per launch; carries the kernel-source hash) and <tag>_layers.csv (one row per
#    conv layer of one image: µs, algorithmic TFLOP/s, issued-MFMA fraction, MFMA-busy, HBM bytes vs algorithmic)
# 3. un-profiled bench lines: default, --conv-mode fp32, --host-input image
# Everything is written under gpurun_out/prof_<tag>/ and the summaries copied to gpurun_out/profiles_<tag>/
# (gpurun merges gpurun_out/ back; copy from there into profiles/ and commit).
set -u
tag=${1:-r06}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles_$tag
mkdir -p $out $dst
cd /tmp && export TMPDIR=/tmp && cd $root
BENCH_PMC="bench.py --steps 2 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --no-forward-path --overlap-seconds 0 --sustain-seconds 0"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $BENCH_PMC > $out/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$i -- python3 $BENCH_PMC > $out/pmc_$i.log 2>&1
done
python3 tools/layer_table.py $out $dst $tag > $out/layer_table.log 2>&1
tail -40 $out/layer_table.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 2 --no-reduced --no-calib --no-mixed --no-forward-path --overlap-seconds 0 --sustain-seconds 0 --no-cpu-baseline > $out/bench_under_rocprof.out 2> $out/bench_under_rocprof.log
grep '^{"metric"' $out/bench_under_rocprof.out | tail -1 > $dst/${tag}_bench_under_rocprof.json
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $dst/${tag}_bench_kernel_stats.csv
# the bench line quotes traffic / MFMA-busy from the committed counter file: make this run's the committed one
cp $dst/${tag}_pmc.json $dst/${tag}_layers.csv $root/profiles/ 2>/dev/null
python3 bench.py 2>/dev/null | tail -1 > $dst/${tag}_bench.json
python3 bench.py --steps 5 --warmup 2 --conv-mode fp32 --no-cpu-baseline --no-reduced --no-mixed --overlap-seconds 0 --sustain-seconds 0 2>/dev/null | tail -1 > $dst/${tag}_bench_fp32_mode.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reduced --no-mixed --overlap-seconds 0 --sustain-seconds 0 --host-input image 2>/dev/null | tail -1 > $dst/${tag}_bench_from_uint8_image.json
# keep the merge-back small: the raw traces stay on the box except the csv files the table was made from
find $out -name "*.db" -delete
ls -la $dst
