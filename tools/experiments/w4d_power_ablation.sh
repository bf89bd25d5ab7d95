#!/bin/bash
# Timing-only ablation of the dual-tile conv kernel's data movement (WRONG results, same operand statistics): what do
# the in-loop weight fetch and the halo hand-over cost a power-limited kernel?  Builds patched copies of
# conv_f16x3.hip into variants/*.so (tools/build_variant.sh), nothing in the shipped source changes.
#   no_wdma : stages >= 2 fetch no weights (both LDS buffers keep the real weights of stages 0 / 1)
#   no_halo : no hand-over (every chunk multiplies chunk 0's halo tiles)
#   no_both : both
# On the GPU box:  tools/experiments/w4d_power_ablation.sh run   -> gpurun_out/ablation/*.json + clock tables
set -e
cd "$(dirname "$0")/../.."
src=smallhardface_amd/csrc/conv_f16x3.hip
if [ "${1:-build}" = "build" ]; then
  mkdir -p variants/_src
  python3 - <<'PY'
import re
s = open("smallhardface_amd/csrc/conv_f16x3.hip").read()
a = "    const unsigned char* ws_ = (const unsigned char*)(wbase + (size_t)stage * 3 * slab);\n    unsigned char* bd_ = Bs + buf * (3 * SLAB_B);"
assert s.count(a) == 2          # the dual-tile kernel's dma_w comes first, the fused pair's second
no_wdma = s.replace(a, "    if (stage >= 2) return;\n" + a, 1)
b = """    stage(c, integral_constant<int, 2>{}, integral_constant<int, 1>{});"""
assert s.count(b) == 1
no_halo = s.replace(b, "    stage(c, integral_constant<int, 2>{}, integral_constant<int, 0>{});")
no_both = no_wdma.replace(b, "    stage(c, integral_constant<int, 2>{}, integral_constant<int, 0>{});")
for n, t in (("no_wdma", no_wdma), ("no_halo", no_halo), ("no_both", no_both)):
    open("variants/_src/%s.hip" % n, "w").write(t)
PY
  for v in no_wdma no_halo no_both; do
    SRC_OVERRIDE=variants/_src/$v.hip tools/build_variant.sh abl_$v conv_f16x3.hip
  done
  exit 0
fi
root=$(pwd)
out=$root/gpurun_out/ablation
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
one() {
  name=$1
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reduced --no-latency 2>$out/$name.err | tail -1 > $out/$name.json
  rm -rf $out/pmc_$name
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib > $out/pmc_$name.log 2>&1
  ( cd tools && python3 pmc_clock_table.py $out/pmc_$name > $out/clock_$name.txt 2>&1 )
  find $out/pmc_$name -name "*.db" -delete
  find $out/pmc_$name -name "*.csv" -delete
}
( one default )
for v in no_wdma no_halo no_both; do
  ( export SHF_LIB=$root/variants/abl_$v.so; one $v )
done
( one default2 )
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.getcwd(), "gpurun_out", "ablation")
for n in ("default", "no_wdma", "no_halo", "no_both", "default2"):
    try:
        d = json.load(open(os.path.join(out, n + ".json")))
    except Exception as e:
        print(n, "failed", e)
        continue
    r = d["roofline"]
    k = "conv_mfma_f16x3_w4d_kernel<true, 4, 2, 3, false>"
    print("%-9s %6.2f images/s  dominant %.3f ms/launch  per-image ms: %s" % (
        n, d["value"], r["avg_launch_ms"], {kk.replace("conv_mfma_f16x3_", ""): v for kk, v in r["kernel_ms_per_image"].items() if "w4d" in kk}))
    for line in open(os.path.join(out, "clock_%s.txt" % n)):
        if "w4d_kernel<true, 4, 2" in line or "kernel " in line:
            print("          ", line.rstrip())
PY
