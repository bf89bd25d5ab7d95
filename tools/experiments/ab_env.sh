#!/bin/bash
# A/B of an environment knob on ONE box: tools/experiments/ab_env.sh KNOB=VALUE [test]
#   alternates `python bench.py` without and with the knob (2 x 2 runs of 30 steps); `test` runs the conv parity files
#   with the knob set first.
set -u
cd ${GRAFT_REPO_ROOT:-.}
kv=$1
if [ "${2:-}" = "test" ]; then ( export $kv; python -m pytest tests/test_gpu_parity.py tests/test_gpu_magnitudes.py tests/test_gpu_reduced_modes.py -x -q -m gpu 2>&1 | tail -3 ); fi
for i in 1 2; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-reduced --no-latency --no-calib 2>/dev/null | tail -1 > gpurun_out/ab_off$i.json
  ( export $kv; python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-reduced --no-latency --no-calib 2>/dev/null | tail -1 > gpurun_out/ab_on$i.json )
done
python - <<'PY'
import json
for n in ("off1", "on1", "off2", "on2"):
    d = json.load(open("gpurun_out/ab_%s.json" % n)); r = d["roofline"]
    print(n, "%.2f img/s" % d["value"], {k.replace("conv_mfma_f16x3_", ""): v for k, v in r["kernel_ms_per_image"].items() if "pc_" in k or "w4d_kernel<true, 4, 2" in k or ", 1, 1, 3, false>" in k})
PY
