set -u
cd $GRAFT_REPO_ROOT
[ "${2:-}" = "test" ] && python -m pytest tests/test_gpu_parity.py tests/test_gpu_magnitudes.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-reduced --no-latency --no-calib --no-mixed --no-forward-path --overlap-seconds 0 --sustain-seconds 0 2>/dev/null | tail -1 > gpurun_out/ab_new$i.json
SHF_LIB=$GRAFT_REPO_ROOT/variants/${1:-pre_dbuf}.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-reduced --no-latency --no-calib --no-mixed --no-forward-path --overlap-seconds 0 --sustain-seconds 0 2>/dev/null | tail -1 > gpurun_out/ab_old$i.json
done
python - <<'PY'
import json
for n in ("new1","old1","new2","old2"):
    d=json.load(open("gpurun_out/ab_%s.json"%n)); r=d["roofline"]
    print(n, "%.2f img/s"%d["value"], {k.replace("conv_mfma_f16x3_",""):v for k,v in r["kernel_ms_per_image"].items() if "w4d" in k or "pc_kernel" in k or "heads3" in k})
PY
