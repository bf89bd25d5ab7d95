#!/bin/bash
# tools/scratch/ab_lds.sh NAME: LDS bank-conflict fraction + MFMA busy of the conv kernels, shipped library vs variants/NAME.so
set -u
root=$(pwd); out=$root/gpurun_out/ablds; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
one() {
  rm -rf $out/pmc_$1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_$1 -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --sustain-seconds 0 > $out/pmc_$1.log 2>&1
  python3 - $out/pmc_$1 <<'PY'
import sys
sys.path.insert(0, "tools")
from layer_table import load_counters, short
c = load_counters(sys.argv[1])
acc = {}
for k, n in enumerate(c["names"]):
    a = acc.setdefault(short(n), [0.0, 0.0])
    a[0] += c["SQ_LDS_BANK_CONFLICT"][k]; a[1] += c["SQ_LDS_IDX_ACTIVE"][k]
for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-60s conflict / active = %.3f" % (n[:60], a[0] / max(1.0, a[1])))
PY
  find $out/pmc_$1 -name "*.db" -delete; find $out/pmc_$1 -name "*.csv" -delete
}
echo "== shipped"; ( one new )
echo "== $1"; ( export SHF_LIB=$root/variants/$1.so; one $1 )
