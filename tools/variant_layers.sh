#!/bin/bash
# Per-layer dispatch times of one bench image under different tile-shape knobs of the dual-tile family (A/B on ONE box):
#   tools/variant_layers.sh   -> gpurun_out/variants/<name>.txt
set -u
root=$(pwd)
out=$root/gpurun_out/variants
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
run() {
  name=$1; shift
  rm -rf $out/tr_$name
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$name -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --sustain-seconds 0 > $out/$name.json 2> $out/$name.err
  python3 tools/trace_layers.py $out/tr_$name > $out/$name.txt 2>&1
  find $out/tr_$name -name "*.db" -delete
}
# (rocprofv3 must start python3 itself: knobs are exported, not passed through `env`)
( run default )
( export SHF_F16X3_W4_MT=2 SHF_F16X3_W4D_NTILE=1; run mt2_n1 )
( export SHF_F16X3_W4_MT=2 SHF_F16X3_W4D_NTILE=2; run mt2_n2 )
( export SHF_F16X3_W4_MT=4 SHF_F16X3_W4D_NTILE=1; run mt4_n1 )
( export SHF_F16X3_W4_MT=4 SHF_F16X3_W4D_NTILE=2; run mt4_n2 )
( export SHF_F16X3_W4=1; run w4_all )
tail -n 40 $out/default.txt
