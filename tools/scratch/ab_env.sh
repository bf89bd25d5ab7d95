#!/bin/bash
# tools/scratch/ab_env.sh VAR=VAL ...: per-dispatch conv times of one bench image, default vs with the variables set, twice each
set -u
root=$(pwd); out=$root/gpurun_out/abenv; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
run() { tag=$1; rm -rf $out/tr_$tag
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$tag -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --sustain-seconds 0 > $out/$tag.json 2> $out/$tag.err
  python3 tools/trace_layers.py $out/tr_$tag > $out/$tag.txt 2>&1; rm -rf $out/tr_$tag; }
for rep in 1 2; do ( run default_$rep ); ( export "$@"; run env_$rep ); done
for t in default_1 env_1 default_2 env_2; do echo "== $t"; grep "conv_\|sum of" $out/$t.txt | tail -8 | cut -c1-120; done
