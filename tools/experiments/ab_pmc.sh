#!/bin/bash
# clock / matrix-pipe occupancy of the conv kernels, shipped library vs variants/$1.so (one rocprofv3 counter pass each)
set -u
root=$(pwd); out=$root/gpurun_out/abpmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
one() {
  rm -rf $out/pmc_$1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_$1 -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib > $out/pmc_$1.log 2>&1
  ( cd tools && python3 pmc_clock_table.py $out/pmc_$1 200 > $out/clock_$1.txt 2>&1 )
  find $out/pmc_$1 -name "*.db" -delete; find $out/pmc_$1 -name "*.csv" -delete
  echo "== $1"; head -12 $out/clock_$1.txt
}
( one new )
( export SHF_LIB=$root/variants/$1.so; one $1 )
( one new2 )
