#!/bin/bash
# tools/scratch/ab_env.sh VAR v1 v2 ...: per-dispatch kernel times of one bench image for each value of environment variable VAR
# (alternating, two repetitions, ONE box) -> a table like ab_lib.sh's
set -u
var=$1; shift
root=$(pwd)
out=$root/gpurun_out/abenv
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
run() {
  val=$1; tag=$2
  rm -rf $out/tr_$tag
  export $var=$val
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$tag -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --sustain-seconds 0 > $out/$tag.json 2> $out/$tag.err
  python3 tools/trace_layers.py $out/tr_$tag > $out/$tag.txt 2>&1
  rm -rf $out/tr_$tag
}
for rep in 1 2; do
  for v in "$@"; do ( run $v v${v}_$rep ); done
done
python3 - "$@" <<'PY'
import sys,os
out=os.path.join(os.getcwd(),"gpurun_out","abenv")
names=sys.argv[1:]
tab={}
for n in names:
    for rep in (1,2):
        rows=[l for l in open(os.path.join(out,"v%s_%d.txt"%(n,rep))) if " us " in l and "grid" in l and ("conv_" in l or "deconv" in l)]
        for k,l in enumerate(rows):
            us=float(l.split()[0]); kn=l.split("shf::")[-1].strip()[:44] if "shf::" in l else l.split()[-1][:44]
            tab.setdefault((k,kn),{}).setdefault(n,[]).append(us)
print("%-3s %-46s"%("#","kernel")+"".join("%12s"%n for n in names))
tot={n:0.0 for n in names}
for (k,kn),d in sorted(tab.items()):
    print("%-3d %-46s"%(k,kn)+"".join("%12.1f"%(sum(d.get(n,[0]))/max(1,len(d.get(n,[])))) for n in names))
    for n in names: tot[n]+=sum(d.get(n,[0]))/max(1,len(d.get(n,[])))
print("%-50s"%"sum of conv kernels"+"".join("%12.1f"%tot[n] for n in names))
PY
