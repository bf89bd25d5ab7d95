#!/bin/bash
# tools/scratch/ab_lib.sh NAME [NAME ...]: per-dispatch kernel times of one bench image for each variants/NAME.so ("default" = the
# shipped library), alternating, on ONE box -> gpurun_out/ab/<name>_<rep>.txt (tools/trace_layers.py) + a one-line summary each
set -u
root=$(pwd)
out=$root/gpurun_out/ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
run() {
  name=$1; tag=$2
  rm -rf $out/tr_$tag
  if [ "$name" != "default" ]; then export SHF_LIB=$root/variants/$name.so; else unset SHF_LIB; fi
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$tag -- python3 bench.py --steps 3 --warmup 1 --no-events --no-cpu-baseline --no-latency --no-reduced --no-calib --no-mixed --sustain-seconds 0 > $out/$tag.json 2> $out/$tag.err
  python3 tools/trace_layers.py $out/tr_$tag > $out/$tag.txt 2>&1
  rm -rf $out/tr_$tag
}
for rep in 1 2; do
  for n in "$@"; do ( run $n ${n}_$rep ); done
done
python3 - "$@" <<'PY'
import sys,re,os
out=os.path.join(os.getcwd(),"gpurun_out","ab")
names=sys.argv[1:]
tab={}
for n in names:
    for rep in (1,2):
        rows=[l for l in open(os.path.join(out,"%s_%d.txt"%(n,rep))) if " us " in l and "grid" in l and ("conv_" in l or "deconv" in l)]
        for k,l in enumerate(rows):
            us=float(l.split()[0]); kn=l.split("shf::")[-1].strip()[:44] if "shf::" in l else l.split()[-1][:44]
            tab.setdefault((k,kn),{}).setdefault(n,[]).append(us)
print("%-3s %-46s"%("#","kernel")+"".join("%14s"%n for n in names))
tot={n:0.0 for n in names}
for (k,kn),d in sorted(tab.items()):
    if not kn.startswith("conv") and "deconv" not in kn: continue
    print("%-3d %-46s"%(k,kn)+"".join("%14.1f"%(sum(d.get(n,[0]))/max(1,len(d.get(n,[])))) for n in names))
    for n in names: tot[n]+=sum(d.get(n,[0]))/max(1,len(d.get(n,[])))
print("%-50s"%"sum of conv kernels"+"".join("%14.1f"%tot[n] for n in names))
PY
