#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...]: libshf_hip.so with one source rebuilt under extra
# flags, written to variants/NAME.so (git-ignored, travels with gpurun); run with SHF_LIB=variants/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p variants
obj=variants/$name.$src.o
extra=""
case $src in tail.hip|merge.hip|pre.hip) extra="-ffp-contract=off";; esac
# SRC_OVERRIDE=/path/to/other/version.hip builds that file in place of csrc/$src (A/B against an older revision)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Ismallhardface_amd/csrc "$@" -c ${SRC_OVERRIDE:-smallhardface_amd/csrc/$src} -o $obj
objs=""
for o in smallhardface_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" != "$src.o" ]; then objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $objs $obj
rm -f $obj
echo variants/$name.so
