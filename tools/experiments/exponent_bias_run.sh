#!/bin/bash
# Would a LOOSE bound on a unit's max |x| do for the activation exponent?  (the next lever.)
# Variant builds with conv_act_exponent() biased by k (e' = max(0, e - k), i.e. the maximum overestimated 2^k times):
#   for k in 4 8 12: copy csrc/{conv_f16x3.hip,conv_common.h,shf_internal.h} to a scratch dir, replace
#   "13 - ((int)(b >> 23) - 127)" by "13 - k - (...)" in conv_common.h, compile conv_f16x3.hip there, link with the other
#   objects into variants/bias$k.so
# then, on the GPU box, the magnitude tests, the C1 every-anchor test and the full-size tests under each library:
cd ${GRAFT_REPO_ROOT:-.}
for k in 0 4 8 12; do
  if [ $k = 0 ]; then unset SHF_LIB; else export SHF_LIB=$PWD/variants/bias$k.so; fi
  echo "== exponent bias $k"
  python -m pytest tests/test_gpu_magnitudes.py "tests/test_gpu_parity.py::test_c1_512_level_vs_oracle" tests/test_gpu_fullsize.py -q -m gpu -k "not multi and not rank" 2>&1 | tail -3
done
