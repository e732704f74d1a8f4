#!/bin/bash
# small-batch latency A/B of two library builds, interleaved: bash tools/ab_lat.sh <libA.so> <libB.so> [batches]
A=$1; B=$2; BS=${3:-1,8}
for r in 1 2; do
  for L in $A $B; do
    echo "== $L"; CHECKERPOSE_AMD_LIB=$PWD/$L timeout -k 10 200 python3 tools/eval_cpu_timeline.py $BS 2>&1 | grep "B="
  done
done
