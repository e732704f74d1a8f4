#!/bin/bash
# GPU-only replay span A/B of two library builds, interleaved: bash tools/ab_span.sh <libA.so> <libB.so>
for r in 1 2 3; do
  for L in $1 $2; do
    echo "== $L"; CHECKERPOSE_AMD_LIB=$PWD/$L timeout -k 10 200 python3 tools/b1_lanes_probe.py 0 2>&1 | grep "B="
  done
done
