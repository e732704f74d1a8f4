#!/bin/bash
# interleaved A/B of a micro-benchmark over several library builds: bash tools/ab_libs.sh "<python command>" lib1.so lib2.so ...
CMD=$1; shift
for r in 1 2 3; do
  for L in "$@"; do
    echo "== $L"; CHECKERPOSE_AMD_LIB=$PWD/$L timeout -k 10 200 $CMD 2>&1 | grep -v amdgpu.ids
  done
done
