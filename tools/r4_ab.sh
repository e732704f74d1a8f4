# usage: bash tools/r4_ab.sh <tag> <libA.so|-> <bench script + args ...>   A/B of the in-tree library against libA, two alternations
cd $GRAFT_REPO_ROOT
tag=$1; A=$2; shift 2
mkdir -p gpurun_out/ab
for i in 1 2; do
  if [ "$A" != "-" ]; then CHECKERPOSE_AMD_LIB=$PWD/$A timeout -k 10 200 python "$@" >> gpurun_out/ab/${tag}_A.log 2>&1; fi
  timeout -k 10 200 python "$@" >> gpurun_out/ab/${tag}_B.log 2>&1
done
grep -h "Cin\|us\b" gpurun_out/ab/${tag}_A.log | sed 's/^/A: /'; grep -h "Cin\|us\b" gpurun_out/ab/${tag}_B.log | sed 's/^/B: /'
