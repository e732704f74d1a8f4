cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "edgeconv_tiled or n4096" > gpurun_out/r4c/tiled_tests.log 2>&1 || { tail -30 gpurun_out/r4c/tiled_tests.log; exit 1; }
tail -3 gpurun_out/r4c/tiled_tests.log
for i in 1 2; do
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so timeout -k 10 120 python tools/edge_tiled_bench.py 32 >> gpurun_out/r4c/tiled_base.log 2>&1
timeout -k 10 120 python tools/edge_tiled_bench.py 32 >> gpurun_out/r4c/tiled_new.log 2>&1
done
timeout -k 10 120 python tools/edge_tiled_bench.py 256 >> gpurun_out/r4c/tiled_new_b256.log 2>&1
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so timeout -k 10 120 python tools/edge_tiled_bench.py 256 >> gpurun_out/r4c/tiled_base_b256.log 2>&1
grep Cin gpurun_out/r4c/*.log
