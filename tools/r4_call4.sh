cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "edgeconv or edge_ or n4096 or contract" > gpurun_out/r4d/edge_tests.log 2>&1 || { tail -30 gpurun_out/r4d/edge_tests.log; exit 1; }
tail -3 gpurun_out/r4d/edge_tests.log
for i in 1 2; do
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so timeout -k 10 120 python tools/edge_tiled_bench.py 32 >> gpurun_out/r4d/tiled_base.log 2>&1
timeout -k 10 120 python tools/edge_tiled_bench.py 32 >> gpurun_out/r4d/tiled_new.log 2>&1
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so timeout -k 10 120 python tools/edge_bench.py 256 >> gpurun_out/r4d/fused_base.log 2>&1
timeout -k 10 120 python tools/edge_bench.py 256 >> gpurun_out/r4d/fused_new.log 2>&1
done
timeout -k 10 120 python tools/edge_tiled_bench.py 256 >> gpurun_out/r4d/tiled_new_b256.log 2>&1
grep Cin gpurun_out/r4d/*.log
