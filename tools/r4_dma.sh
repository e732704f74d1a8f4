cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or edgeconv or edge or mlp or contract" > gpurun_out/dma_tests.log 2>&1; rc=$?; tail -3 gpurun_out/dma_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_env.sh 4 CHECKERPOSE_AMD_LIB=$PWD/build/lib_gdma.so -
bash tools/ab_env.sh 4 CHECKERPOSE_AMD_LIB=$PWD/build/lib_gdma.so - -- --workload lm13_n4096
