cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv2x2" > gpurun_out/h2_tests.log 2>&1; rc=$?; tail -5 gpurun_out/h2_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "e2e or contract or graph or index2feat or refine" > gpurun_out/h2_tests2.log 2>&1; rc=$?; tail -3 gpurun_out/h2_tests2.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_env.sh 4 CHECKERPOSE_AMD_HALO2=0 -
bash tools/ab_env.sh 3 CHECKERPOSE_AMD_HALO2=0 - -- --workload lm13_n4096
