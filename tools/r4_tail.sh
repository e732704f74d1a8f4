cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain_tail" > gpurun_out/tail_test.log 2>&1; rc=$?; tail -5 gpurun_out/tail_test.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chain or fuse_out or e2e or graph or contract" > gpurun_out/tail_test2.log 2>&1; rc=$?; tail -5 gpurun_out/tail_test2.log; [ $rc -eq 0 ] || exit 1
