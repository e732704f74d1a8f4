cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/act_tests.log 2>&1; rc=$?; tail -3 gpurun_out/act_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_env.sh 5 CHECKERPOSE_AMD_LIB=$PWD/build/lib_head.so -
bash tools/ab_env.sh 2 CHECKERPOSE_AMD_LIB=$PWD/build/lib_head.so - -- --dtype fp32 --batch 128 --steps 10
