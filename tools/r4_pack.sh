cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/pack_tests.log 2>&1; rc=$?; tail -3 gpurun_out/pack_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_env.sh 5 CHECKERPOSE_AMD_LIB=$PWD/build/lib_oldpack.so -
bash tools/ab_env.sh 3 CHECKERPOSE_AMD_LIB=$PWD/build/lib_oldpack.so - -- --workload lm13_n4096
