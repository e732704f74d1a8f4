cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4z
( time python bench.py ) > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err
python bench.py --workload lm13_n4096 --no-extras > gpurun_out/r4z/bench_lm13.json 2> gpurun_out/r4z/bench_lm13.err
python bench.py --workload ycbv_rr21 --no-extras > gpurun_out/r4z/bench_ycbv.json 2> gpurun_out/r4z/bench_ycbv.err
python bench_train.py > gpurun_out/r4z/bench_train.json 2> gpurun_out/r4z/bench_train.err
tail -4 gpurun_out/r4z/bench_default.err
