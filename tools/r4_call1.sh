cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
( time python bench.py ) > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
for b in 32 64 128 256; do python bench.py --workload lm13_n4096 --batch $b --no-extras --no-cpu-baseline --steps 10 > gpurun_out/r4a/lm13_b$b.json 2> gpurun_out/r4a/lm13_b$b.err; done
CHECKERPOSE_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-extras --no-cpu-baseline --no-breakdown --steps 10 > gpurun_out/r4a/bench_g2.json 2> gpurun_out/r4a/bench_g2.err
CHECKERPOSE_BENCH_BACKEND=gloo python bench_train.py --gpus 2 --no-cpu-baseline --no-breakdown --steps 10 > gpurun_out/r4a/train_g2.json 2> gpurun_out/r4a/train_g2.err
tail -c 600 gpurun_out/r4a/bench_default.err
