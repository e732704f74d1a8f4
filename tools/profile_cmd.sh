#!/bin/bash
# rocprofv3 evidence for ONE bench command: kernel stats (graph + lanes), kernel stats of the eager replay, and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only -- as MI355X_MICROARCH.md §HBM prescribes).
# Usage on the GPU box:  bash tools/profile_cmd.sh <tag> <script.py> [bench args...]
#   e.g. bash tools/profile_cmd.sh lm13_n4096_bf16_b32 bench.py --workload lm13_n4096
#        bash tools/profile_cmd.sh train_bf16_b32 bench_train.py
# Outputs gpurun_out/prof_<tag>_{lanes,stats,fetch,write}/ (only the small csv files are kept); then, in the build container:
#   python profiles/summarize.py gpurun_out/prof_<tag>_stats gpurun_out/prof_<tag>_fetch gpurun_out/prof_<tag>_write profiles/r03_<tag>
# The program itself follows `--` (no env / bash -c wrappers).  Environment knobs are exported in this shell instead.
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
extra="--no-cpu-baseline --no-breakdown"
if [ "$script" = "bench.py" ]; then extra="$extra --no-extras"; fi
P=gpurun_out/prof_$tag
rm -rf ${P}_lanes ${P}_stats ${P}_fetch ${P}_write
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_lanes -- python3 $script --steps 5 $extra "$@" > ${P}_lanes.log 2>&1 || exit 1
export CHECKERPOSE_AMD_GRAPH=0 CHECKERPOSE_AMD_TRAIN_GRAPH=none
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats -- python3 $script --steps 5 $extra "$@" > ${P}_stats.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${P}_fetch -- python3 $script --steps 2 --warmup 1 $extra "$@" > ${P}_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${P}_write -- python3 $script --steps 2 --warmup 1 $extra "$@" > ${P}_write.log 2>&1 || exit 1
find ${P}_lanes ${P}_stats ${P}_fetch ${P}_write -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
tail -2 ${P}_lanes.log
