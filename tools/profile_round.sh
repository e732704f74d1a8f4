#!/bin/bash
# Round-end evidence: kernel stats of the default bench (graph + lanes), of the eager replay, and the two PMC passes.
# Usage on the GPU box:  bash tools/profile_round.sh   (outputs under gpurun_out/prof_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_stats gpurun_out/prof_lanes gpurun_out/prof_fetch gpurun_out/prof_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lanes -- python3 bench.py --steps 5 --no-cpu-baseline --no-breakdown --no-extras > gpurun_out/prof_lanes.log 2>&1 || exit 1
export CHECKERPOSE_AMD_GRAPH=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 5 --no-cpu-baseline --no-breakdown --no-extras > gpurun_out/prof_stats.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-extras > gpurun_out/prof_fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-extras > gpurun_out/prof_write.log 2>&1 || exit 1
# keep only the small csv files (the merge back is capped)
find gpurun_out/prof_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
ls -la gpurun_out/prof_*/*/ | head -30
