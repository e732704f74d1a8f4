#!/bin/bash
# quick kernel stats (eager replay) of one bench command: bash tools/prof_quick.sh <outdir-tag> <script.py> [args...]; prints the top rows
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHECKERPOSE_AMD_GRAPH=0 CHECKERPOSE_AMD_TRAIN_GRAPH=none
rm -rf gpurun_out/pq_$tag
extra="--no-cpu-baseline --no-breakdown"; if [ "$script" = "bench.py" ]; then extra="$extra --no-extras"; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pq_$tag -- python3 $script --steps 5 $extra "$@" > gpurun_out/pq_$tag.log 2>&1 || exit 1
find gpurun_out/pq_$tag -type f ! -name "*kernel_stats.csv" -delete
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pq_$tag/*/*kernel_stats.csv")[0])))
for r in rows[:${TOP:-14}]:
    print("%-86s n=%5s avg_us=%9.1f tot_ms=%8.2f" % (r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
