#!/bin/bash
# per-(kernel, grid) launch statistics of one bench command (eager replay): bash tools/prof_shapes.sh <tag> <script.py> [args...]
# -> gpurun_out/ps_<tag>.csv : kernel, grid (workgroups), launches, avg_us, total_ms -- which SHAPES of a kernel the time goes to
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$PS_GRAPH" ]; then export CHECKERPOSE_AMD_GRAPH=0 CHECKERPOSE_AMD_TRAIN_GRAPH=none; fi      # PS_GRAPH=1: profile the hipGraph replay
rm -rf gpurun_out/ps_$tag
extra="--no-cpu-baseline --no-breakdown"; if [ "$script" = "bench.py" ]; then extra="$extra --no-extras"; fi
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ps_$tag -- python3 $script --steps 5 $extra "$@" > gpurun_out/ps_$tag.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/ps_$tag/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:48]))
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    a = agg[(name, grid)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
iv.sort()
tail = iv[len(iv) // 2:]                      # second half of the run: steady-state steps only
busy, cur_s, cur_e, ksum = 0, tail[0][0], tail[0][1], 0
gaps = collections.defaultdict(lambda: [0, 0.0])
prev = None
for s_, e_, nm in tail:
    if prev is not None and s_ > prev[1]:
        g_ = gaps[(prev[2], nm)]
        g_[0] += 1
        g_[1] += (s_ - prev[1]) / 1e3
    if prev is None or e_ > prev[1]:
        prev = (s_, e_, nm)
    ksum += e_ - s_
    if s_ > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
span = max(e for _, e, _n in tail) - tail[0][0]
if "${PS_GAPS:-}":
    for (a_, b_), (n_, us_) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:int("${PS_GAPS:-0}")]:
        print("gap after %-48s before %-48s n=%5d avg_us=%7.1f tot_ms=%7.2f" % (a_, b_, n_, us_ / n_, us_ / 1e3))
print("second half of the trace: %d launches, span %.2f ms, GPU busy (union) %.2f ms, sum of kernel durations %.2f ms" % (len(tail), span / 1e6, busy / 1e6, ksum / 1e6))
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open("gpurun_out/ps_$tag.csv", "w") as o:
    o.write("kernel,workgroups,launches,avg_us,total_ms\n")
    for (name, grid), (n, us) in rows:
        o.write('"%s",%d,%d,%.2f,%.3f\n' % (name, grid, n, us / n, us / 1e3))
for (name, grid), (n, us) in rows[:${TOP:-40}]:
    print("%-70s wg=%6d n=%5d avg_us=%8.1f tot_ms=%8.2f" % (name[:70], grid, n, us / n, us / 1e3))
PY
rm -rf gpurun_out/ps_$tag
