#!/bin/bash
# One FETCH_SIZE pass of the eager replay (2 steps): per-kernel average HBM-side read bytes (x2 on gfx950) -> gpurun_out/fetch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf
CHECKERPOSE_AMD_GRAPH=0 timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pf -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-breakdown > gpurun_out/pf.log 2>&1
python3 - <<'PY' > gpurun_out/fetch.txt
import csv, glob, collections
f = glob.glob("gpurun_out/pf/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    a = acc[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]) * 1024 * 2; a[1] += 1
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    if not k.startswith(("void at::", "__amd")):
        print("%-70s n=%4d avg_read_MB=%9.2f" % (k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70], n, s / n / 1e6))
PY
rm -rf gpurun_out/pf
head -12 gpurun_out/fetch.txt
