# kernel timeline of the LAST training step of `bench_train.py` under rocprofv3 (graph replay: forward + backward + optimizer):
# span, idle, concurrency histogram, then every kernel's start / duration -> gpurun_out/timeline_train.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tlt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlt -- python3 bench_train.py --steps 4 --warmup 3 --no-breakdown --no-cpu-baseline --no-deterministic-cost "$@" > gpurun_out/tlt.log 2>&1
python3 - <<'P' > gpurun_out/timeline_train.txt
import csv, glob, collections
f = glob.glob("gpurun_out/tlt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the steps are separated by the host-side loss read: cut at the last zero fill of the gradient arena's forward start = the
# last-but-one occurrence of the per-step first kernel (pack_batch of the forward half)
first = [i for i, r in enumerate(rows) if "pack_batch_kernel" in r["Kernel_Name"]]
# four pack_batch launches per step (views + packs, forward + backward): the step starts at the 4th from the end
start = first[-4]
last = rows[start:]
t0 = last[0]["s"]
print("step span %.1f us, %d kernels" % ((max(r["e"] for r in last) - t0) / 1e3, len(last)))
ev = sorted([(r["s"], 1) for r in last] + [(r["e"], -1) for r in last])
cur, prev, idle, hist = 0, t0, 0, collections.Counter()
for t, d in ev:
    hist[cur] += t - prev
    cur += d; prev = t
print("concurrency histogram (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
print("sum of kernel durations %.1f us" % (sum(r["e"] - r["s"] for r in last) / 1e3))
short = lambda s: s.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
for r in last:
    print("%9.1f %8.1f  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, short(r["Kernel_Name"])))
P
head -3 gpurun_out/timeline_train.txt
rm -rf gpurun_out/tlt
