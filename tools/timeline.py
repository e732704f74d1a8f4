"""Timeline of the last hipGraph replay from a rocprofv3 --kernel-trace CSV: per-kernel start/end, GPU idle gaps, concurrency."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if not r["Kernel_Name"].startswith(("void at::", "__amd_rocclr"))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
n = int(sys.argv[2])        # launches per step
last = rows[-n:]
t0 = last[0]["s"]
tot = (max(r["e"] for r in last) - t0) / 1e3
print("step span %.1f us, %d kernels" % (tot, len(last)))
# busy/idle
ev = sorted([(r["s"], 1) for r in last] + [(r["e"], -1) for r in last])
cur, prev, idle, hist = 0, t0, 0, collections.Counter()
for t, dlt in ev:
    hist[cur] += t - prev
    if cur == 0:
        idle += t - prev
    cur += dlt; prev = t
print("idle %.1f us; concurrency histogram (us):" % (idle / 1e3), {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
short = lambda s: s.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:46]
for r in last:
    print("%9.1f %8.1f  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, short(r["Kernel_Name"])))
