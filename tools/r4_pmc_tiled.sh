cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_t
for c in WRITE_SIZE FETCH_SIZE; do
timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_t/$c -- python3 tools/edge_tiled_bench.py 256 > gpurun_out/pmc_t_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for c, mult in (("WRITE_SIZE", 1.0), ("FETCH_SIZE", 2.0)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/pmc_t/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "edgeconv" not in k: continue
            a = acc[k.replace("void ","").replace("(anonymous namespace)::","").split("(")[0]]; a[0] += float(r["Counter_Value"]) * 1024 * mult; a[1] += 1
    for k, (v, n) in acc.items(): print(c, k, "%.1f MB per launch (%d launches)" % (v / n / 1e6, n))
PY
rm -rf gpurun_out/pmc_t
