cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/h2p
export CHECKERPOSE_AMD_GRAPH=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h2p -- python3 bench.py --workload lm13_n4096 --steps 3 --no-extras --no-cpu-baseline --no-breakdown > gpurun_out/h2p.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/h2p/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "halo_s" in r["Name"] or "conv_igemm" in r["Name"] or "index2feat" in r["Name"]:
        print(r["Name"][:70], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
rm -rf gpurun_out/h2p
