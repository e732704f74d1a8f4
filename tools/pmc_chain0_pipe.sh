cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_pipe_f gpurun_out/pmc_pipe_w
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pipe_f -- python3 tools/chain0_pipe_check.py checkerpose_amd/libcheckerpose_hip.so checkerpose_amd/libcheckerpose_hip_pipe.so > gpurun_out/pmc_pipe_f.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pipe_w -- python3 tools/chain0_pipe_check.py checkerpose_amd/libcheckerpose_hip.so checkerpose_amd/libcheckerpose_hip_pipe.so > gpurun_out/pmc_pipe_w.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for tag, mult in (("f", 2.0), ("w", 1.0)):
    f = glob.glob("gpurun_out/pmc_pipe_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "hr_chain0" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 512:
            acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]) * 1024 * mult / 1e6)
    for k, v in acc.items():
        print(tag, k, "launches %d  MB per launch: min %.1f median %.1f max %.1f" % (len(v), min(v), sorted(v)[len(v)//2], max(v)))
PY
find gpurun_out/pmc_pipe_f gpurun_out/pmc_pipe_w -type f ! -name "*counter_collection.csv" -delete
