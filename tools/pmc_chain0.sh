#!/bin/bash
# SQ counters of hr_chain0_kernel (chain_bench): where do its cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pc
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  d=gpurun_out/pc/$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 tools/chain_bench.py 64 > gpurun_out/pc.log 2>&1 || { tail -5 gpurun_out/pc.log; continue; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        if "hr_chain" not in k:
            continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (s, n) in sorted(d.items()):
        print("   %-28s %14.0f  (avg per launch over %d)" % (c, s / n, n))
PY
rm -rf gpurun_out/pc
