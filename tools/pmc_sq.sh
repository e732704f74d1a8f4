#!/bin/bash
# usage: bash tools/pmc_sq.sh [bench.py args, e.g. --workload lm13_n4096]      (SQ_SCRIPT=bench_train.py: the training step instead)
# SQ counters of every kernel of the default forward (eager replay, 2 steps): MFMA busy share, LDS activity / bank conflicts, VALU per
# MFMA instruction -> gpurun_out/sq_counters.csv (one row per kernel, averages per launch).  Two --pmc passes (counter groups).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/psq
export CHECKERPOSE_AMD_GRAPH=0 CHECKERPOSE_AMD_TRAIN_GRAPH=none
script=${SQ_SCRIPT:-bench.py}
extra="--no-cpu-baseline --no-breakdown"; if [ "$script" = "bench.py" ]; then extra="$extra --no-extras"; fi
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS"; do
  d=gpurun_out/psq/$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $script --steps 2 --warmup 1 $extra "$@" > gpurun_out/psq.log 2>&1 || { tail -5 gpurun_out/psq.log; echo "(counter set skipped: $set)"; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/psq/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith(("void at::", "__amd")) or "pack_" in k:
            continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
cols = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
        "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_LDS"]
with open("gpurun_out/sq_counters.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + cols + ["valu_per_mfma", "lds_conflict_share", "mfma_busy_per_busy_cycle"])
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0]):
        v = {c: (d[c][0] / d[c][1] if c in d and d[c][1] else 0.0) for c in cols}
        n = max((d[c][1] for c in d), default=0)
        w.writerow([k, n] + ["%.0f" % v[c] for c in cols] +
                   ["%.2f" % (v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"]) if v["SQ_INSTS_MFMA"] else "",
                    "%.3f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]) if v["SQ_LDS_IDX_ACTIVE"] else "",
                    "%.2f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]) if v["SQ_BUSY_CYCLES"] else ""])
PY
rm -rf gpurun_out/psq
head -14 gpurun_out/sq_counters.csv | cut -c1-220
