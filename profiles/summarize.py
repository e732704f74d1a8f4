#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_*) into the small per-round summaries committed here.
  python profiles/summarize.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write profiles/r01_bf16_b128
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read -> doubled."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def newest(pattern):
    """gpurun merges a call's outputs INTO gpurun_out/ (older runs of the same tag stay beside them): take the newest file"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


def main(stats_dir, fetch_dir, write_dir, out_prefix):
    rows = list(csv.DictReader(open(newest(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0])))
    ours = [r for r in rows if not r["Name"].startswith(("void at::", "__amd_rocclr"))]
    tot = sum(float(r["TotalDurationNs"]) for r in ours)
    pmc = {}
    for key, d, mult in (("fetch", fetch_dir, 2.0), ("write", write_dir, 1.0)):
        acc = defaultdict(lambda: [0.0, 0])
        f = newest(os.path.join(d, "*", "*counter_collection.csv"))
        if f:
            for r in csv.DictReader(open(f[0])):
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]) * 1024.0 * mult
                a[1] += 1
        pmc[key] = acc
    with open(out_prefix + "_kernel_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct_of_our_kernels", "avg_hbm_read_MB(FETCH_SIZE*2)", "avg_hbm_write_MB"])
        for r in ours:
            n = r["Name"]
            fr, fw = pmc["fetch"].get(n), pmc["write"].get(n)
            w.writerow([short(n), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        "%.2f" % (100 * float(r["TotalDurationNs"]) / tot),
                        "%.3f" % (fr[0] / fr[1] / 1e6) if fr and fr[1] else "", "%.3f" % (fw[0] / fw[1] / 1e6) if fw and fw[1] else ""])
    print(open(out_prefix + "_kernel_summary.csv").read())


def sq_missing(bench_json, sq_csv):
    """kernel symbols of a bench line (`kernel_symbols`: the launches the timed step spends its time in) that have NO row in an SQ
    counter file (tools/pmc_sq.sh) -- round 5 committed counters of a build that was no longer the launched one.  The bench line
    collapses the chain instances to `hr_chain_kernel<Cfg, true>`: any `hr_chain_kernel<ChainCfg<...>, true>` row covers it."""
    import json
    txt = open(bench_json).read()
    line = [ln for ln in txt.splitlines() if ln.startswith("{")][-1]
    syms = list(json.loads(line).get("kernel_symbols", {}))
    have = [r["kernel"] for r in csv.DictReader(open(sq_csv))]

    def covered(sym):
        for part in sym.split(" + "):                                  # a call that issues several launches: "(a<..>) + (b<..>)"
            part = part.strip()
            if part.startswith("(") and part.endswith(")"):
                part = part[1:-1]
            base = part.split("<")[0]
            if "<Cfg" in part:
                tail = part.split("<Cfg", 1)[1]                          # e.g. ", true>"
                ok = any(h.startswith(base + "<ChainCfg<") and h.endswith(tail) for h in have)
            else:
                ok = any(h == part or h.startswith(part) or part.startswith(h) for h in have)   # (the SQ file cuts names at 64 characters)
            if not ok:
                return False
        return True
    return [s_ for s_ in syms if not covered(s_)]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "check":                     # python profiles/summarize.py check <bench line .json> <sq counters .csv>
        miss = sq_missing(sys.argv[2], sys.argv[3])
        if miss:
            sys.exit("SQ counter file %s has no row for: %s" % (sys.argv[3], "; ".join(miss)))
        print("every kernel symbol of %s has SQ counters in %s" % (sys.argv[2], sys.argv[3]))
    else:
        main(*sys.argv[1:5])
