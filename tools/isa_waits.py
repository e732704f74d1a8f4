"""Per kernel: how many MFMAs sit directly behind a full `s_waitcnt lgkmcnt(0)` / `vmcnt(0)` (no other MFMA in between)?
A high share means hipcc sank the operand loads to their use: every MFMA pays an LDS / L2 latency (pin them with
__builtin_amdgcn_sched_barrier).  Usage: python tools/isa_waits.py [file.hip ...]   (compiles to gfx950 assembly)"""
import glob, os, re, subprocess, sys, tempfile
files = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "checkerpose_amd", "csrc", "*.hip")))
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f, "-o", tmp.name],
                           capture_output=True, text=True)
        if r.returncode:
            print(f, "compile failed"); continue
        txt = open(tmp.name).read()
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M):
        L = [l.strip() for l in m.group(0).split("\n") if l.strip() and not l.strip().startswith(";")]
        mf = [i for i, l in enumerate(L) if l.startswith("v_mfma")]
        if not mf:
            continue
        lg = vm = 0
        for i in mf:
            j = i - 1
            while j >= 0 and not L[j].startswith("v_mfma"):
                if L[j].startswith("s_waitcnt"):
                    lg += "lgkmcnt(0)" in L[j]
                    vm += "vmcnt(0)" in L[j]
                    break
                j -= 1
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:90]
        print("%-28s %-90s mfma %4d  behind lgkmcnt(0) %4d  behind vmcnt(0) %4d" % (os.path.basename(f), name, len(mf), lg, vm))
