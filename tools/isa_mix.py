"""Per kernel of a .hip file: static instruction mix of the code between its first and last MFMA (the hot loops): MFMA, VALU,
SALU, LDS, VMEM counts and the top VALU opcodes.  Usage: python tools/isa_mix.py file.hip [kernel-name-substring]"""
import collections, re, subprocess, sys, tempfile
f = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f, "-o", tmp.name],
                       capture_output=True, text=True)
    txt = open(tmp.name).read()
for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if want not in name:
        continue
    L = [l.strip() for l in m.group(0).split("\n") if l.strip() and not l.strip().startswith(";")]
    L = [l for l in L if not l.startswith(".") and not l.split()[0].endswith(":")]
    idx = [i for i, l in enumerate(L) if l.startswith("v_mfma")]
    if not idx:
        continue
    body = L[idx[0]:idx[-1] + 1]
    c = collections.Counter()
    for l in body:
        op = l.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_"))
             else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
        c[k] += 1
    top = collections.Counter(l.split()[0] for l in body if l.startswith("v_") and not l.startswith("v_mfma")).most_common(8)
    print("%-100s %s\n      %s" % (name[:100], dict(c), top))
