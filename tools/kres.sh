#!/bin/bash
# kernel resource usage of one csrc/*.hip: name | VGPRs | AGPRs | spills | scratch | occupancy | LDS   (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; shift
cd "$(dirname "$0")/../checkerpose_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -c $f -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'remark: (.*) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: print(' | '.join(cur))
        n=t.split(':',1)[1].strip()
        try: n=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',n],capture_output=True,text=True).stdout.strip()[:90]
        except Exception: pass
        cur=[n]
    elif any(t.startswith(k) for k in ('VGPRs:','AGPRs:','VGPR Spill','ScratchSize','Occupancy','LDS Size')):
        cur.append(t.replace(' [bytes/lane]','').replace(' [bytes/block]','').replace(' [waves/SIMD]',''))
if cur: print(' | '.join(cur))
"
rm -f /tmp/kres_$$.o
