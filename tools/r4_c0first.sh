cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph or e2e" 2>&1 | tail -3 || exit 1
for F in 0 2 3; do
CHECKERPOSE_AMD_CHAIN0_FIRST=$F python bench.py --no-extras --no-cpu-baseline --no-breakdown --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('chain0_first=$F: %.3f ms  %.0f crops/s' % (d['ms_per_step'], d['value']))"
done
for F in 0 2; do
CHECKERPOSE_AMD_CHAIN0_FIRST=$F python bench.py --batch 64 --no-extras --no-cpu-baseline --no-breakdown --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=64 chain0_first=$F: %.3f ms  %.0f crops/s' % (d['ms_per_step'], d['value']))"
done
CHECKERPOSE_AMD_CHAIN0_FIRST=2 bash tools/r4_timeline.sh
