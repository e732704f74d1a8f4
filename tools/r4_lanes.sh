cd $GRAFT_REPO_ROOT
for b in 1 8; do
for L in 0 1 2 3 4; do
CHECKERPOSE_AMD_MAX_LANES=$L python bench.py --batch $b --no-extras --no-cpu-baseline --no-breakdown --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$b max_lanes=$L: %.3f ms' % d['ms_per_step'])"
done; done
