cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "e2e or graph or contract" 2>&1 | tail -3 || exit 1
for b in 1 8 32; do
for F in -1 1; do
CHECKERPOSE_AMD_FUSE_OUT_MIN_BATCH=$F python bench.py --batch $b --no-extras --no-cpu-baseline --no-breakdown --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$b fuse_out_min=$F: %.3f ms  (%s)' % (d['ms_per_step'], d['config']['launch']))"
done; done
