cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do
CHECKERPOSE_AMD_FUSE_DEFER=$v python bench.py --no-extras --no-cpu-baseline --no-breakdown --steps 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('defer=$v: %.3f ms  %.0f crops/s (%s)' % (d['ms_per_step'], d['value'], d['config']['launch']))"
done; done
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "e2e_fp32_vs_golden or graph_capture or contract or batch_slices or img_feats" 2>&1 | tail -3
