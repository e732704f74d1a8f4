cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
for b in 1 8 32 64; do
  for mr in 32768 1; do
    CHECKERPOSE_AMD_MLP_FUSED_MIN_ROWS=$mr python bench.py --batch $b --no-extras --no-cpu-baseline --no-breakdown --steps 100 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=%d min_rows=%s: %.3f ms  %.0f crops/s  (%s)' % ($b, '$mr', d['ms_per_step'], d['value'], d['config']['launch']))"
  done
done
