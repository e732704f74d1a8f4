# interleaved A/B of bench.py under different environments (the boxes drift by +-2 % within minutes: single runs cannot rank variants)
# AB_SCRIPT / AB_ARGS: another bench script and its fixed arguments (bench_train.py: AB_ARGS="--no-breakdown --no-cpu-baseline --no-deterministic-cost")
# usage: bash tools/ab_env.sh <reps> "<ENV=.. ENV=..>" "<ENV=..>" ... [-- bench args]     ("-" = no extra environment)
cd $GRAFT_REPO_ROOT
reps=$1; shift
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done
[ "$1" = "--" ] && shift
mkdir -p gpurun_out; : > gpurun_out/ab_env.txt
for i in $(seq $reps); do
  k=0
  for v in "${vars[@]}"; do
    e="$v"; [ "$v" = "-" ] && e=""
    env $e python ${AB_SCRIPT:-bench.py} ${AB_ARGS:---no-extras --no-cpu-baseline --no-breakdown --steps 60 --warmup 5} "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('v$k %.3f' % d['ms_per_step'])" | tee -a gpurun_out/ab_env.txt
    k=$((k+1))
  done
done
python - "${vars[@]}" <<'PY'
import collections,statistics,sys
d=collections.defaultdict(list)
for l in open('gpurun_out/ab_env.txt'):
    k,v=l.split(); d[k].append(float(v))
for k,v in sorted(d.items()): print(k, sys.argv[1+int(k[1:])], 'min %.3f median %.3f max %.3f n=%d' % (min(v), statistics.median(v), max(v), len(v)))
PY
