#!/bin/bash
# step span / idle of the default forward under rocprofv3 for two settings of one env knob: bash tools/ab_timeline.sh VAR
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 1 0; do
  rm -rf gpurun_out/tl
  env_line="$1=$v"
  export $1=$v
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --no-extras --no-cpu-baseline --no-breakdown > gpurun_out/tl.log 2>&1
  N=$(python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/tl.log') if l.startswith('{')][-1]); print(d['config']['launch'].split()[3])")
  python3 tools/timeline.py gpurun_out/tl $N > gpurun_out/tlx.txt
  echo "$env_line launches=$N $(sed -n 1,2p gpurun_out/tlx.txt | tr '\n' ' ')"
done; done
rm -rf gpurun_out/tl
