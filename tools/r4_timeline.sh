cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --no-extras --no-cpu-baseline --no-breakdown "$@" > gpurun_out/tl.log 2>&1
N=$(python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/tl.log') if l.startswith('{')][-1]); print(d['config']['launch'].split()[3])")
python3 tools/timeline.py gpurun_out/tl $N > gpurun_out/timeline_r4.txt
head -3 gpurun_out/timeline_r4.txt
rm -rf gpurun_out/tl
