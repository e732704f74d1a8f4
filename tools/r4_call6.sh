cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mlp_ or contract" > gpurun_out/r4g/tests.log 2>&1 || { tail -40 gpurun_out/r4g/tests.log; exit 1; }
tail -3 gpurun_out/r4g/tests.log
python bench.py --no-extras --no-cpu-baseline --steps 10 > gpurun_out/r4g/bench_lmo.json 2> gpurun_out/r4g/bench_lmo.err
CHECKERPOSE_AMD_LIB=$PWD/build/lib_mlp_a.so python bench.py --no-extras --no-cpu-baseline --steps 10 > gpurun_out/r4g/bench_lmo_nofuse.json 2>> gpurun_out/r4g/bench_lmo.err
python bench.py --workload lm13_n4096 --no-extras --no-cpu-baseline --steps 10 > gpurun_out/r4g/bench_lm13.json 2> gpurun_out/r4g/bench_lm13.err
CHECKERPOSE_AMD_LIB=$PWD/build/lib_mlp_a.so python bench.py --workload lm13_n4096 --no-extras --no-cpu-baseline --steps 10 > gpurun_out/r4g/bench_lm13_nofuse.json 2>> gpurun_out/r4g/bench_lm13.err
python - <<'PY'
import json
for f in ("bench_lmo","bench_lmo_nofuse","bench_lm13","bench_lm13_nofuse"):
    try:
        d=json.loads([l for l in open("gpurun_out/r4g/%s.json"%f) if l.startswith("{")][0])
        print(f, d["value"], d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if k in ("mlp_fused","gemm_rows","edge_tiled","edge_fused","conv_igemm")})
    except Exception as e: print(f, "ERR", e)
PY
