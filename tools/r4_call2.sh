cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python tools/margin_probe.py --npoint 512 --batch 8 > gpurun_out/r4b/margin_512.json 2> gpurun_out/r4b/margin_512.err
python tools/margin_probe.py --npoint 512 --batch 8 --selection per_crop > gpurun_out/r4b/margin_512_percrop.json 2>> gpurun_out/r4b/margin_512.err
python tools/margin_probe.py --npoint 4096 --lm --batch 4 > gpurun_out/r4b/margin_4096.json 2> gpurun_out/r4b/margin_4096.err
python -m pytest tests/test_gpu_parity.py -x -q -k "edgeconv or edge_ or contract" > gpurun_out/r4b/edge_tests.log 2>&1
for i in 1 2; do
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so python tools/edge_tiled_bench.py 32 >> gpurun_out/r4b/tiled_base.log 2>&1
python tools/edge_tiled_bench.py 32 >> gpurun_out/r4b/tiled_new.log 2>&1
CHECKERPOSE_AMD_LIB=$PWD/build/lib_base.so python tools/edge_bench.py 256 >> gpurun_out/r4b/fused_base.log 2>&1
python tools/edge_bench.py 256 >> gpurun_out/r4b/fused_new.log 2>&1
done
tail -3 gpurun_out/r4b/edge_tests.log
