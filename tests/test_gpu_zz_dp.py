"""GPU: the data-parallel TRAINING step as the driver will launch it -- `python -m torch.distributed.run`, one process per rank,
real HIP training programs and hipGraph segments with the bucketed asynchronous all-reduce between them -- on the one GPU of
the test box (two ranks share it; backend gloo, since RCCL needs a device per rank).  The children are started by
tests/conftest.py at session start, BEFORE this process initialises the GPU (a process that has must not fork+exec on this
pool), and run beside the other GPU tests; this test only joins them.  What is checked lives in tests/dp_step_child.py."""
import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_dp_training_step_on_one_gpu(request):
    job = getattr(request.config, "_dp_child", None)
    if job is None:
        pytest.skip("the 2-rank child job is only started for `-m gpu` sessions on a box with a GPU")
    proc, log = job
    try:
        rc = proc.wait(timeout=900)
    except Exception:
        proc.kill()
        raise
    out = open(log).read()
    backend, nproc = request.config._dp_plan
    assert rc == 0, out[-4000:]
    assert "DP_STEP_OK world=%d" % nproc in out, out[-4000:]
    print("data-parallel step: %d ranks over %s" % (nproc, backend))


def test_bench_gpus_2_launches_its_own_ranks_on_the_gpu(request):
    """`python bench.py --gpus 2` with no torchrun environment on the GPU box (gloo: both ranks share the one GPU): the parent starts
    `torch.distributed.run --nproc-per-node 2` itself, rank 0's line says n_gpus 2 and names both ranks (the child job of
    tests/conftest.py runs it behind the training-step job; this test reads its output)."""
    import json
    job = getattr(request.config, "_dp_child", None)
    if job is None:
        pytest.skip("the child job is only started for `-m gpu` sessions on a box with a GPU")
    proc, log = job
    proc.wait(timeout=900)
    out = open(log).read()
    assert "BENCH_SELF_LAUNCH_RC=0" in out, out[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"ranks_seen"' in ln]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    backend, nproc = request.config._dp_plan
    assert d["n_gpus"] == nproc and d["ranks_seen"] == list(range(nproc)) and len(d["per_rank_ms"]) == nproc and d["backend"] == backend
    assert d["config"]["global_batch"] == 8 * nproc and d["value"] > 0
