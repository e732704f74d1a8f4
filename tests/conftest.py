import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from checkerpose_amd import _abi
    return _abi.load()


DP_TEST_MAX_RANKS = int(os.environ.get("CHECKERPOSE_DP_TEST_MAX_RANKS", "4"))


def dp_child_plan(ndev):
    """(backend, ranks) of the data-parallel child job: a node with several GPUs runs one rank per GPU over RCCL ("nccl" IS RCCL on
    ROCm: the gradient all-reduce over xGMI that BASELINE config #3 names), up to DP_TEST_MAX_RANKS = 4 -- the pool's process guard
    allows a job six GPU processes at once (the ranks + this pytest process), and the 8-rank case is the driver's to launch
    (`CHECKERPOSE_DP_TEST_MAX_RANKS=8` on a node without that guard); a 1-GPU box rehearses the same code with two ranks sharing the
    GPU over gloo"""
    return ("nccl", min(int(ndev), DP_TEST_MAX_RANKS)) if int(ndev) >= 2 else ("gloo", 2)


def pytest_sessionstart(session):
    """`-m gpu` sessions on a GPU box: start the 2-rank data-parallel child job of tests/test_gpu_zz_dp.py NOW, before anything in
    this process touches the GPU (torch.cuda.device_count() does not initialise it on this image)."""
    import subprocess
    import tempfile
    cfg = session.config
    cfg._dp_child = None
    mexpr = cfg.getoption("-m") or ""
    if "gpu" not in mexpr or "not gpu" in mexpr or cfg.getoption("collectonly", False):
        return
    kexpr = cfg.getoption("-k") or ""
    if kexpr and "dp" not in kexpr:
        return
    try:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < 1:
            return
    except Exception:
        return
    backend, nproc = dp_child_plan(ndev)
    cfg._dp_plan = (backend, nproc)
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    log = tempfile.NamedTemporaryFile(prefix="dp_child_", suffix=".log", delete=False)
    env = dict(os.environ, CHECKERPOSE_BENCH_BACKEND=backend, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # one child shell, two jobs one after the other (never more than two extra processes on the GPU): the 2-rank training step, then
    # `python bench.py --gpus 2` launching its OWN two ranks (no torchrun environment: bench.self_launch), a tiny batch
    env_clean = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = ("%s -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port %d %s; rc=$?; "
           "%s %s --gpus %d --batch 8 --steps 3 --warmup 2 --no-extras --no-cpu-baseline --no-breakdown; echo BENCH_SELF_LAUNCH_RC=$?; exit $rc"
           % (sys.executable, nproc, port, os.path.join(ROOT, "tests", "dp_step_child.py"), sys.executable, os.path.join(ROOT, "bench.py"), nproc))
    proc = subprocess.Popen(["bash", "-c", cmd], stdout=log, stderr=subprocess.STDOUT, env=env_clean, cwd=ROOT)
    cfg._dp_child = (proc, log.name)


def pytest_sessionfinish(session, exitstatus):
    job = getattr(session.config, "_dp_child", None)
    if job is not None and job[0].poll() is None:
        job[0].kill()
