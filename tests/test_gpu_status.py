"""GPU (-m gpu): the device-side error channel (include/checkerpose_hip.h: cp_device_status).  The pipelined 64 x 64 HRNet chain
(hr_chain0p_kernel) hands rows between its waves over LDS counters with BOUNDED waits; a wait that runs out used to end in wrong
numbers silently.  Now the wave ORs CP_STATUS_CHAIN0_HANDOVER into the device's sticky status word and the drop-in model raises
where it synchronises (program build, invalidate(), check_device_status()).  The failure is forced in a child process that loads the
test-hook build (`make -C checkerpose_amd/csrc knobs`: hr_chain0.hip under -DCP_DEBUG_KNOBS) with CP_C0_FORCE_TIMEOUT=1 (every
hand-over wait gives up after one poll)."""
import os
import subprocess
import sys

import pytest
import torch

from checkerpose_amd import _abi
from tests.common import build_net, det_image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS_LIB = os.path.join(ROOT, "checkerpose_amd", "libcheckerpose_hip_knobs.so")

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from checkerpose_amd import _abi
from tests.common import build_net, det_image
torch.set_grad_enabled(False)
net = build_net(seed=1).cuda().eval()
net.set_compute_dtype("bf16")
img = det_image(64, seed=3).cuda()
net(img, None)                       # 64 crops: the per-crop chain launches (engine.CHAIN_MIN_BATCH = 40)
names = [c[2] for c in net.program_for(64).progs[0].calls]
assert any(n.startswith("hr_chain") for n in names), names[:20]
try:
    net.check_device_status()
    print("STATUS none")
except RuntimeError as e:
    print("STATUS raised:", str(e)[:160])
try:
    net.check_device_status()        # the word was cleared by the first look
    print("SECOND none")
except RuntimeError as e:
    print("SECOND raised")
"""


def test_device_status_stays_clear_in_the_shipped_build():
    torch.set_grad_enabled(False)
    net = build_net(seed=1).cuda().eval()
    net.set_compute_dtype("bf16")
    img = det_image(64, seed=3).cuda()
    for _ in range(3):               # eager, capture, replay
        net(img, None)
    net.check_device_status()        # no exception
    assert _abi.device_status(clear=False) == 0


def test_forced_handover_timeout_is_reported_not_swallowed():
    assert os.path.exists(KNOBS_LIB), "build it: make -C checkerpose_amd/csrc knobs (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, CHECKERPOSE_AMD_LIB=KNOBS_LIB, CP_C0_FORCE_TIMEOUT="1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "STATUS raised:" in out and "CP_STATUS_CHAIN0_HANDOVER" in out, out[-2000:]
    assert "SECOND none" in out, out[-2000:]
    env2 = dict(os.environ, CHECKERPOSE_AMD_LIB=KNOBS_LIB)          # the same build without the forced timeout: clean
    r2 = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env2, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "STATUS none" in r2.stdout, (r2.stdout + r2.stderr)[-2000:]
