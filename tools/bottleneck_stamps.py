"""In-kernel phase clock of the fused Bottleneck (workgroup 0): needs a -DCP_DEBUG_KNOBS build of bottleneck_fused.hip linked into a
second .so:  CHECKERPOSE_AMD_LIB=.../libcheckerpose_hip_knobs.so python tools/bottleneck_stamps.py [B]"""
import ctypes as C
import runpy
import sys
import numpy as np
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "256", sys.argv[2] if len(sys.argv) > 2 else "256"]
runpy.run_path("tools/bottleneck_bench.py")          # the stamps are those of the LAST launch (argv[2]: 256 = identity, 64 = projection shortcut)
from checkerpose_amd import _abi
raw = C.CDLL(_abi.load()._name)
names = ["stage (wait halo, LDS writes)", "barrier", "conv1 + epilogue", "barrier", "conv2 + epilogue", "barrier", "conv3 + residual", "barrier",
         "store", "barrier"]
buf = (C.c_ulonglong * 80)()
fn = raw.cp_debug_bottleneck_stamps
fn.restype = C.c_int
assert fn(buf) == 0
a = np.array(list(buf), dtype=np.float64).reshape(8, 10)
tot = a.sum(1)
print("clock ticks (100 MHz) per wave over workgroup 0's tiles: %s" % np.round(tot).astype(int).tolist())
for k, nme in enumerate(names):
    print("  %-30s mean %8.0f (%.1f %%)   min %8.0f max %8.0f" % (nme, a[:, k].mean(), 100 * a[:, k].mean() / tot.mean(), a[:, k].min(), a[:, k].max()))
