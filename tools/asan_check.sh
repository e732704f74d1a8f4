#!/bin/bash
# Host side of the C-ABI library under AddressSanitizer + UBSan (CPU box; SURVEY.md section 5 "ASan build of host side").
#   tools/asan_check.sh            builds checkerpose_amd/libcheckerpose_hip_asan.so (`make asan`: host-only objects of every .hip)
#                                  and runs the CPU tests that call into the library + the bench launcher's --dry-run against it.
# The sanitizer runtime must be the first DSO of the (uninstrumented) python process, hence LD_PRELOAD; python's own allocations are
# not leak-checked (detect_leaks=0).  Any ASan / UBSan report aborts the process (non-zero exit).
set -e
cd "$(dirname "$0")/.."
make -C checkerpose_amd/csrc asan -j8 >/dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD="$RT" ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
export CHECKERPOSE_AMD_LIB="$PWD/checkerpose_amd/libcheckerpose_hip_asan.so"
python -m pytest tests/test_abi.py tests/test_host_logic.py tests/test_preprocess.py tests/test_pnp.py -x -q -m "not gpu" -p no:cacheprovider
python bench.py --dry-run --steps 3
echo "asan_check: clean"
