#!/bin/bash
# Host-side AddressSanitizer / UBSan run of the C-ABI shim in the CPU container: builds
# viabel_amd/libviabel_hip_asan.so (`make asan`: host code instrumented, device code untouched) and runs the
# CPU-side boundary tests against it.  Never run on the GPU pool (GPU sanitizers are refused there).
set -e
cd "$(dirname "$0")/.."
make -C viabel_amd/csrc asan -j8 > /dev/null
rt=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export VIABEL_AMD_LIB=$PWD/viabel_amd/libviabel_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_PRELOAD=$rt
python tests/cabi_null_probe.py
python -m pytest tests/test_cabi_symbols.py tests/test_cabi_null_ctx.py tests/test_legacy_rng_cpu.py -q -p no:cacheprovider
