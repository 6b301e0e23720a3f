#!/bin/bash
# The CPU suite (pytest -m "not gpu") with the host library and the oracle built with AddressSanitizer + UBSan
# (device code is not instrumented: GPU sanitizers are unavailable on this pool).
#   tools/run_sanitized.sh [pytest args]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C $root/merian-quake_amd asan -j4 > /dev/null
make -C $root/oracle asan > /dev/null
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$rt
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export MQHIP_LIB=$root/merian-quake_amd/lib/libmqhip_asan.so MQORACLE_LIB=$root/oracle/libmqoracle_asan.so MQHIP_NO_TORCH_PRELOAD=1
cd $root && exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
