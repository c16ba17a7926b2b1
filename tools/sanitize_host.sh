#!/bin/bash
# Host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (device code is built but not
# instrumented; GPU sanitizers are not available on the pool): the loaders, the BIH build, the culling boxes, the upload
# re-pack and the C-ABI surface, driven by the CPU tests and the loader fuzz.
#   usage: bash tools/sanitize_host.sh [fuzz seconds = 60]
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
OUT=$PWD/squigly-trace_amd/libsanitize_host.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-fast-math -fno-slp-vectorize -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -x hip \
    squigly-trace_amd/csrc/sq_device.hip squigly-trace_amd/csrc/sq_bih_device.hip squigly-trace_amd/csrc/sq_host.cpp -o "$OUT"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$RT SQ_LIB_PATH=$OUT
python -m pytest tests/test_host.py tests/test_cull.py -x -q -m "not gpu" -p no:cacheprovider
python tests/fuzz_loader.py "${1:-60}" 9000000 | tail -1
rm -f "$OUT"
