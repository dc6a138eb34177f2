#!/bin/bash
# Host-side code of the library (argument checks, sweep planner, tile preparation) under AddressSanitizer + UBSan.
# CPU only: the device code is compiled without sanitizers (not available for gfx950 on this pool), the tests run are
# the ones that need no GPU.  Usage: bash tools/asan_host.sh
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/lgconv_asan
mkdir -p "$out"
hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Iinclude -fsanitize=address,undefined \
      -fno-omit-frame-pointer -Wno-option-ignored -shared -o "$out/liblgconv_hip.so" gnn-ecommerce_amd/csrc/lgconv_hip.hip
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LGCN_LIB_PATH="$out/liblgconv_hip.so" \
    python -m pytest tests/test_abi_and_host.py tests/test_ingest_serving.py -x -q -m "not gpu"
