#!/bin/bash
# Host code of the library (API glue, parsers, compiler, substring extraction, host walk) under AddressSanitizer + UBSan: the
# CPU test suite against an instrumented build (no GPU needed; GPU sanitizers are not available on this pool).
# test_deep_nesting... is left out: ASan's inflated frames overflow the stack at the 1500 nested groups the test allows.
set -e
cd "$(dirname "$0")/../halo2_regex_amd/csrc"; O=/tmp/hrx_asan; mkdir -p $O
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
for f in hrx_api.cpp hrx_defs.cpp hrx_host_walk.cpp hrx_compile.cpp hrx_substr.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -c -o $O/${f%.cpp}.o $f
done
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -shared-libsan -o $O/libhrx_asan.so $O/*.o hrx_kernel.o hrx_kernel_pm.o hrx_kernel_pp.o hrx_kernel_pmd.o hrx_kernel_sm.o hrx_kernel_mp.o hrx_kernel_spec.o hrx_kernel_tp.o hrx_place.o -Wl,-rpath,/opt/rocm/lib
cd ../..
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$RT HRX_LIB_PATH=$O/libhrx_asan.so \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_compiler.py::test_deep_nesting_is_a_parse_error_not_a_crash 2>&1 | tee $O/log.txt | tail -3
echo "sanitizer reports: $(grep -a -c 'runtime error\|AddressSanitizer' $O/log.txt)"
