#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the C-ABI library and the CPU tests that call into it (no GPU: registration and hipRTC compile checks of run-time
# Hamiltonians, hj_plan_substep's device-free launch planning, bench.py --plan-only).  The sanitizers instrument the HOST code of hj_api.hip / hj_rtc.hip only
# (-Xarch_host; GPU ASan is not available on this pool); the kernel objects are the product's.  Scratch under gpurun_out/asan/.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/asan; mkdir -p $out; cd $root/levelsetpy_amd/csrc
make -s -j8 >/dev/null
F="-ffp-contract=on -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer"
/opt/rocm/bin/hipcc $F -c -o $out/hj_api.o hj_api.hip
/opt/rocm/bin/hipcc $F -c -o $out/hj_rtc.o hj_rtc.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fsanitize=address,undefined -shared-libsan -o $out/libhj_asan.so $out/hj_api.o $out/hj_rtc.o inst_*.o term_*.o -ldl
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $root
HJ_LIB=$out/libhj_asan.so LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_cabi.py tests/test_host_logic.py tests/test_bench_launcher.py -q -s -m "not gpu" > $out/cpu_tests.log 2>&1 || true
tail -1 $out/cpu_tests.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $out/cpu_tests.log)"
# UBSan alone also runs on the GPU box (no interceptors): build with  -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=vptr , link with
# -fsanitize=undefined -fno-sanitize=vptr into an in-tree directory (gpurun_out/ does not travel), and run
#   HJ_LIB=<that library> LD_PRELOAD=<libclang_rt.ubsan_standalone-x86_64.so> python -m pytest tests -m gpu --deselect tests/test_gpu_fuzz.py
# (profiles/r05_host_sanitizers.txt: 620 passed, 0 reports)
