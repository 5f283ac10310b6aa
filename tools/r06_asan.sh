#!/bin/bash
# Round 6: the HOST half of the engine under AddressSanitizer + UndefinedBehaviorSanitizer on the GPU box.
#   make -C bevy_firework_amd/csrc asan        (here: the five fw_engine_*.cpp units as plain C++ with the sanitizers)
#   tools/r06_asan.sh                           (GPU box) -> gpurun_out/asan/{lifecycle,fuzz,parity,soak}.log + asan.* reports
# The sanitizer runtime is preloaded into the (uninstrumented) python; leak detection off (the HIP runtime's own allocations);
# protect_shadow_gap=0: the ROCm runtime reserves address ranges of its own.
R=$PWD; OUT=$R/gpurun_out/asan; mkdir -p $OUT; rm -f $OUT/asan.* $OUT/ubsan.*
RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:log_path=$OUT/asan:detect_stack_use_after_return=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$OUT/ubsan
export FW_LIB_PATH=$R/bevy_firework_amd/csrc/libfirework_hip_asan.so
timeout 1500 python -m pytest tests/test_gpu_lifecycle.py -m gpu -q > $OUT/lifecycle.log 2>&1; tail -2 $OUT/lifecycle.log
FW_FUZZ_EXTRA=${FW_FUZZ_EXTRA:-100} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_limits.py tests/test_gpu_host_fast.py tests/test_gpu_range.py -m gpu -q > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
FW_SOAK_FRAMES=${FW_SOAK_FRAMES:-3000} timeout 1500 python tools/soak_r05.py > $OUT/soak.log 2>&1; tail -3 $OUT/soak.log
echo "sanitizer reports:"; ls $OUT | grep -E '^(asan|ubsan)\.' | wc -l; for f in $OUT/asan.* $OUT/ubsan.*; do [ -f "$f" ] && head -40 "$f"; done 2>/dev/null | head -150
