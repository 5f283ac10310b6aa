#!/bin/bash
# The sanitizer pass again on the round's FINAL build (component planes: realloc_segment transposes, the readers take a layout flag):
# lifecycle histories (every ring <-> compacting transition) + the range / limits tests under gcc ASan + UBSan, then -- without the
# sanitizers -- fuzz cases and lifecycle histories the committed suite does not run (other seeds).   tools/r06_asan_final.sh   (GPU box)
R=$PWD; OUT=$R/gpurun_out/asan_final; mkdir -p $OUT; rm -f $OUT/asan.* $OUT/ubsan.*
(
RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export LD_PRELOAD="$RT"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:abort_on_error=0:log_path=$OUT/asan:detect_stack_use_after_return=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$OUT/ubsan
export FW_LIB_PATH=$R/bevy_firework_amd/csrc/libfirework_hip_asan.so
echo "---- lifecycle under ASan + UBSan"; timeout 900 python -m pytest tests/test_gpu_lifecycle.py -m gpu -q < /dev/null 2>&1 | tail -2
echo "---- range + limits under ASan + UBSan"; timeout 900 python -m pytest tests/test_gpu_range.py tests/test_gpu_limits.py -m gpu -q -k "not attach and not instance" < /dev/null 2>&1 | tail -2
)
echo "sanitizer reports: $(ls $OUT | grep -E '^(asan|ubsan)\.' | wc -l)"; for f in $OUT/asan.* $OUT/ubsan.*; do [ -f "$f" ] && head -30 "$f"; done 2>/dev/null | head -90
echo "---- 400 further fuzz cases (seeds from 20000)"; FW_FUZZ_EXTRA=400 FW_FUZZ_OFFSET=20000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "random_spawner_matches" < /dev/null 2>&1 | tail -2
echo "---- 200 further lifecycle histories (seeds from 7000)"; FW_LIFECYCLE_CASES=200 FW_LIFECYCLE_OFFSET=7000 timeout 900 python -m pytest tests/test_gpu_lifecycle.py -m gpu -q < /dev/null 2>&1 | tail -2
