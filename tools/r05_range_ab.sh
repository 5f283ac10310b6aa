#!/bin/bash
# Range-ring launches, round 5 against the round-4 library on the same box, three interleaved repetitions:
#   configs[2] (256 x 64Ki), one GPU's share of configs[4] (512 x 8192), configs[2] with every plane kept (FW_NOSPIN=0: the
#   particles "spin", rotation / angular-velocity planes are read and written), stress_test_collision (the colliding FIFO kernel)
#   variants: this build (young tiles chosen per launch) / this build with 1024-slot young tiles always (FW_RANGE_YOUNG_BIG=0) /
#   the r04 library                                              tools/r05_range_ab.sh OUT.txt      (GPU box, repo root)
export FW_ENABLE_KNOBS=1
OUT=$1; : > $OUT
R=$PWD
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 %-46s %8.2f us/frame  update launches %8.2f us/frame  %5.0f GB/s algorithmic  live %d' % (d['config'][:46], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps'] or 0, d['live']))"; }
for rep in 1 2 3; do
  for v in "r05" "r05-young1024 FW_RANGE_YOUNG_BIG=0" "r04 FW_LIB_PATH=$R/variants/r04/libfirework_hip.so"; do
    set -- $v; name=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      timeout 600 python tools/bench_configs.py c3 c5 2>/dev/null | line "rep$rep $name"
      FW_NOSPIN=0 timeout 600 python tools/bench_configs.py c3 2>/dev/null | line "rep$rep $name every-plane-kept"
      [ "$name" != "r05-young1024" ] && timeout 600 python tools/bench_configs.py cc 2>/dev/null | line "rep$rep $name" ) | tee -a $OUT
  done
done
