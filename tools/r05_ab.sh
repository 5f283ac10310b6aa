#!/bin/bash
# Same-box A/B of library builds / switches on the configurations of tools/bench_configs.py, interleaved repetitions:
#   tools/r05_ab.sh OUT.txt REPS "WORKLOADS" "name|lib.so or -|ENV=v ENV=v" ...        (GPU box, repo root)
# WORKLOADS: words of tools/bench_configs.py (c3 = configs[2], c5 = one GPU's share of configs[4], c4 = configs[3], cc =
# stress_test_collision); "c3np" = configs[2] with every plane kept (FW_NOSPIN=0: rotation / angular velocity read and written).
export FW_ENABLE_KNOBS=1
OUT=$1; REPS=$2; W=$3; shift 3
R=$PWD
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 %-50s %8.2f us/frame  update launches %8.2f us/frame  %5.0f GB/s algorithmic  live %d' % (d['config'][:50], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps'] or 0, d['live']))"; }
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    IFS='|' read -r name lib envs <<< "$spec"
    ( [ "$lib" != "-" ] && export FW_LIB_PATH=$R/$lib
      for kv in $envs; do export "$kv"; done
      for w in $W; do
        if [ "$w" = "c3np" ]; then FW_NOSPIN=0 timeout 600 python tools/bench_configs.py c3 2>/dev/null | line "rep$rep $name [every plane kept]"
        else timeout 600 python tools/bench_configs.py $w 2>/dev/null | line "rep$rep $name"; fi
      done ) | tee -a $OUT
  done
done
