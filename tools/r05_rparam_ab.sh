#!/bin/bash
# The per-frame records of the range launch in pinned host memory (product) against fine-grained device memory the host writes
# through the large BAR (fw_ctx::param_bar: the product's choice where the platform maps it; FW_PARAM_BAR=0: pinned), same box,
# interleaved.   tools/r05_rparam_ab.sh OUT.txt
export FW_ENABLE_KNOBS=1
OUT=$1; : > $OUT; R=$PWD
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 %-46s %8.2f us/frame  update launches %8.2f us/frame  live %d' % (d['config'][:46], d['us_per_step'], d['update_kernels_us_per_frame'], d['live']))"; }
for rep in 1 2 3; do
  for v in "pinned FW_PARAM_BAR=0" "bar-written FW_PARAM_BAR=1"; do
    set -- $v; name=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      echo "rep$rep $name few small emitters (us per frame pipelined / synchronised): $(timeout 300 python tools/r05_few_quick.py 2>/dev/null)"
      timeout 600 python tools/bench_configs.py c3 c5 2>/dev/null | line "rep$rep $name" ) | tee -a $OUT
  done
done
