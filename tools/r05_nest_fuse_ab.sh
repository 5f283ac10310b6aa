#!/bin/bash
# configs[3] (sparks -> smoke, 3.9M particles) with the Nested entry inside the FIFO launch (round 5) against the separate
# fw_k_nest pass, same box, three interleaved repetitions:   tools/r05_nest_fuse_ab.sh OUT.txt   (GPU box, repo root)
export FW_ENABLE_KNOBS=1
OUT=$1; : > $OUT
for rep in 1 2 3; do
  for fuse in 1 0; do
    FW_NEST_FUSE=$fuse timeout 600 python tools/bench_configs.py c4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rep$rep FW_NEST_FUSE=$fuse %s: %.2f us/frame  update launches %.2f us/frame  live %d  %.0f GB/s algorithmic (frame)' % (d['config'][:30], d['us_per_step'], d['update_kernels_us_per_frame'], d['live'], d['algorithmic_GBps']))" | tee -a $OUT
  done
done
