#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# A/B of the range-ring path on configs[2] / configs[4]'s share (run on the GPU box): tools/range_ab.sh
VARIANTS=("${@:-}")
for v in "${VARIANTS[@]}"; do
  echo "== ${v:-default}"
  env $v FW_HOST_PROF=1 python tools/bench_configs.py c3 c5 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('  %-55s %7.1f us/step  kernels %7.1f us  %5.0f GB/s algorithmic' % (d['config'][:55], d['us_per_step'], d['update_kernels_us_per_frame'], d['algorithmic_GBps']))
    elif 'table uploads' in ln: print('  ' + ln.strip())
"
done
