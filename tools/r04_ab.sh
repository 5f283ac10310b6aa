#!/bin/bash
# A/B of library builds on configs[1] (bench.py), configs[2] and one GPU's share of configs[4] (tools/bench_configs.py):
#   tools/r04_ab.sh OUT.txt name=path.so name=path.so ...      (run on the GPU box, from the repo root; 3 interleaved repetitions)
export FW_ENABLE_KNOBS=1
OUT=$1; shift
R=$PWD
: > $OUT
for rep in 1 2 3; do
  for nv in "$@"; do
    name=${nv%%=*}; so=${nv#*=}
    export FW_LIB_PATH=$R/$so
    c1=$(timeout 300 python bench.py --steps 400 --warmup 60 --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us/step  kernel %.2f us  frac %.3f' % (d['ms_per_step']*1000, d['roofline']['avg_kernel_us'], d['roofline']['frac']))")
    echo "rep$rep $name configs[1]: $c1" | tee -a $OUT
    timeout 600 python tools/bench_configs.py c5 c3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rep$rep $name %s: %.1f us/step  update kernels %.1f us/frame  %.0f GB/s algorithmic' % (d['config'][:24], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps']))" | tee -a $OUT
  done
done
