mkdir -p gpurun_out/r04f
export FW_ENABLE_KNOBS=1
for rep in 1 2 3; do
  for v in "yr4 variants/yr4.so" "yr8 variants/yr8.so" "yr2 variants/yr2.so"; do
    set -- $v
    FW_LIB_PATH=$PWD/$2 timeout 600 python tools/bench_configs.py c5 c3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rep$rep $1 %s: %.1f us/step  update kernels %.1f us/frame  %.0f GB/s algorithmic' % (d['config'][:24], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps']))" | tee -a gpurun_out/r04f/ab.txt
  done
done
