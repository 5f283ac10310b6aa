mkdir -p gpurun_out/r04e
(timeout 1500 python -m pytest tests/test_gpu_range.py tests/test_gpu_configs.py tests/test_gpu_limits.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r04e/pytest.txt
cat gpurun_out/r04e/pytest.txt
export FW_ENABLE_KNOBS=1
AB=$PWD/bevy_firework_amd/csrc/libfirework_hip_ab.so
for rep in 1 2 3; do
  for v in "mask variants/mask.so 0" "xcd0 $AB 0" "xcd1 $AB 1"; do
    set -- $v
    so=$2; [ "${so#/}" = "$so" ] && so=$PWD/$so
    FW_LIB_PATH=$so FW_RANGE_XCD=$3 timeout 600 python tools/bench_configs.py c5 c3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rep$rep $1 %s: %.1f us/step  update kernels %.1f us/frame  %.0f GB/s algorithmic' % (d['config'][:24], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps']))" | tee -a gpurun_out/r04e/ab.txt
  done
done
