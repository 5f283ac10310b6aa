mkdir -p gpurun_out/r04c
(timeout 1500 python -m pytest tests/test_gpu_range.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_limits.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r04c/pytest.txt
cat gpurun_out/r04c/pytest.txt
export FW_ENABLE_KNOBS=1
for rep in 1 2 3; do
  for v in "mask variants/mask.so 4" "seq0 variants/seq2.so 0" "seq4 variants/seq2.so 4" "seq8 variants/seq2.so 8"; do
    set -- $v
    FW_LIB_PATH=$PWD/$2 FW_RANGE_SEQ_TILES=$3 timeout 600 python tools/bench_configs.py c5 c3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rep$rep $1 %s: %.1f us/step  update kernels %.1f us/frame  %.0f GB/s algorithmic' % (d['config'][:24], d['us_per_step'], d['update_kernels_us_per_frame'], d['update_kernels_algorithmic_GBps']))" | tee -a gpurun_out/r04c/ab.txt
  done
done
python tools/bench_configs.py cc 2>&1 | tee gpurun_out/r04c/collision.txt | cut -c1-400
FW_FIFO=0 FW_RANGE=0 python tools/bench_configs.py cc 2>&1 | tee gpurun_out/r04c/collision_general.txt | cut -c1-400
