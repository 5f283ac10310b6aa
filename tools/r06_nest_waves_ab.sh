#!/bin/bash
# Round 6: fw_k_update_fifo_nest at 4 waves per SIMD (128 VGPRs, 48 B/lane scratch, 3 spilled VGPRs) against the
# form without scratch (amdgpu_waves_per_eu(3): 143 VGPRs, 0 scratch; tools/build_variant.sh nest3 -DFW_NEST_WAVES=3), configs[3],
# same box, interleaved.   tools/r06_nest_waves_ab.sh   (GPU box)
R=$PWD
for rep in 1 2 3; do
  for v in "4-waves-product -" "3-waves-no-scratch variants/nest3/libfirework_hip.so"; do set -- $v
    ( [ "$2" != "-" ] && export FW_LIB_PATH=$R/$2
      python tools/bench_configs.py c4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1 rep$rep', round(d['us_per_step'],2), 'us per frame, update kernels', round(d['update_kernels_us_per_frame'],2), 'us,', d['live'], 'live')" )
  done
done
