#!/bin/bash
# Thousands of small emitters (the regime the reference advertises, README.md:11-16): this build (a wave per small type,
# fw_k_small.hip) against the same build with FW_SMALL=0 (a workgroup of the compacting kernels per type) and the r04 library,
# same box, interleaved; then the host-phase table (ab build, FW_HOST_PROF) and 4096 x 200 on one and two contexts.
#   tools/r05_small_emitters.sh OUT.txt        (GPU box, repo root)
export FW_ENABLE_KNOBS=1
OUT=$1; : > $OUT; R=$PWD
for rep in 1 2; do
  echo "== rep $rep: this build (wave per small type)" | tee -a $OUT
  timeout 300 python tools/small_emitters.py 2>/dev/null | tee -a $OUT
  echo "== rep $rep: this build, FW_SMALL=0 (a workgroup per type)" | tee -a $OUT
  FW_SMALL=0 timeout 300 python tools/small_emitters.py 2>/dev/null | tee -a $OUT
  echo "== rep $rep: r04 library" | tee -a $OUT
  FW_LIB_PATH=$R/variants/r04/libfirework_hip.so timeout 300 python tools/small_emitters.py 2>/dev/null | tee -a $OUT
done
for n in 2048 4096 8192; do
  echo "== host phases of fw_step, $n x 200 (ab build, FW_HOST_PROF): wave per small type / FW_SMALL=0" | tee -a $OUT
  timeout 300 python tools/r04_host_prof.py $n 200 2>&1 | grep -E "us/step|host half|table uploads" | tee -a $OUT
  FW_SMALL=0 timeout 300 python tools/r04_host_prof.py $n 200 2>&1 | grep -E "us/step|host half" | tee -a $OUT
done
if [ -x examples/many_contexts ]; then
  for c in 1 2; do timeout 120 examples/many_contexts 4096 200 $c 2>&1 | tail -1 | tee -a $OUT; done
  for c in 1 2; do timeout 120 examples/many_contexts 2048 200 $c 2>&1 | tail -1 | tee -a $OUT; done
fi
