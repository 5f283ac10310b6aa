#!/bin/bash
# The host half of a frame with thousands of small emitters: FW_HOST_FAST (solo segments: one pass over a segment record per frame;
# ops written in place into the parameter slot) against FW_HOST_FAST=0 (the round-4 passes), same box, interleaved; then the
# host-phase tables (ab build, FW_HOST_PROF).
#   tools/r05_host_fast.sh OUT.txt        (GPU box, repo root)
export FW_ENABLE_KNOBS=1
OUT=$1; : > $OUT
for rep in 1 2 3; do
  for n in 2048 4096; do
    for c in 1 2; do
      echo -n "rep $rep this build:      " | tee -a $OUT; timeout 120 examples/many_contexts $n 200 $c 2>&1 | tail -1 | tee -a $OUT
      echo -n "rep $rep FW_HOST_FAST=0:  " | tee -a $OUT; FW_HOST_FAST=0 timeout 120 examples/many_contexts $n 200 $c 2>&1 | tail -1 | tee -a $OUT
    done
  done
done
for n in 2048 4096 8192; do
  echo "== host phases of fw_step, $n x 200 (ab build, FW_HOST_PROF): this build / FW_HOST_FAST=0" | tee -a $OUT
  timeout 300 python tools/r04_host_prof.py $n 200 2>&1 | grep -E "us/step|host half|table uploads" | tee -a $OUT
  FW_HOST_FAST=0 timeout 300 python tools/r04_host_prof.py $n 200 2>&1 | grep -E "us/step|host half" | tee -a $OUT
done
echo "== tools/small_emitters.py: this build / FW_HOST_FAST=0" | tee -a $OUT
timeout 300 python tools/small_emitters.py 2>/dev/null | tee -a $OUT
FW_HOST_FAST=0 timeout 300 python tools/small_emitters.py 2>/dev/null | tee -a $OUT
