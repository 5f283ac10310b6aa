#!/bin/bash
# round 6: tests/test_gpu_golden.py::test_global_burst_into_a_nested_fed_type_stays_in_bounds[fifo] failed ONCE with "check 2 of an
# update kernel failed" in a 4-process xdist run (the GPU shared by four test processes) and never alone: N processes looping over it
N=${1:-6}; REPS=${2:-8}; K=${3:-test_global_burst_into_a_nested_fed_type_stays_in_bounds}
mkdir -p gpurun_out/contention
for p in $(seq 1 $N); do
  ( for i in $(seq 1 $REPS); do python -m pytest tests/test_gpu_golden.py -m gpu -q -x -k "$K" 2>&1 | tail -25 > gpurun_out/contention/p${p}_$i.log; done ) &
done
wait
grep -l 'failed\|Error' gpurun_out/contention/*.log | head; grep -h 'passed\|failed' gpurun_out/contention/*.log | sort | uniq -c
