#!/bin/bash
# N processes of tools/r06_burst_repro.py at once; the trace tail of the failing ones is kept
N=${1:-6}; REPS=${2:-15}
mkdir -p gpurun_out/burst
for p in $(seq 1 $N); do
  ( python tools/r06_burst_repro.py $REPS > gpurun_out/burst/p$p.out 2> gpurun_out/burst/p$p.err; tail -150 gpurun_out/burst/p$p.err > gpurun_out/burst/p$p.tail; rm gpurun_out/burst/p$p.err ) &
done
wait
cat gpurun_out/burst/*.out
