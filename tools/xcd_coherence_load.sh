#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# tools/xcd_coherence alone, then as N concurrent processes on the one GPU (run on the GPU box)
N=${1:-8}; T=${2:-20}; BIG=$4
for mode in ${3:-0 1 2 3 4 5}; do
echo "mode $mode alone:"; ./tools/xcd_coherence $T 20000 $mode $BIG
echo "mode $mode, $N processes:"; for i in $(seq $N); do ./tools/xcd_coherence $T 20000 $mode $BIG & done; wait
done
