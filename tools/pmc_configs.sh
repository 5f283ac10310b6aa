#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# HBM-side traffic (PMC) of the update kernels of the other configs: tools/pmc_configs.sh <outdir> "c3 c4"   (GPU box)
# FETCH_SIZE / WRITE_SIZE in their own rocprofv3 passes (kernel-trace only), summarised per kernel by analyze_pmc.py.
OUT=$1; W=${2:-c3}
R=$PWD; export TMPDIR=/tmp; mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT -o $c --output-format csv -- python $R/tools/bench_configs.py $W > $R/$OUT/$c.log 2>&1
done
cd $R; python profiles/analyze_pmc.py $OUT 60
