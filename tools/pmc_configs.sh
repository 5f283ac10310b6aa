#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# HBM-side traffic (PMC) of the update kernels of the other configs: tools/pmc_configs.sh <outdir> "c3 c4 c5"   (GPU box)
# FETCH_SIZE / WRITE_SIZE in their own rocprofv3 passes (kernel-trace only), ONE configuration per pass (several of them
# run the same kernel), summarised per kernel over the last 60 dispatches by analyze_pmc.py.
OUT=$1; W=${2:-c3}
R=$PWD; export TMPDIR=/tmp; mkdir -p $OUT
for w in $W; do
  mkdir -p $OUT/$w; cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/$w -o $c --output-format csv -- python $R/tools/bench_configs.py $w > $R/$OUT/$w/$c.log 2>&1
  done
  cd $R; echo "== $w: $(grep -o '"config": "[^"]*"' $OUT/$w/FETCH_SIZE.log | head -1)  (KiB per launch; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md)"
  python profiles/analyze_pmc.py $OUT/$w 60
done
