#!/bin/bash
# the parts of tools/profile_round.sh r03 that ran on the wrong path the first time (FW_FIFO=0 alone no longer means
# "compacting path": range rings take over) + the per-config PMC passes + the range soak
export FW_ENABLE_KNOBS=1
R=$PWD; OUT=gpurun_out/profile_r03; mkdir -p $OUT; export TMPDIR=/tmp
FW_FIFO=0 FW_RANGE=0 timeout 900 python bench.py --no-cpu > $OUT/bench_general_path.json 2>> $OUT/bench.err
FW_FIFO=0 FW_RANGE=0 timeout 200 python tools/launch_gaps.py > $OUT/launch_gaps.txt 2>&1
FW_FIFO=0 FW_RANGE=0 timeout 200 python tools/tile_timeline.py > $OUT/tile_timeline.txt 2>&1
FW_FIFO=0 FW_RANGE=0 FW_TL_JITTER=1 timeout 200 python tools/tile_timeline.py > $OUT/tile_timeline_variable_dt.txt 2>&1
FW_FIFO=0 FW_RANGE=0 timeout 300 python tools/fused_sizes.py > $OUT/fused_sizes_general_path.txt 2>&1
FW_FIFO=0 FW_RANGE=0 timeout 300 python tools/var_dt.py 400 > $OUT/var_dt_general_path.txt 2>&1
timeout 300 python tools/dbg_modes.py > $OUT/dbg_modes.txt 2>&1
rm -rf $OUT/pmc_cfg; ./tools/pmc_configs.sh $OUT/pmc_cfg "c3 c4 c5" > $OUT/pmc_configs.txt 2>&1; rm -rf $OUT/pmc_cfg
timeout 1200 python tools/soak_range.py > $OUT/soak_range.txt 2>&1
ls $OUT | wc -l
