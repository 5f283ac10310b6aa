#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# PMC passes for the bench (run on the GPU box): tools/pmc.sh <outdir> [lib.so]
# Counters are collected in their own runs (kernel-trace only), one rocprofv3 pass per counter group.
OUT=$1; LIB=${2:-}
R=$PWD; export TMPDIR=/tmp; mkdir -p $OUT; cd /tmp
[ -n "$LIB" ] && export FW_LIB_PATH=$R/$LIB
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT -o $name --output-format csv -- python $R/bench.py --steps 100 --warmup 20 --no-cpu --no-events > $R/$OUT/$name.log 2>&1; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES
run waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R; ls $OUT
