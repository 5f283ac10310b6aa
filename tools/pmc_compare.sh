#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# Counter comparison of the real update kernel against its structural twin (run on the GPU box):
#   tools/pmc_compare.sh <outdir>
# One rocprofv3 pass per counter group (kernel-trace only), for `./tools/launchgap twin` and for the C++ stress test.
OUT=$1; R=$PWD; export TMPDIR=/tmp; mkdir -p $OUT; cd /tmp
export LD_LIBRARY_PATH=$R/bevy_firework_amd/csrc
grp() {
  name=$1; shift
  timeout -k 5 60 rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT -o twin_$name --output-format csv -- $R/tools/launchgap twin > $R/$OUT/twin_$name.log 2>&1
  timeout -k 5 60 rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT -o real_$name --output-format csv -- $R/examples/stress_test 1000000 200 > $R/$OUT/real_$name.log 2>&1
}
# (at most four counters of one block per pass: a request the hardware cannot schedule makes rocprofv3 abort and hang)
grp icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
grp utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
grp tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
grp tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum
grp tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
grp tcc2 TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
grp tcc3 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum
grp sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
grp ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
grp td TD_TC_STALL_sum TD_TD_BUSY_sum
cd $R; ls $OUT | head -50
