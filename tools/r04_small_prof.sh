#!/bin/bash
# rocprofv3 --kernel-trace --stats of the many-small-emitters regime (tools/r04_host_prof.py's loop, product build):
#   tools/r04_small_prof.sh [emitters] [live per emitter]  -> gpurun_out/small_prof/
N=${1:-2048}; PER=${2:-200}
R=$PWD; OUT=$R/gpurun_out/small_prof; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/tmp
FW_LIB_PATH=$R/bevy_firework_amd/csrc/libfirework_hip.so timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $OUT/tmp -o small --output-format csv -- python $R/tools/r04_host_prof.py $N $PER > $OUT/small_${N}x${PER}.out 2>&1
f=$(find $OUT/tmp -name "small_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/small_${N}x${PER}_kernel_stats.csv
t=$(find $OUT/tmp -name "small_kernel_trace.csv" | head -1)
[ -n "$t" ] && python $R/profiles/analyze_trace.py $t 300 > $OUT/small_${N}x${PER}_trace_summary.txt 2>&1
rm -rf $OUT/tmp
cd $R; tail -3 $OUT/small_${N}x${PER}.out; head -12 $OUT/small_${N}x${PER}_kernel_stats.csv | cut -c1-200
