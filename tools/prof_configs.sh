#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# rocprofv3 --kernel-trace --stats of the HBM-resident configurations (run on the GPU box):
#   tools/prof_configs.sh r03       -> gpurun_out/prof_configs_r03/{configs2,configs4_share}_{kernel_stats.csv,trace_summary.txt,bench.json}
# Every hbm_resident figure of the bench line is then reproducible from a CSV under profiles/.
TAG=${1:-r04}
R=$PWD; OUT=$R/gpurun_out/prof_configs_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
export FW_ENABLE_KNOBS=1
for cfg in "c3 configs2" "c5 configs4_share" "c4 configs3_nested" "cc stress_test_collision" "c3 configs2_compacting FW_RANGE=0" "c5 configs4_share_compacting FW_RANGE=0"; do
  set -- $cfg
  [ -n "$3" ] && export $3 || unset FW_RANGE
  rm -rf $OUT/tmp_$2
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tmp_$2 -o $2 --output-format csv -- python $R/tools/bench_configs.py $1 > $OUT/$2_bench.json 2> $OUT/$2.err
  f=$(find $OUT/tmp_$2 -name "$2_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$2_kernel_stats.csv
  t=$(find $OUT/tmp_$2 -name "$2_kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/profiles/analyze_trace.py $t 300 > $OUT/$2_trace_summary.txt 2>&1
  rm -rf $OUT/tmp_$2
done
cd $R; ls $OUT
