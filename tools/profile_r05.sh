#!/bin/bash
# The committed profile artefacts of round 5 (run on the GPU box, from the repo root):  tools/profile_r05.sh
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
TAG=r05
R=$PWD; OUT=gpurun_out/profile_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# 1. the bench line itself (default flags), the driver's flags, and the same line with every type on the compacting path
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/bench_20_steps.json 2>> $OUT/bench.err
# 2. rocprofv3 kernel trace + stats of the same command (the kernel of the headline configuration only)
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT -o ${TAG}_stats --output-format csv -- python $R/bench.py --no-cpu --no-extras > $R/$OUT/stats_bench.json 2>/dev/null; cd $R
f=$(find $OUT -name "${TAG}_stats_kernel_trace.csv" | head -1); [ -n "$f" ] && python profiles/analyze_trace.py $f 600 > $OUT/trace_summary.txt
f=$(find $OUT -name "${TAG}_stats_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_stats_kernel_stats.csv
# 3. PMC passes (own runs, kernel-trace only): the headline kernel, then the traffic of the other configs' update kernels
./tools/pmc.sh $OUT/pmc > /dev/null 2>&1
python profiles/analyze_pmc.py $OUT/pmc > $OUT/pmc_summary.txt
./tools/pmc_configs.sh $OUT/pmc_cfg "c3 c4 c5" > $OUT/pmc_configs.txt 2>&1; rm -rf $OUT/pmc_cfg
# 4. rocprofv3 kernel-trace summaries of configs[2] / configs[4]'s share / configs[3] (ONE launch per frame now) / stress_test_collision
timeout 1500 tools/prof_configs.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_configs_$TAG/*_kernel_stats.csv gpurun_out/prof_configs_$TAG/*_trace_summary.txt gpurun_out/prof_configs_$TAG/*_bench.json $OUT/ 2>/dev/null
# ... and of thousands of small emitters: the wave-per-type kernel against a workgroup per type
cd /tmp
for v in "small_emitters 1" "small_emitters_workgroup_per_type 0"; do set -- $v
  rm -rf $R/$OUT/tmp_$1; FW_SMALL=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/tmp_$1 -o $1 --output-format csv -- python $R/tools/small_emitters.py > $R/$OUT/$1.log 2>&1
  f=$(find $R/$OUT/tmp_$1 -name "$1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/$1_kernel_stats.csv
  t=$(find $R/$OUT/tmp_$1 -name "$1_kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/profiles/analyze_trace.py $t 300 > $R/$OUT/$1_trace_summary.txt 2>&1
  rm -rf $R/$OUT/tmp_$1
done
# ... and of hundreds of mid-size emitters (1024 x 1000 particles): a workgroup per type (the wide role of fw_k_update_small) against the compacting kernels
for v in "mid_emitters_wide 2048" "mid_emitters_compacting 0"; do set -- $v
  rm -rf $R/$OUT/tmp_$1; FW_WIDE_MAX=$2 FW_CASES=1024x1000 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/tmp_$1 -o $1 --output-format csv -- python $R/tools/r05_mid_kernel.py > $R/$OUT/$1.log 2>&1
  f=$(find $R/$OUT/tmp_$1 -name "$1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/$1_kernel_stats.csv
  t=$(find $R/$OUT/tmp_$1 -name "$1_kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/profiles/analyze_trace.py $t 300 > $R/$OUT/$1_trace_summary.txt 2>&1
  rm -rf $R/$OUT/tmp_$1
done
cd $R
# 5. every config on one GPU, the examples at their own sizes, the few-emitters sweep, the spill sweep, the soak
timeout 600 python tools/bench_configs.py c1 c3 c4 c5 cc > $OUT/configs.txt 2>&1
timeout 600 python tools/r04_examples_latency.py > $OUT/examples_latency.txt 2>&1
timeout 900 python tools/r04_few_small_emitters.py > $OUT/few_small_emitters.txt 2>&1
timeout 300 python tools/r05_mid_emitters.py > $OUT/mid_emitters_default.txt 2>&1
FW_CASES=192x600,256x600,384x600,512x600,768x600,256x1000,384x1000,512x1000,768x1000,1024x1000,256x1500,512x1500,768x1500,1024x1500 timeout 300 python tools/r05_mid_kernel.py > $OUT/mid_kernel_default.txt 2>&1
timeout 1500 python tools/soak_r05.py > $OUT/soak_r05.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT/pmc/*/ 2>/dev/null
ls $OUT
