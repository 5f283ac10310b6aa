#!/bin/bash
# The committed profile artefacts of round 4 (run on the GPU box, from the repo root):  tools/profile_r04.sh
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
TAG=r04
R=$PWD; OUT=gpurun_out/profile_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# 1. the bench line itself (default flags), and the same line with every type on the compacting path
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/bench_20_steps.json 2>> $OUT/bench.err   # the driver's flags
FW_FIFO=0 FW_RANGE=0 timeout 900 python bench.py --no-cpu > $OUT/bench_general_path.json 2>> $OUT/bench.err
# 2. rocprofv3 kernel trace + stats of the same command (the kernel of the headline configuration only)
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT -o ${TAG}_stats --output-format csv -- python $R/bench.py --no-cpu --no-extras > $R/$OUT/stats_bench.json 2>/dev/null; cd $R
f=$(find $OUT -name "${TAG}_stats_kernel_trace.csv" | head -1); [ -n "$f" ] && python profiles/analyze_trace.py $f 600 > $OUT/trace_summary.txt
f=$(find $OUT -name "${TAG}_stats_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_stats_kernel_stats.csv
# 3. PMC passes (own runs, kernel-trace only): the headline kernel, then the traffic of the other configs' update kernels
./tools/pmc.sh $OUT/pmc > /dev/null 2>&1
python profiles/analyze_pmc.py $OUT/pmc > $OUT/pmc_summary.txt
./tools/pmc_configs.sh $OUT/pmc_cfg "c3 c4 c5 cc" > $OUT/pmc_configs.txt 2>&1; rm -rf $OUT/pmc_cfg
# 4. rocprofv3 kernel-trace summaries of configs[2] / configs[4]'s share / configs[3] / stress_test_collision, both paths
timeout 1500 tools/prof_configs.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_configs_$TAG/*_kernel_stats.csv gpurun_out/prof_configs_$TAG/*_trace_summary.txt gpurun_out/prof_configs_$TAG/*_bench.json $OUT/ 2>/dev/null
# 5. every config on one GPU, the small-emitter regime, lifetime ranges in a Nested spawner
timeout 600 python tools/bench_configs.py c1 c3 c4 c5 cc > $OUT/configs.txt 2>&1
timeout 300 python tools/small_emitters.py > $OUT/small_emitters.txt 2>&1
timeout 600 python tools/r04_nested_range.py > $OUT/nested_range_lifetimes.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT/pmc/*/ 2>/dev/null
ls $OUT
