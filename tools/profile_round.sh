#!/bin/bash
# Produce the committed profile artefacts of a round (run on the GPU box):  tools/profile_round.sh r01
TAG=${1:-r01}
R=$PWD; OUT=gpurun_out/profile_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# 1. the bench line itself (default flags)
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. rocprofv3 kernel trace + stats of the same command
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT -o ${TAG}_stats --output-format csv -- python $R/bench.py --no-cpu > $R/$OUT/stats_bench.json 2>/dev/null; cd $R
python profiles/analyze_trace.py $OUT/${TAG}_stats_kernel_trace.csv 600 > $OUT/trace_summary.txt
# 3. PMC passes (own runs, kernel-trace only)
./tools/pmc.sh $OUT/pmc > /dev/null 2>&1
python profiles/analyze_pmc.py $OUT/pmc > $OUT/pmc_summary.txt
# 4. memory microbenchmark (measured roofline of the kernel's load/store shape)
./tools/membw 64 > $OUT/membw.txt 2>&1
# 5. in-kernel timelines, launch period, the rows ranked next, size sweep, feature twins
timeout 200 python tools/launch_gaps.py > $OUT/launch_gaps.txt 2>&1
timeout 200 python tools/tile_timeline.py > $OUT/tile_timeline.txt 2>&1
timeout 300 python tools/bench_next_rows.py > $OUT/next_rows.txt 2>&1
timeout 300 python tools/fused_sizes.py > $OUT/fused_sizes.txt 2>&1
[ -x tools/launchgap ] && timeout 200 ./tools/launchgap > $OUT/launchgap.txt 2>&1
rm -f $OUT/${TAG}_stats_kernel_trace.csv $OUT/pmc/*kernel_trace.csv   # large; the summaries are what is kept
ls $OUT
