#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# Produce the committed profile artefacts of a round (run on the GPU box):  tools/profile_round.sh r02
# (rounds 2 and 3; round 4 uses tools/profile_r04.sh.  Steps that set FW_DEBUG / FW_HOST_PROF / FW_DERIVED / FW_FIFO_NESTED need the
# `ab` build since round 4: export FW_LIB_PATH=$PWD/bevy_firework_amd/csrc/libfirework_hip_ab.so first.)
TAG=${1:-r02}
R=$PWD; OUT=gpurun_out/profile_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# 1. the bench line itself (default flags)
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
FW_FIFO=0 FW_RANGE=0 timeout 900 python bench.py --no-cpu > $OUT/bench_general_path.json 2>> $OUT/bench.err   # the same line with every type on the compacting path
# 2. rocprofv3 kernel trace + stats of the same command (without the CPU baseline and the extra workloads: the
#    kernel of the headline configuration only)
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT -o ${TAG}_stats --output-format csv -- python $R/bench.py --no-cpu --no-extras > $R/$OUT/stats_bench.json 2>/dev/null; cd $R
python profiles/analyze_trace.py $OUT/${TAG}_stats_kernel_trace.csv 600 > $OUT/trace_summary.txt
# 3. PMC passes (own runs, kernel-trace only)
./tools/pmc.sh $OUT/pmc > /dev/null 2>&1
python profiles/analyze_pmc.py $OUT/pmc > $OUT/pmc_summary.txt
./tools/pmc_configs.sh $OUT/pmc_cfg "c3 c4 c5" > $OUT/pmc_configs.txt 2>&1; rm -rf $OUT/pmc_cfg   # traffic of the update kernels of configs[2] / [3] / [4]'s share
# 4. memory microbenchmarks (measured roofline of the kernel's load/store shape; copy sweep at 1 GiB)
./tools/membw 64 > $OUT/membw.txt 2>&1
./tools/membw 1024 copy > $OUT/copy_sweep.txt 2>&1
[ -x tools/inplace ] && ./tools/inplace > $OUT/inplace.txt 2>&1   # what in-place ring updates of various plane sets can reach
# 5. in-kernel timelines, launch period, the rows ranked next, size sweep, ablations
FW_FIFO=0 FW_RANGE=0 timeout 200 python tools/launch_gaps.py > $OUT/launch_gaps.txt 2>&1   # (in-kernel timestamps exist in the general path's kernels)
FW_FIFO=0 FW_RANGE=0 timeout 200 python tools/tile_timeline.py > $OUT/tile_timeline.txt 2>&1   # (instrumentation of the general path's kernels)
FW_FIFO=0 FW_RANGE=0 FW_TL_JITTER=1 timeout 200 python tools/tile_timeline.py > $OUT/tile_timeline_variable_dt.txt 2>&1
timeout 300 python tools/fused_sizes.py > $OUT/fused_sizes.txt 2>&1
FW_FIFO=0 FW_RANGE=0 timeout 300 python tools/fused_sizes.py > $OUT/fused_sizes_general_path.txt 2>&1
timeout 300 python tools/dbg_modes.py > $OUT/dbg_modes.txt 2>&1
timeout 300 python tools/var_dt.py 400 > $OUT/var_dt.txt 2>&1
FW_FIFO=0 FW_RANGE=0 timeout 300 python tools/var_dt.py 400 > $OUT/var_dt_general_path.txt 2>&1
# 6. the other BASELINE configs on one GPU, the small-emitter regime and the host half of fw_step
timeout 600 python tools/bench_configs.py > $OUT/configs.txt 2>&1
timeout 300 python tools/small_emitters_gpu.py > $OUT/small_emitters.txt 2>&1
FW_HOST_PROF=1 timeout 300 python tools/small_emitters.py >> $OUT/small_emitters.txt 2>&1
# 6b. range rings (round 3): same-box A/B against the compacting path and the two knobs that lost, the size sweep, and
#     rocprofv3 kernel-trace summaries of configs[2] / configs[4]'s share / configs[3] (tools/prof_configs.sh)
timeout 900 tools/range_ab.sh "" "FW_RANGE=0" > $OUT/range_ab.txt 2>&1   # (the knobs FW_RANGE_DEVREC / FW_RANGE_FOLD of round 3 lost and are gone)
(timeout 600 python tools/range_sweep.py; echo "FW_RANGE=0:"; FW_RANGE=0 timeout 600 python tools/range_sweep.py) > $OUT/range_sweep.txt 2>&1
timeout 1200 tools/prof_configs.sh $TAG > /dev/null 2>&1; cp gpurun_out/prof_configs_$TAG/*_kernel_stats.csv gpurun_out/prof_configs_$TAG/*_trace_summary.txt gpurun_out/prof_configs_$TAG/*_bench.json $OUT/ 2>/dev/null
timeout 600 python tools/nt_sweep.py > $OUT/nt_sweep.txt 2>&1   # plain against fully non-temporal ring kernels over 0.2-2.6 GB
[ -f bevy_firework_amd/csrc/libfirework_hip_timeline.so ] && (timeout 300 python tools/range_timeline.py; FW_TL_EMITTERS=256x65536 timeout 300 python tools/range_timeline.py) 2>&1 | cut -c1-2500 > $OUT/range_timeline_run.txt
(timeout 300 python tools/bench_next_rows.py; echo "FW_DERIVED=0 (scale / colour planes kept next to the records):"; FW_DERIVED=0 timeout 300 python tools/bench_next_rows.py) > $OUT/next_rows.txt 2>&1
# 7. configs[3] (Nested): rocprofv3 kernel stats of the steady state
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/nested -o nested --output-format csv -- python $R/tools/nested_prof.py > $R/$OUT/nested_step.txt 2>/dev/null; cd $R
python profiles/analyze_trace.py $OUT/nested/nested_kernel_trace.csv 400 > $OUT/nested_trace_summary.txt 2>&1
(timeout 200 python tools/nested_ab.py; echo 'FW_FIFO_NESTED=0:'; FW_FIFO_NESTED=0 timeout 200 python tools/nested_ab.py) > $OUT/nested_ab.txt 2>&1   # configs[3] on rings / on the general path
cp $OUT/nested/nested_kernel_stats.csv $OUT/configs3_nested_kernel_stats.csv 2>/dev/null
[ -x tools/launchgap ] && timeout 200 ./tools/launchgap > $OUT/launchgap.txt 2>&1
rm -rf $OUT/${TAG}_stats_kernel_trace.csv $OUT/pmc/*kernel_trace.csv $OUT/nested   # large; the summaries are what is kept
ls $OUT
