#!/bin/bash
# The committed profile artefacts of round 6 (run on the GPU box, from the repo root):  tools/profile_r06.sh [part ...]
#   parts: bench stats pmc configs inst vardt misc   (default: all)
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
TAG=r06
R=$PWD; OUT=gpurun_out/profile_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
PARTS=${@:-bench stats pmc configs inst vardt misc}
has() { [[ " $PARTS " == *" $1 "* ]]; }
prof() {  # prof <name> <tail> <command...>: rocprofv3 kernel trace + stats of a command -> <name>_kernel_stats.csv, <name>_trace_summary.txt
  local name=$1 tail=$2; shift 2
  rm -rf $R/$OUT/tmp_$name
  (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/tmp_$name -o $name --output-format csv -- "$@" > $R/$OUT/${name}_run.json 2> $R/$OUT/${name}.err)
  local f=$(find $OUT/tmp_$name -name "${name}_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  local t=$(find $OUT/tmp_$name -name "${name}_kernel_trace.csv" | head -1); [ -n "$t" ] && python profiles/analyze_trace.py $t $tail > $OUT/${name}_trace_summary.txt 2>&1
  rm -rf $OUT/tmp_$name
}
pmc2() {  # pmc2 <name> <command...>: FETCH_SIZE and WRITE_SIZE in their own passes (kernel-trace only) -> <name>_pmc.txt
  local name=$1; shift
  mkdir -p $OUT/pmc_$name
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp; timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/pmc_$name -o $c --output-format csv -- "$@" > $R/$OUT/pmc_$name/$c.log 2>&1)
  done
  (echo "== $name  (KiB per launch, last 60 dispatches; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md)"; python profiles/analyze_pmc.py $OUT/pmc_$name 60) > $OUT/${name}_pmc.txt 2>&1
  rm -rf $OUT/pmc_$name
}
# 1. the bench line itself (default flags) and with the driver's flags
if has bench; then
  timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/bench_20_steps.json 2>> $OUT/bench.err
fi
# 2. rocprofv3 kernel trace + stats of the same command (the kernel of the headline configuration only)
if has stats; then
  prof ${TAG}_stats 600 python $R/bench.py --no-cpu --no-extras
  mv $OUT/${TAG}_stats_trace_summary.txt $OUT/trace_summary.txt 2>/dev/null
fi
# 3. PMC passes (own runs, kernel-trace only): the headline kernel
if has pmc; then
  ./tools/pmc.sh $OUT/pmc > /dev/null 2>&1
  python profiles/analyze_pmc.py $OUT/pmc > $OUT/pmc_summary.txt
  rm -rf $OUT/pmc
fi
# 4. the other configs: kernel stats + trace summaries (range rings and FW_RANGE=0), PMC traffic of their update kernels
if has configs; then
  for cfg in "c3 configs2" "c5 configs4_share" "c4 configs3_nested" "cc stress_test_collision" "c3 configs2_compacting FW_RANGE=0"; do
    set -- $cfg
    [ -n "$3" ] && export $3 || unset FW_RANGE
    prof $2 300 python $R/tools/bench_configs.py $1
  done
  unset FW_RANGE
  for w in c3 c5 c4; do pmc2 pmc_$w python $R/tools/bench_configs.py $w; done
  cat $OUT/pmc_c3_pmc.txt $OUT/pmc_c5_pmc.txt $OUT/pmc_c4_pmc.txt > $OUT/pmc_configs.txt 2>/dev/null; rm -f $OUT/pmc_c?_pmc.txt
  timeout 600 python tools/bench_configs.py c1 c3 c4 c5 cc > $OUT/configs.txt 2>&1
fi
# 5. the frame a renderer asks for: update + 64-byte instance records (configs[1] plain attach, configs[2] windowed attach)
if has inst; then
  prof instance_records_configs1 300 python $R/tools/r06_instance_records.py c1
  prof instance_records_configs2 100 python $R/tools/r06_instance_records.py c2
  pmc2 instance_records_configs2 python $R/tools/r06_instance_records.py c2
fi
# 6. the compacting path under a dt that never repeats: schedules side by side, kernel stats and HBM traffic of
#    the look-back kernel (no earlier round has a PMC pass of it)
if has vardt; then
  timeout 900 python tools/r06_compacting_vardt.py > $OUT/compacting_vardt.txt 2>&1
  cat > /tmp/r06_vardt_only.py <<'EOF'
import os, sys
os.environ["FW_ENABLE_KNOBS"] = "1"; os.environ["FW_RANGE"] = "0"
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bevy_firework_amd import workloads
from bevy_firework_amd.system import ParticleSystem
jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]
with ParticleSystem(seed=workloads.SEED) as ps:
    for e, (s_, tf_) in enumerate(workloads.many_emitters(256, 65536)):
        ps.spawn(s_, tf_, uid=e)
    ps.update(jit[0])
    for k in range(160):
        ps.step(jit[k % 64])
    ps.synchronize()
print("configs[2], FW_RANGE=0, dt = 1/60 (1 + 0.1 sin(0.7 k)): 160 frames")
EOF
  prof compacting_vardt_lookback 60 python /tmp/r06_vardt_only.py
  pmc2 compacting_vardt_lookback python /tmp/r06_vardt_only.py
fi
# 7. error budget of the parity tolerance, path thresholds on this box, N ranks rehearsed on one device, the examples, the soak
if has misc; then
  timeout 1500 python tools/r06_error_budget.py > $OUT/parity_error_budget.txt 2>&1
  timeout 900 python tools/threshold_sweep.py > $OUT/threshold_check.txt 2>&1
  (timeout 600 ./examples/sharded --ranks-on-one-device 8 --emitters 4096 --live 8192 --frames 200 --reduce-every 16 | grep -v '^frame '; echo "--- one context with one GPU's share (512 emitters), for comparison:"; timeout 600 ./examples/sharded --ranks-on-one-device 1 --emitters 512 --live 8192 --frames 200 --reduce-every 16 | grep -v '^frame ') > $OUT/ranks_on_one_device.txt 2>&1
  timeout 600 python tools/r04_examples_latency.py > $OUT/examples_latency.txt 2>&1
  FW_SOAK_FRAMES=10000 timeout 1500 python tools/soak_r05.py > $OUT/soak_r06.txt 2>&1
fi
find $OUT -name "*kernel_trace.csv" -delete
ls $OUT
