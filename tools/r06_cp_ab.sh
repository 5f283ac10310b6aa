#!/bin/bash
# Round 6: the component-plane build (in-tree) against main's library (variants/main6), same box, interleaved.   tools/r06_cp_ab.sh   (GPU box)
export FW_ENABLE_KNOBS=1
one() {  # one <label> <lib or ""> <env...> -- <cmd...>
  local label=$1 lib=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local out
  out=$(env ${lib:+FW_LIB_PATH=$PWD/$lib} "${envs[@]}" timeout 300 "$@" < /dev/null 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l)
    if 'us_per_step' in d: print('%-58s %8.1f us per step  update kernels %8.1f us' % (d['config'][:58], d['us_per_step'], d.get('update_kernels_us_per_frame', 0)))
    elif 'ms_per_step' in d: print('%-58s %8.2f us per step  kernel %8.2f us' % ('configs[1] (bench.py --no-cpu --no-extras)', d['ms_per_step'] * 1e3, d['roofline']['avg_kernel_us']))
")
  echo "$label | $out"
}
for rep in 1 2; do
  for v in "main6   variants/main6/libfirework_hip.so" "planes  "; do
    set -- $v; name=$1; lib=$2
    one "$name" "$lib" -- python bench.py --no-cpu --no-extras --steps 400
    one "$name" "$lib" -- python tools/bench_configs.py c3 c4 c5
    one "$name compacting" "$lib" FW_RANGE=0 -- python tools/bench_configs.py c3
    one "$name compacting" "$lib" FW_RANGE=0 FW_FIFO=0 -- python tools/bench_configs.py c4
  done
done
