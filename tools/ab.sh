#!/bin/bash
export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# A/B bench of library builds: tools/ab.sh variants/a.so variants/b.so ...   (run on the GPU box)
for rep in 1 2 3; do
  for so in "$@"; do
    FW_LIB_PATH=$PWD/$so timeout 300 python bench.py --steps 400 --warmup 60 --no-cpu --no-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'rep$rep', round(d['ms_per_step']*1000,2), 'us/step', d['config']['live_particles'])"
  done
done
