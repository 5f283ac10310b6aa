#!/bin/bash
# Round 6: is ONE launch over all range rings of a world as fast as the memory system allows?  The rehearsal of N ranks on one device
# (examples/sharded --ranks-on-one-device N) ran 8 x 512 emitters x 8192 concurrently in 316 us per frame -- the same particles take 441 us
# in one context.  Two candidate causes, separated here: (1) several launches in flight at once (N contexts, the total work fixed),
# (2) the cache form (a 278 MB context streams plain, a 2.2 GB one non-temporally: FW_NT_MB).     tools/r06_concurrency.sh   (GPU box)
run() { echo "## $*"; "$@" 2>&1 | grep -E "^ranks_on_one_device|firework error"; }
for shape in "4096 8192" "256 65536"; do
  set -- $shape
  for R in 1 2 4 8; do run ./examples/sharded --ranks-on-one-device $R --emitters $1 --live $2 --frames 200 --reduce-every 16; done
  export FW_ENABLE_KNOBS=1
  FW_NT_MB=1000000 run ./examples/sharded --ranks-on-one-device 1 --emitters $1 --live $2 --frames 200 --reduce-every 16
  FW_NT_MB=0 FW_NT_WO_MB=0 run ./examples/sharded --ranks-on-one-device 8 --emitters $1 --live $2 --frames 200 --reduce-every 16
  FW_NT_MB=0 FW_NT_WO_MB=0 run ./examples/sharded --ranks-on-one-device 1 --emitters $1 --live $2 --frames 200 --reduce-every 16
  unset FW_ENABLE_KNOBS
done
