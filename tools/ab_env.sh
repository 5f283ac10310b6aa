export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# usage: VAR=NAME VALS="a b c" bash tools/ab_env.sh   -- configs[1] kernel time under each value of an environment knob
for v in $VALS; do echo "$VAR=$v"; env $VAR=$v ${EXTRA:-} python bench.py --no-cpu --no-extras --steps 400 2>&1 | python tools/print_bench.py; done
