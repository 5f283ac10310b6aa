export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
for d in ${DBGS:-0 2 64 68}; do echo "FW_DEBUG=$d"; FW_DEBUG=$d python bench.py --no-cpu --no-extras --steps 400 2>&1 | python tools/print_bench.py; done
