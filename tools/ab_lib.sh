export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# A/B of library builds on configs[1]: tools/ab_lib.sh variants/a.so variants/b.so ...   (run on the GPU box)
for so in "$@"; do echo "$so"; FW_LIB_PATH=$PWD/$so python bench.py --no-cpu --no-extras --steps 400 2>&1 | python tools/print_bench.py; done
