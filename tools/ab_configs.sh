export FW_ENABLE_KNOBS=1   # the library honours its A/B switches only with this set
# A/B of library builds on the other configs: tools/ab_configs.sh c3 variants/a.so variants/b.so ...   (c1 c3 c4 c5 of tools/bench_configs.py)
W=$1; shift
for so in "$@"; do echo "$so"; FW_LIB_PATH=$PWD/$so python tools/bench_configs.py $W 2>&1 | cut -c1-150; done
