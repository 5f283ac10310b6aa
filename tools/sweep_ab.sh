export FW_ENABLE_KNOBS=1
for v in "" "FW_RANGE_SPREAD_NEW=1" "" "FW_RANGE_SPREAD_NEW=1"; do echo "== ${v:-default}"; env $v python tools/range_sweep.py; done
