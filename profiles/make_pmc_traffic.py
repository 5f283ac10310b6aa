#!/usr/bin/env python3
"""profiles/<round>/pmc_summary.txt -> profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM-side bytes per fw_k_update launch from the rocprofv3 PMC passes (tools/pmc.sh): FETCH_SIZE and WRITE_SIZE are
reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams, so it is doubled
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported."""
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06/pmc_summary.txt"
txt = open(src).read()
# the steady-state kernel of the bench: the in-place FIFO ring update (configs[1]'s particle type has one lifetime value; no
# attached instance buffers, base / emissive / scale planes all written: write mask 7); with FW_FIFO=0 the streaming
# kernel of the general path (forecast frames, inline spawn ops, per-tile forecast entries, lone segment)
for kern in ("fw_k_update_fifo<false, 0, 0, false, 4>", "fw_k_update_fifo<false, 7, 0, false, 4>", "fw_k_update_fifo<false, 7, 0, false>", "fw_k_update_fifo<false, 7, 0>", "fw_k_update_fifo<false, 7>", "fw_k_update_stream<1, false, false, true>"):
    m = re.search(re.escape(kern) + r"\s+FETCH_SIZE=([0-9.e+]+)", txt)
    if m:
        break
fetch = float(m.group(1))
write = float(re.search(re.escape(kern) + r"\s+WRITE_SIZE=([0-9.e+]+)", txt).group(1))
out = {
    "source": src,
    "kernel": kern,
    "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
    "fetch_bytes_corrected": 2 * fetch * 1024, "write_bytes": write * 1024,
    "fw_k_update_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "note": "FETCH_SIZE x2 (gfx950 correction for 16 B/lane streams); separate --pmc passes of `bench.py --steps 100`",
}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(out)
