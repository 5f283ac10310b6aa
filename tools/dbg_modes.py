"""Kernel time of the 1M-particle frame under the profiling-only ablations (FW_DEBUG bits; results are wrong when set):
what each structural feature of the update kernel costs.  Run on the GPU box: python tools/dbg_modes.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = (
    ("FIFO ring path (fw_k_update_fifo; configs[1]'s particle type has one lifetime value)", {},
     ((0, "full kernel"), (2, "no arithmetic (planes streamed through, same loads and stores)"))),
    ("general path (FW_FIFO=0 FW_RANGE=0: fw_k_update_stream, survivor forecast)", {"FW_FIFO": "0", "FW_RANGE": "0"},
     ((0, "full kernel"), (16, "forecast table read TWICE per tile (the difference = what the read costs)"),
      (2, "no arithmetic (planes streamed through)"), (18, "both"))),
)
for title, extra, modes in SETS:
    print(title)
    for dbg, what in modes:
        env = dict(os.environ, FW_DEBUG=str(dbg), **extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extras", "--steps", "400"],
                             env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(f"  FW_DEBUG={dbg:2d}  {d['roofline']['avg_kernel_us']:6.2f} us kernel  {d['ms_per_step'] * 1e3:6.2f} us step  "
              f"{d['roofline']['particles_per_launch']:9.0f} particles/launch   {what}")
