import sys, json
d = json.loads(sys.stdin.read())
r = d["roofline"]
def p(name, x):
    print("  %-28s %8.1f us  %6.0f GB/s (%s)" % (name, x["avg_kernel_us"], x["achieved"], x["update_path"]))
p("configs1", r); p("configs1 variable dt", r["variable_dt"])
h = r["hbm_resident"]; p("configs2", h); p("configs2 var dt", h["variable_dt"]); p("configs2 compacting", h["compacting_path"]); p("configs2 compacting var dt", h["compacting_path"]["variable_dt"]); p("configs2 every plane", h["with_every_plane_kept"])
p("ring 16M", r["hbm_resident_ring"])
c = r["configs4_one_gpu"]; print("  configs4 one gpu %.1f us" % (c["ms_per_step"] * 1000))
