#!/usr/bin/env python3
"""bench.py -- particles updated/sec on the stress-test workloads (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the hot path (fw_step = spawn_particles + update_particles with
compaction) over the whole resident particle set; inputs are already in HBM when the timed
region starts.

Workloads (BASELINE.json `configs`, built by bevy_firework_amd/workloads.py):
  N = 1 (default)  configs[1]: one emitter, rate 1e6/s, lifetime 1 s -> 983 333 live in steady
                   state (the reference drops one frame of emission per cycle wrap), Point emission,
                   linear 2-key scale/colour curves, dt = 1/60.  This is the headline `value`.
  N > 1 (default)  configs[4]: 4096 Sphere emitters x 8192 live (32M particles in total), emitter e on
                   rank e mod N (strong scaling: the total is fixed), no particle ever crosses GPUs; the
                   only exchange is the RCCL all-reduce of per-frame live counts, a bucket of
                   --reduce-every frames per collective, fed from a device ring the update kernel
                   writes (bevy_firework_amd/sharding.py -- the same class the tests drive).
  --workload configs1|configs4 forces either at any N (configs4 at N=1 is the base point of the curve).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment LAUNCHES the N ranks itself (it re-executes
itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one device per
rank) and exits non-zero when fewer than N devices are visible: it never falls back to one GPU or to configs[1].

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline      HBM roofline of the dominant kernel (fw_k_update*), from HIP events attached to each
                dispatch on the kernel's own stream; for N=1 also `hbm_resident` (configs[2], 16.8M particles:
                a working set far beyond the 256 MiB Infinity Cache) and `variable_dt` (configs[1] stepped with
                a jittering dt, what an unmodified Bevy `Update` schedule delivers, plugin.rs:26-31)
  cpu_baseline  the C oracle (a port of the reference's CPU algorithm, AoS + per-particle heap vector +
                clone/filter/collect) on a bounded sample: configs[1] on 1 thread (the reference runs one
                spawner on one core, core.rs:583-586) and, under `many_emitters`, configs[2]'s emitters on all
                host cores, one spawner per thread like par_iter_mut
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SURVEY_BYTES = 156  # 64 B read + 92 B written per particle-update (every plane rewritten)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
INFINITY_CACHE_BYTES = 256 << 20  # MI355X_MICROARCH.md: 256 MiB memory-side cache; a working set below it is not an HBM stream
# Round 6: scale, base colour and emissive colour are pure functions of (age, lifetime, initial_scale) (core.rs:601-605, 652-655)
# and the update no longer stores them for any type (FW_TYPE_DERIVED): 36 of SURVEY's 92 written bytes are not moved.
# `algorithmic_bytes_per_particle` is what the library reports for the path the type is on (fw_debug_update_path);
# `at_survey_156B_per_particle` keeps the SURVEY figure next to it -- a rate of bytes the kernel does NOT move, above the peak at
# configs[1]: a comparison with earlier rounds, not a bandwidth.


def path_bytes(ps):
    """(update path, algorithmic B / particle-update, moved B / particle-update) of the system's first spawner, from the
    library (fw_debug_update_path).  General (compacting, ping-pong) path: every plane is rewritten at the particle's new
    slot -- 64 B read + 100 B written, of which 8 B (initial_scale, lifetime: unchanged values) are not in SURVEY.md's
    156 B; a colour plane with a one-key gradient is not written at all (-16 B each, algorithmic and moved alike).
    FIFO ring path (one lifetime value: particles never move): 64 B read + only the planes the update changes."""
    h = next(iter(ps.spawners.values()))
    mode, moved, algo = h.update_path(0)
    return mode, algo, moved


def cpu_baseline(args, dt, world=1, check=False, workload="auto"):
    """Reference CPU path restated in C (oracle/), timed on this box: bounded samples of the same workloads.
    world == 1: configs[1] on one thread (the headline) + configs[2]'s emitters on all cores under `many_emitters`.
    world > 1: the workload of that line -- configs[4]'s emitters (8192 live each), one spawner per thread on all host cores,
    on rank 0 while the other ranks wait at the barrier (outside the timed region).
    check: the launch-plumbing test (tests/test_bench_launch.py) -- the same code on a sample of a few hundred particles."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    from bevy_firework_amd import workloads

    def many(emitters, live, fill2, frames2, label, per_thread=1):
        # one spawner per thread (the reference's par_iter_mut over spawners, core.rs:583-585): as many emitters as there
        # are host cores (ctypes releases the GIL inside the oracle calls)
        threads = max(1, min(emitters, os.cpu_count() or 1, 2 if check else 1 << 30))
        used = min(emitters, threads * per_thread)
        ems = workloads.many_emitters(emitters, live)[:used]
        sp = [oracle.OracleSpawner(s, seed=workloads.SEED, uid=e, transform=tf_) for e, (s, tf_) in enumerate(ems)]

        def run(o_, k):
            m = 0
            for _ in range(k):
                o_.spawn(dt)
                m += o_.count(0)
                o_.update(dt)
            return m

        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(lambda o_: run(o_, fill2), sp))
            t0 = time.perf_counter()
            n2 = sum(ex.map(lambda o_: run(o_, frames2), sp))
            el2 = time.perf_counter() - t0
        return {
            "value": n2 / el2, "unit": "particles/s", "cores": threads, "kind": "port",
            "sample": f"{label}: {used} of the {emitters} emitters x {live} live, one spawner per task on {threads} threads, "
                      f"{frames2} frames after a {fill2}-frame fill ({n2 // frames2} particles per frame); host has "
                      f"{os.cpu_count()} cores",
        }

    if world > 1 or workload == "configs4":
        if check:
            return many(8, 300, 5, 2, "launch check (not a measurement)")
        # (76 frames to the steady state of lifetimes up to 1.2 s; 8192-particle emitters are ~1 ms of one core per frame)
        return many(args.emitters, args.live_per_emitter, 76, 60, "configs[4]", per_thread=8)
    spawner, tf = workloads.one_million(rate=args.rate)
    o = oracle.OracleSpawner(spawner, seed=workloads.SEED, uid=0, transform=tf)
    fill = int(round(1.0 / float(dt))) + 2
    for _ in range(fill):
        o.step(dt)
    frames, n = args.cpu_frames, 0
    t0 = time.perf_counter()
    for _ in range(frames):
        o.spawn(dt)
        n += o.count(0)  # particles entering update_particles
        o.update(dt)
    el = time.perf_counter() - t0
    out = {
        "value": n / el, "unit": "particles/s", "cores": 1, "kind": "port",
        "sample": f"configs[1]: {frames} frames at {o.count(0)} live after a {fill}-frame fill, same settings/seed as the "
                  "GPU run; 1 thread because the reference runs one spawner on one core (core.rs:583-586); "
                  f"host has {os.cpu_count()} cores",
    }
    out["many_emitters"] = many(256, 65536, 76, 6, "configs[2]")
    return out


def roofline_dict(label, kt, per_launch, launches, mode, algo, moved):
    """the roofline object of one kernel: algorithmic bytes per launch / average launch duration against the HBM peak"""
    achieved = per_launch * algo / kt / 1e9
    return {"workload": label, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "particles_per_launch": per_launch, "avg_kernel_us": kt * 1e6, "launches": launches, "update_path": mode,
            "algorithmic_bytes_per_particle": algo, "moved_bytes_per_particle": moved,
            "at_survey_156B_per_particle": {"achieved": per_launch * SURVEY_BYTES / kt / 1e9,
                                            "frac": per_launch * SURVEY_BYTES / kt / 1e9 / HBM_PEAK_GBS,
                                            "note": "SURVEY 8(d)'s 156 B x particles / kernel time: bytes the kernel no longer "
                                                    "moves (scale / colour planes are left to the readers), kept for comparison "
                                                    "with rounds 1-5 only"},
            # which memory level the launch streams from: the planes it touches against the 256 MiB Infinity Cache
            "working_set_bytes": per_launch * moved,
            "bound": "infinity_cache" if per_launch * moved < INFINITY_CACHE_BYTES else "hbm"}


def kernel_roofline(ps, step, frames, label):
    """HIP-event timing of the update dispatches over `frames` calls of step(); algorithmic GB/s and fraction of peak"""
    ps.kernel_timing(True)
    for k in range(frames):
        step(k)
    ev_ms, launches, particles = ps.kernel_timing_read()
    ps.kernel_timing(False)
    if not launches:
        return None
    mode, algo, moved = path_bytes(ps)
    return roofline_dict(label, ev_ms * 1e-3 / launches, particles / launches, launches, mode, algo, moved)


def gather_ranks(dist, world, device, elapsed, updated, live, roof):
    """one row per rank {wall time of the timed region, particles updated, live particles, average update-kernel duration (us),
    particles per launch, launches}: what rank 0 needs for the whole-job line (all_gather: RCCL on the GPU, gloo in the
    launch check)"""
    row = [float(elapsed), float(updated), float(live), float(roof["avg_kernel_us"]) if roof else 0.0,
           float(roof["particles_per_launch"]) if roof else 0.0, float(roof["launches"]) if roof else 0.0]
    if dist is None:
        return [row]
    import torch

    t = torch.tensor(row, dtype=torch.float64, device=device)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    return [[float(v) for v in x.tolist()] for x in allt]


def assemble_line(args, world, workload, rows, roof, extras, cpu, hist, rccl_ranks, measured_copy, reduce_every):
    """rank 0's ONE JSON line from the per-rank rows (gather_ranks), rank 0's roofline object, the extra measurements and the
    CPU baseline.  Pure bookkeeping: the launch check (tests/test_bench_launch.py) runs it on stand-in numbers."""
    elapsed = max(r[0] for r in rows)  # the slowest rank's clock around the barrier-bracketed region
    updated, live = int(sum(r[1] for r in rows)), int(sum(r[2] for r in rows))
    per_rank_ms = [r[0] / args.steps * 1e3 for r in rows]
    value = updated / elapsed
    if workload == "configs1":
        wl = ("configs[1]: 1 emitter x rate 1e6/s x lifetime 1 s per GPU (983 333 live), Point emission, linear 2-key "
              "scale/colour curves, dt=1/60, spawn + update + removal of the dead (order kept) every step")
        scaling, em_total = "weak", world
    else:
        wl = (f"configs[4]: {args.emitters} Sphere emitters x {args.live_per_emitter} live (radial velocity, lifetimes "
              "0.8-1.2 s, per-emitter constants), emitter e on rank e mod N, dt=1/60, spawn + update + order-preserving "
              "removal of the dead every step, RCCL all-reduce of per-frame live counts")
        scaling, em_total = "strong", args.emitters
    out = {
        "metric": "particles updated/sec (stress_test, 1M live)" if workload == "configs1" else
                  "particles updated/sec (stress_test emitter array)",
        "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": wl, "emitters_total": em_total, "live_particles": live, "sharding": "emitter e -> rank e mod N",
            "rccl_ranks": rccl_ranks,  # dist.get_world_size() of the nccl (= RCCL) group; 1 = no group
            "self_launched": os.environ.get("FW_BENCH_SELF_LAUNCHED") == "1",
            "live_count_allreduce_every": reduce_every,
            "per_rank_ms_per_step": per_rank_ms,
            "per_rank_live": [int(r[2]) for r in rows],
            "allreduced_live_count_last_frame": hist[-1] if hist else None,
            "speedup_vs_configs4_one_gpu": (extras["configs4_one_gpu"]["ms_per_step"] / (elapsed / args.steps * 1e3)
                                            if world > 1 and extras.get("configs4_one_gpu") else None),
            "update_mode": os.environ.get("FW_UPDATE_MODE", "fused"),
            "records_in_host_written_device_memory": extras.get("param_bar"),
        },
        "hbm_gbs_algorithmic_whole_step": value * (roof["algorithmic_bytes_per_particle"] if roof else SURVEY_BYTES) / 1e9,
    }
    if roof:
        # HBM-side bytes per launch of this kernel: PMC counters cannot be read from inside the process, so this is the
        # figure of the separate `rocprofv3 --pmc` passes over this same command (tools/pmc.sh ->
        # profiles/pmc_traffic.json), labelled as such -- not a measurement of the run that prints it
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and workload == "configs1" and world == 1:
            try:
                tj = json.load(open(tp))
                traffic = tj.get("fw_k_update_bytes_per_launch")
                traffic_src = ("traffic_from_profile: profiles/pmc_traffic.json (" + str(tj.get("source", "rocprofv3 --pmc passes")) +
                               "), not measured by this run")
            except Exception:
                traffic = None
        fifo = roof.get("update_path") == "fifo"
        roof.update({
            "kernel": ("fw_k_update_fifo (in-place ring update of a one-lifetime particle type, any dt)" if fifo else
                       "fw_k_update_range (in-place range rings: young part in place, old part compacted in place, any dt)"
                       if roof.get("update_path") == "range" else
                       "fw_k_update_stream (forecast frames; fw_k_update<fused> when dt changes)"),
            "traffic": traffic, "traffic_source": traffic_src,
            "measured_hbm_copy_GBps": measured_copy / 1e9,
            "note": ("at 1M particles the ring's planes (128 B per particle: four state planes read and written) are resident in the "
                     "256 MiB Infinity Cache: `bound` says so, `frac` is a cache-resident figure against the HBM peak, not an HBM "
                     "one; `hbm_resident` is configs[2], a 16.8M-particle working set whose lifetimes are a "
                     "range: in-place range rings (fw_k_update_range), with the compacting kernels on the same workload "
                     "under `compacting_path`; `with_instance_records` is the frame a renderer asks for (update + 64-byte "
                     "ParticleInstance records, render.rs:95-115)" if fifo and world == 1 else
                     "one GPU's share of configs[4] (lifetime ranges: in-place range rings), 64 B algorithmic per particle "
                     "(position+age and velocity in and out); the share's ~280 MB exceed the 256 MiB Infinity Cache" if (world > 1 or workload == "configs4") else
                     "at 1M particles the ping-pong working set sits in the 256 MiB Infinity Cache; "
                     "`hbm_resident` is the same kernel on a 16.8M-particle working set"),
            "timing": "hipEvent start/stop attached to each update dispatch on the context's stream "
                      "(hipExtLaunchKernel: the packet's begin/end timestamps, the same duration rocprofv3 "
                      "--kernel-trace reports); second pass over the same steady state, kept out of `value`",
        })
        if world > 1:
            # every rank timed its own dispatches; the headline figures of the object are the SLOWEST rank's
            algo = roof["algorithmic_bytes_per_particle"]
            per_rank = []
            for r, row in enumerate(rows):
                kt_us, ppl = row[3], row[4]
                ach = ppl * algo / (kt_us * 1e-6) / 1e9 if kt_us > 0 else None
                per_rank.append({"rank": r, "avg_kernel_us": kt_us, "particles_per_launch": ppl, "launches": int(row[5]),
                                 "achieved": ach, "frac": ach / HBM_PEAK_GBS if ach is not None else None})
            timed = [x for x in per_rank if x["achieved"] is not None]
            if timed:
                worst = min(timed, key=lambda x: x["frac"])
                roof.update({"achieved": worst["achieved"], "frac": worst["frac"], "avg_kernel_us": worst["avg_kernel_us"],
                             "particles_per_launch": worst["particles_per_launch"], "rank_reported": worst["rank"],
                             "kernel_us_min": min(x["avg_kernel_us"] for x in timed),
                             "kernel_us_max": max(x["avg_kernel_us"] for x in timed)})
            roof["per_rank"] = per_rank
            roof["workload"] = (f"{workload}: each rank's share ({args.emitters // world} emitters x {args.live_per_emitter} live); "
                                "achieved / frac are the slowest rank's, per-rank figures under `per_rank`")
        roof.update(extras)
    out["roofline"] = roof
    if extras.get("configs4_one_gpu"):
        out["config"]["configs4_one_gpu"] = extras["configs4_one_gpu"]
    out["cpu_baseline"] = cpu
    return out


def flush_c_stdio():
    """fflush(NULL): whatever native libraries left in the C-level stdout / stderr buffers goes out now, not at exit"""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks here (one per device) and hand their exit code back.  Never a
    silent fallback: fewer than N visible devices is an error (the curve is N ranks or nothing)."""
    import socket
    import subprocess

    check = os.environ.get("FW_BENCH_LAUNCH_CHECK") == "1"  # CPU test of the launch plumbing: gloo ranks, no device
    if not check:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible MI355X devices, found {have}; "
                             "refusing to fall back to fewer ranks\n")
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)
    env["FW_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def launch_check(args, world, rank):
    """FW_BENCH_LAUNCH_CHECK=1 (tests/test_bench_launch.py, CPU only): the ranks prove that they exist -- a gloo group of
    `world` processes, one all-reduce -- and rank 0 prints the line the real run would print, assembled by the SAME code
    (gather_ranks, assemble_line, cpu_baseline) from stand-in numbers: marked `launch_check`, never a measurement."""
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    workload = args.workload if args.workload != "auto" else ("configs1" if world == 1 else "configs4")
    # stand-ins: rank r "ran" for 1 + r/10 s, updated 1000 (r + 1) particles, its kernel took 50 + r us per launch
    roof = roofline_dict(workload, (50.0 + rank) * 1e-6, 100.0 * (rank + 1), 3, "range", 64, 68)
    rows = gather_ranks(dist, world, "cpu", 1.0 + 0.1 * rank, 1000 * (rank + 1), 100 * (rank + 1), roof)
    if rank == 0:
        cpu = None if args.no_cpu else cpu_baseline(args, 1.0 / 60.0, world, check=True, workload=workload)
        out = assemble_line(args, world, workload, rows, roof, {}, cpu, [], dist.get_world_size(), 0.0, args.reduce_every)
        out.update({"launch_check": True, "rccl_ranks": dist.get_world_size(), "ranks_seen": int(t.item()),
                    "self_launched": os.environ.get("FW_BENCH_SELF_LAUNCHED") == "1"})
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=120)
    ap.add_argument("--workload", choices=["auto", "configs1", "configs4"], default="auto")
    ap.add_argument("--rate", type=float, default=1.0e6, help="configs1: particles/s per emitter (lifetime 1 s)")
    ap.add_argument("--emitters", type=int, default=4096, help="configs4: emitters in total (sharded e mod N)")
    ap.add_argument("--live-per-emitter", type=int, default=8192)
    ap.add_argument("--reduce-every", type=int, default=16, help="frames per bucketed live-count all-reduce")
    ap.add_argument("--cpu-frames", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP events")
    ap.add_argument("--no-extras", action="store_true", help="N=1: skip the hbm_resident / variable_dt / configs4 figures")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: one rank per GPU, no fallback")
    if os.environ.get("FW_BENCH_LAUNCH_CHECK") == "1":
        return launch_check(args, int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")))

    import numpy as np
    import torch

    from bevy_firework_amd import sharding, workloads
    from bevy_firework_amd.system import ParticleSystem

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the particle path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} devices are visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("FW_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # RCCL writes a version banner to the C-level stdout when its communicator is created -- buffered there, it would come out at
        # process exit, AFTER the JSON line.  Create the communicator now and flush C stdio: the line stays the last thing on stdout.
        _w = torch.zeros(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(_w)
        torch.cuda.synchronize()
        flush_c_stdio()

    dt = np.float32(1.0 / 60.0)
    workload = args.workload if args.workload != "auto" else ("configs1" if world == 1 else "configs4")
    stream = torch.cuda.Stream()

    def make_system():
        return ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream)

    if workload == "configs1":  # every rank its own 1M-particle emitter when N > 1 (weak; not the default there)
        spawners = [workloads.one_million(rate=args.rate) for _ in range(world)]
        fill = int(round(1.0 / float(dt))) + 2
    else:
        spawners = workloads.many_emitters(args.emitters, args.live_per_emitter)
        fill = 76  # lifetimes in [0.8, 1.2] s
    sh = sharding.ShardedParticleSystem(make_system, spawners, rank, world, reduce_every=args.reduce_every,
                                        torch_stream=stream, exchange=dist is not None)
    ps = sh.system

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n_steps):
        for _ in range(n_steps):
            sh.step(dt)

    # fill to steady state (setup), then the untimed warm-up
    sh.update(dt)  # the first frame also pushes the spawner transforms
    run(fill)
    run(args.warmup)
    barrier()
    before = ps.updated_total()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    updated = ps.updated_total() - before
    live = ps.live_count()
    # Second pass over the same steady state with a hipEvent pair attached to every update dispatch on the kernel's
    # own stream (hipExtLaunchKernel start/stop events: the packet's begin / end timestamps, i.e. the duration
    # rocprofv3 --kernel-trace reports).  A separate pass so that per-dispatch signals cannot touch `value`.
    # (N > 1: every rank times its own share -- the line reports the slowest rank's kernel and all of them under `per_rank`)
    roof, measured_copy = None, 0.0
    if not args.no_events:
        roof = kernel_roofline(ps, lambda k: ps.step(dt), min(args.steps, 1000), workload)
        if rank == 0:
            measured_copy = ps.measure_copy_bandwidth(1 << 30, 20)  # float4 copy, 1 GiB -> 1 GiB (read + written bytes / s)
    barrier()

    rows = gather_ranks(dist, world, "cuda", elapsed, updated, live, roof)
    hist = sh.global_live_history if dist is not None else []  # the RCCL-reduced per-frame totals (brought to the host only here)

    extras = {}
    # (per-frame records / small op tables in device memory the host writes through the large BAR, or in pinned host memory: fw_ctx::param_bar)
    extras["param_bar"] = bool(ps.param_bar()) if ps is not None and hasattr(ps, "param_bar") else None
    rccl_ranks = dist.get_world_size() if dist is not None else 1
    # N > 1: the base point of the strong-scaling curve -- ALL emitters of configs[4] on rank 0's GPU alone, measured in
    # the same process right after the timed region (the other ranks wait at the barrier), so the line carries
    # speedup_vs_configs4_one_gpu next to the sharded figure
    if world > 1 and workload == "configs4" and not args.no_extras:
        if rank == 0:
            with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p4:
                for e, (s_, tf_) in enumerate(workloads.many_emitters(args.emitters, args.live_per_emitter)):
                    p4.spawn(s_, tf_, uid=e)
                p4.update(dt)
                for _ in range(76 + 10):
                    p4.step(dt)
                p4.synchronize()
                b0 = p4.updated_total()
                t1 = time.perf_counter()
                for _ in range(40):
                    p4.step(dt)
                p4.synchronize()
                el = time.perf_counter() - t1
                extras["configs4_one_gpu"] = {"particles_per_s": (p4.updated_total() - b0) / el, "ms_per_step": el / 40 * 1e3,
                                              "live_particles": p4.live_count(),
                                              "workload": f"{args.emitters} emitters x {args.live_per_emitter} live on one GPU "
                                                          "(rank 0's device, same process, after the timed region)"}
        barrier()
    if rank == 0 and world == 1 and workload == "configs1" and not args.no_events and not args.no_extras:
        # (a) variable dt on the SAME system: a host that steps with the wall-clock delta never repeats dt bit for bit
        jit = [np.float32((1.0 / 60.0) * (1.0 + 0.1 * np.sin(0.7 * k))) for k in range(64)]
        for k in range(32):
            ps.step(jit[k % 64])
        extras["variable_dt"] = kernel_roofline(ps, lambda k: ps.step(jit[k % 64]), 300,
                                                "configs[1] stepped with dt = 1/60 * (1 + 0.1 sin(0.7 k))")
        ps.close()
        ps = None
        # (b) a working set far beyond the Infinity Cache: configs[2], 256 emitters x 64Ki (16.8M particles)
        with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p2:
            for e, (s_, tf_) in enumerate(workloads.many_emitters(256, 65536)):
                p2.spawn(s_, tf_, uid=e)
            p2.update(dt)
            for _ in range(76 + 20):
                p2.step(dt)
            torch.cuda.synchronize()
            b0 = p2.updated_total()
            t1 = time.perf_counter()
            for _ in range(100):
                p2.step(dt)
            p2.synchronize()
            el = time.perf_counter() - t1
            whole = (p2.updated_total() - b0) / el
            extras["hbm_resident"] = kernel_roofline(p2, lambda k: p2.step(dt), 100,
                                                     "configs[2]: 256 emitters x 65 536 live (16.8M particles)")
            extras["hbm_resident"]["whole_step_particles_per_s"] = whole
            extras["hbm_resident"]["whole_step_ms"] = el / 100 * 1e3
            # ... stepped with a dt that never repeats (range rings need no forecast: the same speed is the claim)
            for k in range(32):
                p2.step(jit[k % 64])
            extras["hbm_resident"]["variable_dt"] = kernel_roofline(
                p2, lambda k: p2.step(jit[k % 64]), 100, "configs[2] stepped with dt = 1/60 * (1 + 0.1 sin(0.7 k))")
            # ... and with every plane kept (FW_NOSPIN=0: what the kernel moved before planes that cannot change were elided;
            # the knob is read when a context is created)
        saved_nospin, saved_knobs, saved_range = os.environ.get("FW_NOSPIN"), os.environ.get("FW_ENABLE_KNOBS"), os.environ.get("FW_RANGE")
        os.environ["FW_ENABLE_KNOBS"] = "1"  # (the library honours its A/B switches only with this set)
        try:
            # the same workload on the compacting path (FW_RANGE=0: every lifetime-range type ping-pongs and is compacted as a
            # whole, with the survivor forecast -- what round 2 measured), fixed dt and a dt that never repeats
            os.environ["FW_RANGE"] = "0"
            with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p2c:
                for e, (s_, tf_) in enumerate(workloads.many_emitters(256, 65536)):
                    p2c.spawn(s_, tf_, uid=e)
                p2c.update(dt)
                for _ in range(76 + 20):
                    p2c.step(dt)
                torch.cuda.synchronize()
                extras["hbm_resident"]["compacting_path"] = kernel_roofline(
                    p2c, lambda k: p2c.step(dt), 60, "configs[2] with FW_RANGE=0: the compacting kernels (survivor forecast)")
                for k in range(32):
                    p2c.step(jit[k % 64])
                tf0 = p2c.tf_frames()
                vd = kernel_roofline(
                    p2c, lambda k: p2c.step(jit[k % 64]), 60,
                    "... stepped with a dt that never repeats (round 6, threshold forecast: fw_k_fc_resolve in front of the streaming "
                    "schedule; `avg_kernel_us` is the update launch alone, `whole_frame` includes the resolve launch; rounds 1-5: "
                    "decoupled look-back)")
                vd["threshold_forecast_frames"] = p2c.tf_frames() - tf0  # of the 60 timed ones
                p2c.synchronize()
                tw = time.perf_counter()
                for k in range(60):
                    p2c.step(jit[k % 64])
                p2c.synchronize()
                wf = (time.perf_counter() - tw) / 60
                vd["whole_frame"] = {"us": wf * 1e6, "achieved": vd["particles_per_launch"] * vd["algorithmic_bytes_per_particle"] / wf / 1e9,
                                     "frac": vd["particles_per_launch"] * vd["algorithmic_bytes_per_particle"] / wf / 1e9 / vd["peak"]}
                extras["hbm_resident"]["compacting_path"]["variable_dt"] = vd
            if saved_range is None:
                del os.environ["FW_RANGE"]
            else:
                os.environ["FW_RANGE"] = saved_range
            os.environ["FW_NOSPIN"] = "0"
            with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p2b:
                for e, (s_, tf_) in enumerate(workloads.many_emitters(256, 65536)):
                    p2b.spawn(s_, tf_, uid=e)
                p2b.update(dt)
                for _ in range(76 + 20):
                    p2b.step(dt)
                torch.cuda.synchronize()
                extras["hbm_resident"]["with_every_plane_kept"] = kernel_roofline(
                    p2b, lambda k: p2b.step(dt), 60, "configs[2] with FW_NOSPIN=0: rotation and angular-velocity planes kept")
        finally:
            if saved_nospin is None:
                os.environ.pop("FW_NOSPIN", None)
            else:
                os.environ["FW_NOSPIN"] = saved_nospin
            if saved_range is None:
                os.environ.pop("FW_RANGE", None)
            else:
                os.environ["FW_RANGE"] = saved_range
            if saved_knobs is None:
                del os.environ["FW_ENABLE_KNOBS"]
            else:
                os.environ["FW_ENABLE_KNOBS"] = saved_knobs
        # (b') the ring path far beyond the Infinity Cache: configs[1]'s emitter at 16x the rate (16.4M particles in one ring)
        with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p3:
            s_, tf_ = workloads.one_million(rate=16.0 * args.rate)
            p3.spawn(s_, tf_, uid=0)
            p3.update(dt)
            for _ in range(62 + 20):
                p3.step(dt)
            torch.cuda.synchronize()
            extras["hbm_resident_ring"] = kernel_roofline(p3, lambda k: p3.step(dt), 100,
                                                          "configs[1] at 16x the rate: one ring of 15.7M live particles")
        # (b'') the frame a RENDERER asks for: update + the 64-byte ParticleInstance record of every survivor (render.rs:95-115, 403)
        # written by the update kernel itself into an attached device buffer (fw_spawner_attach_instances[_window]) -- what a Bevy
        # host runs every frame for a visible spawner (render.rs:382-403).  configs[1] (one FIFO ring) and configs[2] (256 range
        # rings; the windowed attach keeps them rings: the renderer draws buffer[first, first + count), render.rs:922-926).
        recs = {}
        for key, label, ems_, fill_, frames_, cap_, window_ in (
                ("configs1", "configs[1] + instance records", [workloads.one_million(rate=args.rate)], 62 + 20, 300, 1 << 20, False),
                ("configs2", "configs[2] + instance records (windowed)", workloads.many_emitters(256, 65536), 76 + 20, 60, 110000, True)):
            with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as pi:
                bufs = []
                for e, (s_, tf_) in enumerate(ems_):
                    h_ = pi.spawn(s_, tf_, uid=e)
                    b_ = torch.empty(cap_ * 64, dtype=torch.uint8, device=f"cuda:{local_rank}")
                    bufs.append(b_)
                    (h_.attach_instances_window if window_ else h_.attach_instances)(b_.data_ptr(), cap_, particle_type=0)
                pi.update(dt)
                for _ in range(fill_):
                    pi.step(dt)
                torch.cuda.synchronize()
                b0 = pi.updated_total()
                t1 = time.perf_counter()
                for _ in range(frames_):
                    pi.step(dt)
                pi.synchronize()
                el = time.perf_counter() - t1
                whole = (pi.updated_total() - b0) / el
                r_ = kernel_roofline(pi, lambda k: pi.step(dt), frames_, label)
                if r_:
                    r_["whole_step_particles_per_s"], r_["whole_step_ms"] = whole, el / frames_ * 1e3
                    r_["kernel"] = ("fw_k_update_fifo<INST>" if r_["update_path"] == "fifo" else
                                    "fw_k_update_range<INST>" if r_["update_path"] == "range" else "fw_k_update_stream<INST>")
                    r_["record_bytes_per_particle"] = 64
                    r_["attach"] = "fw_spawner_attach_instances_window" if window_ else "fw_spawner_attach_instances"
                recs[key] = r_
                del bufs
        extras["with_instance_records"] = recs
        # (c) configs[4] on this one GPU: the base point of the multi-GPU curve
        with ParticleSystem(device=local_rank, seed=workloads.SEED, stream=stream.cuda_stream) as p4:
            for e, (s_, tf_) in enumerate(workloads.many_emitters(args.emitters, args.live_per_emitter)):
                p4.spawn(s_, tf_, uid=e)
            p4.update(dt)
            for _ in range(76 + 10):
                p4.step(dt)
            torch.cuda.synchronize()
            b0 = p4.updated_total()
            t1 = time.perf_counter()
            for _ in range(40):
                p4.step(dt)
            p4.synchronize()
            el = time.perf_counter() - t1
            extras["configs4_one_gpu"] = {"particles_per_s": (p4.updated_total() - b0) / el, "ms_per_step": el / 40 * 1e3,
                                          "live_particles": p4.live_count(),
                                          "workload": f"{args.emitters} emitters x {args.live_per_emitter} live on one GPU"}
    if ps is not None:
        ps.close()

    # the CPU baseline: rank 0, outside the timed region, the other ranks at the barrier below
    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(args, dt, world, workload=workload)
    if rank == 0:
        out = assemble_line(args, world, workload, rows, roof, extras, cpu, hist, rccl_ranks, measured_copy,
                            args.reduce_every if dist is not None else None)
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
