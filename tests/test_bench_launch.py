"""bench.py --gpus N must give N ranks or fail: it never silently measures one GPU.

CPU-only checks of the launch plumbing.  FW_BENCH_LAUNCH_CHECK=1 makes the ranks stop after proving that they exist (a
gloo group + one all-reduce) -- the particle path itself has no CPU fallback and is not touched here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(FW_BENCH_LAUNCH_CHECK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE line
    out = lines[0]
    assert out["launch_check"] and out["self_launched"]
    assert out["rccl_ranks"] == 2 and out["ranks_seen"] == 2 and out["n_gpus"] == 2
    # ... and the line is the one a real N > 1 run prints, assembled by the same code from stand-in numbers: it carries the
    # whole-job figures, a roofline object with every rank's kernel in it and a CPU baseline (round 3's N > 1 line had
    # `cpu_baseline: null` and described rank 0 only)
    assert out["scaling"] == "strong" and out["config"]["workload"].startswith("configs[4]")
    assert out["value"] == (1000 + 2000) / 1.1 and out["ms_per_step"] == 1.1 / 3 * 1e3  # sum over ranks / the slowest rank
    assert out["config"]["per_rank_live"] == [100, 200] and out["config"]["live_particles"] == 300
    assert len(out["config"]["per_rank_ms_per_step"]) == 2
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "per_rank", "kernel_us_min", "kernel_us_max"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "infinity_cache") and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert [r["rank"] for r in roof["per_rank"]] == [0, 1] and all(r["frac"] > 0 for r in roof["per_rank"])
    assert roof["frac"] == min(r["frac"] for r in roof["per_rank"]) and roof["kernel_us_max"] == 51.0
    cpu = out["cpu_baseline"]
    assert cpu is not None and cpu["value"] > 0 and cpu["unit"] == "particles/s" and cpu["kind"] == "port" and cpu["cores"] >= 1


def test_gpus_n_with_too_few_devices_fails_loudly():
    # no GPU in this container: the direct call must refuse, not fall back to one rank / configs[1]
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to fall back" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no JSON line of a mis-measured run


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # ... and a launcher that gives ONE rank to --gpus 8 is refused as well (round 2's silent n_gpus: 1 line)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
