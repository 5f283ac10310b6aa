import os, sys
os.environ["FW_FUZZ_EXTRA"] = "3000"
os.environ.setdefault("FW_ENABLE_KNOBS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fuzz as F
from bevy_firework_amd import settings as S
from bevy_firework_amd.system import ParticleSystem
from parity import Pair
case, seed_base, const_p = int(sys.argv[1]), 9000, 0.2
rng = np.random.default_rng(seed_base + case)
spawner = F._spawner(rng, scale=1.0 if case % 3 else 4.0, const_p=const_p)
for p in spawner.particle_settings:
    p.particles_destroyed = (lambda dead: None) if rng.random() < 0.6 else None
print("types", [(p.lifetime, p.scale_curve.kind, len(p.scale_curve.values), p.collision_settings is not None) for p in spawner.particle_settings])
print("entries", [(e.particle_index, e.emission_mode, e.emission_pacing.kind) for e in spawner.emission_settings])
with ParticleSystem(device=0, seed=F.SEED) as system:
    pair = Pair(system, spawner, S.Transform(), seed=F.SEED, uid=(seed_base // 30) + case)
    buf = None
    for i, dt in enumerate(F._steps(rng, 30)):
        dt = np.float32(dt)
        a = int(rng.integers(0, 8))
        act = ""
        if a == 0:
            tf = S.Transform(tuple(float(c) for c in rng.uniform(-3.0, 3.0, size=3)), tuple(float(c) for c in (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))))
            pair.gpu.set_transform(tf); pair.cpu.set_origin(tf.translation, tf.rotation); act = "transform"
        elif a == 1:
            v = tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3)); pair.gpu.set_parent_velocity(v); pair.cpu.set_parent_velocity(v); act = "pv"
        elif a == 2:
            m = S.EffectModifier(float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0))); pair.gpu.set_modifier(m); pair.cpu.set_modifier(m); act = f"modifier {m}"
        elif a == 3:
            n = int(rng.integers(0, 3000)); pair.queue(n); act = f"queue {n}"
        elif a == 4 and i > 4:
            t = int(rng.integers(0, pair.n_types)); parts = pair.cpu.particles(t)[:: int(rng.integers(1, 4))].copy()
            pair.gpu.write_particles(t, parts); pair.cpu.write_particles(t, parts); act = f"write type {t} n {len(parts)}"
        elif a == 5 and buf is None:
            buf = torch.full((60000 * 16,), float("nan"), dtype=torch.float32, device="cuda"); pair.gpu.attach_instances(buf.data_ptr(), 60000, particle_type=0); act = "attach"
        elif a == 6 and i > 8 and rng.random() < 0.4:
            pair.gpu.update_settings(spawner); pair.cpu.reset(); act = "reset"
            if buf is not None: pair.gpu.attach_instances(buf.data_ptr(), 60000, particle_type=0)
        system.update(dt); pair.step_cpu(dt)
        print(i, "dt %.4f" % dt, act, "counts", pair.gpu.counts(), "paths", [pair.gpu.update_path(t)[0] for t in range(pair.n_types)])
        for t in range(pair.n_types):
            if spawner.particle_settings[t].particles_destroyed is None: continue
            g, c = pair.gpu.destroyed(t), pair.cpu.destroyed(t)
            if len(g) != len(c): print("   destroyed count differs", t, len(g), len(c)); continue
            bad = np.flatnonzero(g["scale"] != c["scale"])
            if len(bad):
                j = bad[0]
                print("   type", t, "destroyed", len(g), "bad", len(bad), "first", j, "gpu", {k: g[k][j] for k in ("age", "lifetime", "initial_scale", "scale")}, "cpu", {k: c[k][j] for k in ("age", "lifetime", "initial_scale", "scale")})
                sys.exit(0)
