"""Shared helpers for the parity tests: run the HIP backend and the oracle side by side."""
import numpy as np

import oracle
from bevy_firework_amd import settings as S

EXACT_FIELDS = ("age", "lifetime", "initial_scale", "scale", "base_color", "emissive_color", "pbr")
TRIG_FIELDS = ("position", "velocity", "rotation", "angular_velocity")
RTOL = 1e-5  # BASELINE.json north_star: positions/velocities/colours within 1e-5 relative fp32


def assert_particles_match(gpu: np.ndarray, cpu: np.ndarray, exact_all: bool = False, what: str = ""):
    assert len(gpu) == len(cpu), f"{what}: count {len(gpu)} != oracle {len(cpu)}"
    for f in EXACT_FIELDS:
        assert np.array_equal(gpu[f], cpu[f]), f"{what}: field {f} not bit-exact"
    for f in TRIG_FIELDS:
        if exact_all:
            assert np.array_equal(gpu[f], cpu[f]), f"{what}: field {f} not bit-exact"
        else:
            scale = max(1.0, float(np.max(np.abs(cpu[f]))) if len(cpu) else 1.0)
            ok = np.isclose(gpu[f], cpu[f], rtol=RTOL, atol=RTOL * scale)
            assert ok.all(), f"{what}: field {f}: {np.count_nonzero(~ok)} values outside {RTOL} rel"


class Pair:
    """The same spawner on the HIP backend and on the oracle, stepped in lockstep."""

    def __init__(self, system, spawner: S.ParticleSpawner, transform=None, seed=0, uid=0, modifier=None):
        transform = transform or S.Transform()
        self.gpu = system.spawn(spawner, transform, uid=uid, modifier=modifier)
        self.cpu = oracle.OracleSpawner(spawner, seed=seed, uid=uid, transform=transform)
        if modifier is not None:
            self.cpu.set_modifier(modifier)
        self.spawner = spawner
        self.n_types = len(spawner.particle_settings)

    def queue(self, n):
        self.gpu.queue_particles(n)
        self.cpu.queue_particles(n)

    def step_cpu(self, dt):
        self.cpu.step(np.float32(dt))

    def check(self, exact_all=False, what=""):
        assert self.gpu.counts() == self.cpu.counts(), f"{what}: counts {self.gpu.counts()} != {self.cpu.counts()}"
        for t in range(self.n_types):
            assert_particles_match(self.gpu.particles(t), self.cpu.particles(t), exact_all, f"{what} type {t}")
