"""Lifecycle fuzz at PRODUCT DEFAULTS (no knob set): which update path a particle type takes is decided when its spawner is built,
from what the context holds at that moment -- few segments or many (fw_ctx::range_few, with hysteresis), a FIFO ring or none, more
one-lifetime types than one FIFO launch holds (fw_ctx::n_spilled), a Nested child type whose derived capacity passes 262 144
slots -- and spawners that come and go move types from one path to another WITH THEIR PARTICLES (drop_few_rings, fifo_to_range,
fifo_to_general).  The three-path fuzz of tests/test_gpu_fuzz.py forces the thresholds to 0 and never runs any of this.  Here:
random create / destroy / update_settings of spawners of random sizes, so that a context crosses 64 segments in both directions,
gains and loses its first FIFO ring, receives a ninth large one-lifetime type, with the state of the spawners compared to the
oracle after EVERY transition and every few frames.  Needs an MI355X."""
import os

import numpy as np
import pytest

from bevy_firework_amd import settings as S
from parity import Pair
from test_gpu_fuzz import _curve, _gradient, _randvec, _unit

pytestmark = pytest.mark.gpu
SEED = 0x11FE
CASES = int(os.environ.get("FW_LIFECYCLE_CASES", "200"))
OFF = int(os.environ.get("FW_LIFECYCLE_OFFSET", "0"))
RANGE_FEW, SMALL_MIN = 192, 352  # fw_ctx::range_few / small_min (csrc/fw_engine.h)
KNOBS = ("FW_ENABLE_KNOBS", "FW_SMALL", "FW_SMALL_MAX", "FW_SMALL_MIN", "FW_HOST_FAST", "FW_PARAM_BAR", "FW_FIFO", "FW_FIFO_MIN", "FW_FIFO_SMALL", "FW_RANGE", "FW_RANGE_MIN", "FW_RANGE_SMALL", "FW_RANGE_FEW", "FW_NOSPIN",
         "FW_NEST_FUSE", "FW_NT_MB", "FW_NT_WO_MB", "FW_FORECAST", "FW_STREAM", "FW_STATIC_NEW", "FW_UPDATE_MODE", "FW_FIFO_STREAM")


def _type(rng, lo, hi, one_value, spin):
    life = S.RandF32.constant(float(lo)) if one_value else S.RandF32(float(lo), float(hi))
    return S.ParticleSettings(
        lifetime=life, scale_curve=_curve(rng), initial_scale=S.RandF32(0.01, float(rng.uniform(0.02, 0.2))),
        acceleration=tuple(float(c) for c in rng.uniform(-8.0, 8.0, size=3)),
        angular_acceleration=tuple(float(c) for c in rng.uniform(-2.0, 2.0, size=3)) if spin and rng.random() < 0.5 else (0.0, 0.0, 0.0),
        linear_drag=float(rng.uniform(0.0, 1.0)), angular_drag=float(rng.uniform(0.0, 1.0)), base_color=_gradient(rng),
        emissive_color=_gradient(rng), pbr=bool(rng.random() < 0.5))


def _entry(rng, t, rate, spin):
    shape = [S.EmissionShape.Point(), S.EmissionShape.Sphere(float(rng.uniform(0.1, 1.5))),
             S.EmissionShape.Circle(_unit(rng), float(rng.uniform(0.1, 1.5)))][int(rng.integers(0, 3))]
    return S.EmissionSettings(
        particle_index=t, emission_pacing=S.EmissionPacing.rate(float(rate)), emission_shape=shape, initial_velocity=_randvec(rng, 5.0),
        initial_velocity_radial=S.RandF32(0.0, float(rng.uniform(0.0, 2.0))), inherit_parent_velocity=bool(rng.random() < 0.5),
        initial_angular_velocity=_randvec(rng, 6.0) if spin else S.RandVec3.constant((0.0, 0.0, 0.0)))


def make(rng, kind):
    """one spawner of a given size class -> (spawner, the paths its types take in a context that applies the usual thresholds)"""
    spin = bool(rng.random() < 0.4)
    if kind == "tiny":  # a few hundred particles: a small range ring in a context of few segments, the compacting path among many
        lo = float(rng.uniform(0.12, 0.4))
        return S.ParticleSpawner([_type(rng, lo, lo + rng.uniform(0.05, 0.3), rng.random() < 0.4, spin)], [_entry(rng, 0, rng.uniform(300.0, 1500.0), spin)])
    if kind == "dust":  # ~50-250 particles: always eligible for the wave-per-type kernel (which runs from SMALL_MIN such types on)
        lo = float(rng.uniform(0.12, 0.3))
        return S.ParticleSpawner([_type(rng, lo, lo + 0.1, rng.random() < 0.4, spin)], [_entry(rng, 0, rng.uniform(200.0, 600.0), spin)])
    if kind == "mid":  # a derived capacity past 8192 slots: a range ring at any time (lifetime range), ~4-6k particles
        lo = float(rng.uniform(0.2, 0.3))
        return S.ParticleSpawner([_type(rng, lo, lo + 0.2, False, spin)], [_entry(rng, 0, rng.uniform(14000.0, 20000.0), spin)])
    if kind == "fifo":  # one lifetime value, derived capacity past 32768 slots: a FIFO ring, ~18-24k particles
        lo = float(rng.uniform(0.28, 0.36))
        return S.ParticleSpawner([_type(rng, lo, lo, True, spin)], [_entry(rng, 0, rng.uniform(62000.0, 70000.0), spin)])
    if kind == "two":  # two types, three entries: a tiny one and a mid one in one spawner
        lo = float(rng.uniform(0.15, 0.3))
        return S.ParticleSpawner([_type(rng, lo, lo + 0.2, rng.random() < 0.5, spin), _type(rng, 0.2, 0.45, False, False)],
                                 [_entry(rng, 0, 900.0, spin), _entry(rng, 1, 15000.0, False), _entry(rng, 0, 500.0, spin)])
    big = kind == "nested_big"  # sparks -> smoke; big: the child type's derived capacity passes 262 144 slots (a FIFO ring)
    plife = 0.4 if big else float(rng.uniform(0.25, 0.4))
    sparks = _type(rng, plife, plife + 0.1, big or rng.random() < 0.5, spin)
    smoke = _type(rng, 0.15, 0.3, big or rng.random() < 0.5, False)
    # (big: the sparks' derived capacity passes 32 768 slots -- a FIFO ring -- and the smoke's, parents' capacity x 8 x 1.25, 262 144:
    # both in FIFO rings, the Nested entry inside their launch; ~18k sparks, ~55k puffs)
    e0 = _entry(rng, 0, 46000.0 if big else rng.uniform(400.0, 1200.0), spin)
    e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                            emission_pacing=S.EmissionPacing.CountOverDuration(8.0 if big else float(rng.uniform(2.0, 6.0)), 1.0, 0.0, float(rng.uniform(0.5, 1.0))),
                            inherit_parent_velocity=bool(rng.random() < 0.5), initial_velocity=_randvec(rng, 2.0))
    return S.ParticleSpawner([sparks, smoke], [e0, e1])


class World:
    def __init__(self, system, rng, case):
        self.system, self.rng, self.case = system, rng, case
        self.pairs, self.kinds, self.uid = [], [], 100000 * (case + 1)
        self.frames = 0
        self.seen = set()  # update paths seen (for the bookkeeping test at the end of the file)

    def add(self, kind):
        tf = S.Transform(tuple(float(c) for c in self.rng.uniform(-3.0, 3.0, size=3)))
        self.pairs.append(Pair(self.system, make(self.rng, kind), tf, seed=SEED, uid=self.uid))
        self.kinds.append(kind)
        self.uid += 1

    def remove(self, k):
        self.system.despawn(self.pairs[k].gpu)
        self.pairs[k].cpu.close()
        del self.pairs[k], self.kinds[k]

    def rebuild(self, k):  # Changed<ParticleSpawner> with the same settings: emission state reset, particles dropped (core.rs:343-365)
        self.pairs[k].gpu.update_settings(self.pairs[k].spawner)
        self.pairs[k].cpu.reset()

    def segments(self):
        return sum(p.n_types for p in self.pairs)

    def paths(self):
        return [tuple(p.gpu.update_path(t)[0] for t in range(p.n_types)) for p in self.pairs]

    def step(self, n, base=1.0 / 60.0):
        for _ in range(n):
            r = self.rng.random()
            dt = np.float32(base if r < 0.75 else (0.0 if r < 0.8 else self.rng.uniform(0.003, 0.03)))
            self.system.update(dt)
            for p in self.pairs:
                p.step_cpu(dt)
            self.frames += 1

    def check(self, what, limit=10):
        """the state of (up to `limit` randomly chosen) spawners against the oracle -- the large ones always"""
        for row in self.paths():
            self.seen.update(row)
        order = [k for k, kind in enumerate(self.kinds) if kind in ("fifo", "nested_big", "mid", "two")]
        rest = [k for k in range(len(self.pairs)) if k not in order]
        self.rng.shuffle(rest)
        for k in (order + rest)[:limit]:
            self.pairs[k].check(what=f"case {self.case} frame {self.frames} ({what}) spawner {k} [{self.kinds[k]}] paths {self.paths()[k]}")


def scenario_many(w):
    """few -> many -> few: the context crosses fw_ctx::range_few (192 segments) upwards -- every small range ring continues on the
    compacting path -- and comes back below HALF of it, where new small types take rings again"""
    for _ in range(int(w.rng.integers(1, 5))):
        w.add(str(w.rng.choice(["tiny", "tiny", "mid", "two"])))
    w.step(int(w.rng.integers(8, 30)))
    w.check("few")
    assert all(p in ("range", "fifo") for row in w.paths() for p in row), w.paths()  # (few segments: everybody on a ring)
    while w.segments() <= RANGE_FEW:
        w.add("tiny")
        if w.rng.random() < 0.1:
            w.step(1)
    # (off their rings: workgroups of the compacting kernels -- the wave-per-type kernel takes over from SMALL_MIN eligible types on)
    assert all(p == "general" for row, kind in zip(w.paths(), w.kinds) for p in row if kind == "tiny"), w.paths()
    w.check("right after the context outgrew fw_ctx::range_few")
    w.step(int(w.rng.integers(5, 25)))
    w.check("many")
    while w.segments() > int(w.rng.integers(8, 30)):
        w.remove(int(w.rng.integers(0, len(w.pairs))))
        if w.rng.random() < 0.1:
            w.step(1)
    w.add("tiny")
    assert w.paths()[-1][0] == "range", (w.segments(), w.paths()[-1])  # back below half the limit: a small ring again
    if w.rng.random() < 0.5:
        w.rebuild(int(w.rng.integers(0, len(w.pairs))))
    w.step(int(w.rng.integers(10, 30)))
    w.check("few again")


def scenario_small_mode(w):
    """few -> hundreds -> fewer: small types leave their rings past fw_ctx::range_few segments (workgroups of the compacting kernels),
    move to the wave-per-type kernel TOGETHER when SMALL_MIN of them exist (fw_ctx::small_on: a flag per segment, particles stay where
    they are), and back below three quarters of that"""
    for _ in range(int(w.rng.integers(1, 4))):
        w.add("tiny")
    if w.rng.random() < 0.5:  # (a spawner with a Nested entry: its frames run the separate spawn / nest passes -- the small types of such a
        w.add("nested_small")  # frame find their new particles materialised instead of spawning them themselves)
    w.step(int(w.rng.integers(5, 15)))
    w.check("few")
    while w.segments() <= RANGE_FEW:
        w.add("dust")
    assert not any(row == ("small",) for row, kind in zip(w.paths(), w.kinds) if kind == "dust"), w.paths()
    w.step(int(w.rng.integers(3, 10)))
    w.check("on the compacting kernels", limit=8)
    while sum(kind == "dust" for kind in w.kinds) < SMALL_MIN + int(w.rng.integers(0, 12)):
        w.add("dust")
        if w.rng.random() < 0.03:
            w.step(1)
    assert all(row == ("small",) for row, kind in zip(w.paths(), w.kinds) if kind == "dust"), [r for r in w.paths() if r != ("small",)]
    w.check("right after the wave-per-type kernel took over", limit=14)
    w.step(int(w.rng.integers(8, 20)))
    w.check("one wave per type", limit=12)
    while w.segments() >= SMALL_MIN * 3 // 4 - int(w.rng.integers(0, 10)):
        w.remove(int(w.rng.integers(0, len(w.pairs))))
        if w.rng.random() < 0.03:
            w.step(1)
    # (the dust is back on workgroups of the compacting kernels; a "tiny" type that sustains more than a wave's worth -- a WIDE type --
    # keeps its workgroup of the small kernel down to a quarter of fw_ctx::wide_min eligible types)
    assert not any(row == ("small",) for row, kind in zip(w.paths(), w.kinds) if kind == "dust"), w.paths()
    w.check("right after the compacting kernels took over again", limit=14)
    w.step(int(w.rng.integers(8, 20)))
    w.check("a workgroup per type again", limit=12)


def scenario_fifo(w):
    """a context of small range rings gains its first FIFO ring (the small rings leave: a FIFO launch and a range launch would run
    one after the other), loses it, gets new small spawners, gains one again"""
    for _ in range(int(w.rng.integers(2, 7))):
        w.add(str(w.rng.choice(["tiny", "tiny", "two", "mid", "nested_small"])))
    w.step(int(w.rng.integers(10, 35)))
    w.check("small rings")
    w.add("fifo")
    assert w.paths()[-1] == ("fifo",)
    assert all(row != ("range",) for row, kind in zip(w.paths(), w.kinds) if kind == "tiny"), w.paths()
    w.check("right after the FIFO ring arrived")
    w.step(int(w.rng.integers(10, 30)))
    w.check("next to a FIFO ring")
    w.remove(len(w.pairs) - 1)
    for _ in range(int(w.rng.integers(1, 4))):
        w.add("tiny")
    w.step(int(w.rng.integers(5, 20)))
    w.check("the FIFO ring gone")
    w.add("fifo")
    if w.rng.random() < 0.5:
        w.rebuild(int(w.rng.integers(0, len(w.pairs))))
    w.step(int(w.rng.integers(10, 30)))
    w.check("a FIFO ring again")


def scenario_nested(w):
    """a small Nested spawner on range rings (derived child capacity below 262 144 slots) meets a large one (its child type takes a
    FIFO ring: its Nested entry runs inside the FIFO launch) -- and is left alone with it again"""
    w.add("nested_small")
    if w.rng.random() < 0.5:
        w.add("tiny")
    w.step(int(w.rng.integers(15, 40)))
    w.check("small Nested spawner")
    assert w.paths()[0] == ("range", "range"), w.paths()
    w.add("nested_big")
    assert w.paths()[-1] == ("fifo", "fifo"), w.paths()
    w.check("right after the large Nested spawner arrived")
    w.step(int(w.rng.integers(20, 45)))
    w.check("two Nested spawners", limit=4)
    fused, separate = w.system.nest_frames()
    assert fused == 0 and separate > 30, (fused, separate)  # (a context with TWO Nested spawners, one of them not in FIFO rings: the separate passes)
    w.remove(0)  # ... the small one leaves: the large spawner's entry runs inside its FIFO launch
    if w.kinds[0] == "tiny":
        w.remove(0)
    w.step(int(w.rng.integers(10, 30)))
    w.check("the large Nested spawner alone", limit=4)
    assert w.system.nest_frames()[0] >= 10, w.system.nest_frames()
    w.remove(len(w.pairs) - 1)
    w.add("nested_small")
    w.step(int(w.rng.integers(10, 30)))
    w.check("small ones again")


def scenario_spill(w):
    """more large one-lifetime types than one FIFO launch holds: the ninth takes a range ring and the eight FIFO rings follow it
    where they stand (fw_ctx::n_spilled); then enough of them leave"""
    n0 = int(w.rng.integers(6, 9))
    for _ in range(n0):
        w.add("fifo")
        if w.rng.random() < 0.3:
            w.step(int(w.rng.integers(1, 6)))
    w.step(int(w.rng.integers(15, 30)))
    w.check("FIFO rings", limit=3)
    while sum(kind == "fifo" for kind in w.kinds) < 9:
        w.add("fifo")
    assert all(row == ("range",) for row in w.paths()), w.paths()
    w.check("right after the ninth large one-lifetime type", limit=9)
    w.step(int(w.rng.integers(12, 30)))
    w.check("range rings", limit=4)
    for _ in range(int(w.rng.integers(2, 7))):
        w.remove(int(w.rng.integers(0, len(w.pairs))))
    w.add("fifo")
    assert w.paths()[-1] == ("range",)  # (converted rings exist: the newcomer joins them)
    w.step(int(w.rng.integers(8, 20)))
    w.check("fewer of them", limit=4)


SCENARIOS = [scenario_many] * 7 + [scenario_fifo] * 7 + [scenario_nested] * 3 + [scenario_spill] * 2 + [scenario_small_mode]
SEEN = {}


@pytest.mark.parametrize("case", range(OFF, OFF + CASES))
def test_lifecycle_at_product_defaults(case, monkeypatch):
    from bevy_firework_amd.system import ParticleSystem

    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(77000 + case)
    with ParticleSystem(device=0, seed=SEED) as system:
        w = World(system, rng, case)
        SCENARIOS[case % len(SCENARIOS)](w)
        SEEN[case] = (SCENARIOS[case % len(SCENARIOS)].__name__, sorted(w.seen), w.frames, sum(sum(p.gpu.counts()) for p in w.pairs))


def test_lifecycle_cases_were_not_trivial():
    if len(SEEN) < 40:
        pytest.skip("the lifecycle cases did not run in this session")
    names = {v[0] for v in SEEN.values()}
    assert names == {"scenario_many", "scenario_fifo", "scenario_nested", "scenario_spill", "scenario_small_mode"}, names
    assert all(set(v[1]) >= {"range", "general"} for v in SEEN.values() if v[0] == "scenario_many"), SEEN
    assert all(set(v[1]) >= {"range", "general", "small"} for v in SEEN.values() if v[0] == "scenario_small_mode"), SEEN
    assert all("fifo" in v[1] for v in SEEN.values() if v[0] not in ("scenario_many", "scenario_small_mode")), SEEN
    assert sum(v[3] for v in SEEN.values()) > 20000 * len(SEEN) // 10, SEEN
