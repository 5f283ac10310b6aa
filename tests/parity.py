"""Shared helpers for the parity tests: tolerances, the HIP backend and the oracle side by side, and the committed
multi-frame golden trajectories (tests/golden/trajectories.npz, produced by the independent numpy restatement)."""
import os

import numpy as np

from bevy_firework_amd import settings as S

EXACT_FIELDS = ("age", "lifetime", "initial_scale", "scale", "base_color", "emissive_color", "pbr")
TRIG_FIELDS = ("position", "velocity", "rotation", "angular_velocity")
# BASELINE.json north_star: "positions/velocities/colors within 1e-5 relative fp32".  Colours, scale, age and lifetime
# involve no libm call and are compared bit for bit.  The four vector fields depend on sin/cos (spawn cones, shapes,
# the per-frame quaternion step), where the three implementations use three libms, and are compared PER ELEMENT:
#     |got - want| <= RTOL * max(|want element|, |want vector|_2) + ATOL
# - the vector norm enters because a rotation error moves a component by a fraction of the vector's LENGTH, not of that
#   component (a velocity pointing almost along +y has an x component whose error comes from the y magnitude);
# - ATOL is an absolute floor of one fp32 ulp at magnitude 4..8 (ulp(4) = 4.8e-7): a position is a sum
#   `origin + offset` and later `position + velocity * dt` (core.rs:454, 626), so a component that cancels to nearly
#   zero still carries the rounding of its O(1..10) operands.  Round 6 measured how much of it is USED
#   (profiles/r06/parity_error_budget.txt, tools/r06_error_budget.py): at configs[0]..[4] sizes NO element needs the floor and the
#   worst error is 0.22 of the allowance; over the whole GPU suite seven scenarios do (fuzz cases 0 / 22 / 36 / 37 and the
#   spinning particles of test_irregular_dt_zero_steps_and_spinning_particles: 22 of 1354 tests fail with ATOL = 0), all of them
#   pass with 2.5e-7 (profiles/r06/parity_atol.txt).  5e-7 is a quarter of the 2e-6 of rounds 2-5 and 1/3000 of round 1's floor.
RTOL = 1e-5
ATOL = float(os.environ.get("FW_TEST_ATOL", "5e-7"))  # (the override exists for tools/r06_error_budget.py-style experiments only)


def planes_left_to_readers() -> int:
    """1 when the update of EVERY type leaves scale and colours to its readers (FW_TYPE_DERIVED for all: the product's default
    from round 6 on), 0 under FW_DERIVED=0 / 1 (planes stored unless an instance buffer is attached: rounds 3-5).  The
    byte-accounting assertions of the suite are written for both."""
    return 0 if os.environ.get("FW_ENABLE_KNOBS") == "1" and os.environ.get("FW_DERIVED", "2") in ("0", "1") else 1
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def trig_field_errors(got: np.ndarray, want: np.ndarray):
    """-> (boolean ok per element, worst error as a multiple of the allowance)"""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    norm = np.sqrt((want * want).sum(axis=-1, keepdims=True))
    allow = RTOL * np.maximum(np.abs(want), norm) + ATOL
    err = np.abs(got - want)
    ok = (err <= allow) | (got == want)  # equal infinities compare equal
    worst = float((err / allow).max()) if err.size else 0.0
    return ok, worst


def _where(got: np.ndarray, want: np.ndarray) -> str:
    """which particles of a field differ (for the assertion message)"""
    bad = got != want
    if bad.ndim > 1:
        bad = bad.reshape(len(bad), -1).any(axis=1)
    idx = np.flatnonzero(bad)
    if not len(idx):
        return "no element differs (NaN?)"
    i = int(idx[0])
    return (f"{len(idx)} of {len(bad)} particles differ, indices {i}..{int(idx[-1])}; first: got {got[i]!r} want {want[i]!r}"
            + (f"; got[{i + 1}] {got[i + 1]!r} want[{i + 1}] {want[i + 1]!r}" if i + 1 < len(bad) else ""))


def assert_particles_match(gpu: np.ndarray, cpu: np.ndarray, exact_all: bool = False, what: str = ""):
    assert len(gpu) == len(cpu), f"{what}: count {len(gpu)} != expected {len(cpu)}"
    for f in EXACT_FIELDS:
        if f in cpu.dtype.names:
            assert np.array_equal(gpu[f], cpu[f]), f"{what}: field {f} not bit-exact: {_where(gpu[f], cpu[f])}"
    for f in TRIG_FIELDS:
        if exact_all:
            assert np.array_equal(gpu[f], cpu[f]), f"{what}: field {f} not bit-exact: {_where(gpu[f], cpu[f])}"
        elif len(cpu):
            ok, worst = trig_field_errors(gpu[f], cpu[f])
            assert ok.all(), (f"{what}: field {f}: {np.count_nonzero(~ok)} elements outside rtol {RTOL} + atol {ATOL} "
                              f"(worst {worst:.2f}x the allowance)")


class Pair:
    """The same spawner on the HIP backend and on the oracle, stepped in lockstep."""

    def __init__(self, system, spawner: S.ParticleSpawner, transform=None, seed=0, uid=0, modifier=None):
        import oracle

        transform = transform or S.Transform()
        self.gpu = system.spawn(spawner, transform, uid=uid, modifier=modifier)
        self.cpu = oracle.OracleSpawner(spawner, seed=seed, uid=uid, transform=transform)
        if modifier is not None:
            self.cpu.set_modifier(modifier)
        self.spawner = spawner
        self.cpu_transform = transform
        self.n_types = len(spawner.particle_settings)

    def queue(self, n):
        self.gpu.queue_particles(n)
        self.cpu.queue_particles(n)

    def step_cpu(self, dt):
        self.cpu.step(np.float32(dt))

    def check(self, exact_all=False, what=""):
        assert self.gpu.counts() == self.cpu.counts(), f"{what}: counts {self.gpu.counts()} != {self.cpu.counts()}"
        for t in range(self.n_types):
            try:
                assert_particles_match(self.gpu.particles(t), self.cpu.particles(t), exact_all, f"{what} type {t}")
            except AssertionError as e:
                # read the same state again: a second read that matches says the first READ was wrong, not the state
                try:
                    assert_particles_match(self.gpu.particles(t), self.cpu.particles(t), exact_all, "")
                    again = "a SECOND read of the same state matches"
                except AssertionError as e2:
                    again = f"a second read differs too: {str(e2)[:160]}"
                g, c = self.gpu.particles(t), self.cpu.particles(t)
                per_field = {}
                if len(g) == len(c):
                    for f in c.dtype.names:
                        bad = g[f] != c[f]
                        if bad.ndim > 1:
                            bad = bad.reshape(len(bad), -1).any(axis=1)
                        per_field[f] = int(np.count_nonzero(bad))
                raise AssertionError(f"{e} [{again}; path {self.gpu.update_path(t)}; a third read, differing particles per field "
                                     f"(not bit-equal; trig fields differ legitimately): {per_field}]") from None


# ---- golden trajectories ---------------------------------------------------------------------------------------
_GOLDEN = None


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = np.load(os.path.join(GOLDEN_DIR, "trajectories.npz"))
    return _GOLDEN


def golden_particles(name: str, frame: int, t: int) -> np.ndarray:
    """the stored state of particle type `t` at checkpoint `frame`, as a PARTICLE_DTYPE-like record array (no pbr)"""
    g = golden()
    pre = f"{name}/f{frame}/t{t}/"
    n = len(g[pre + "age"])
    dt = np.dtype([(k, np.float32, s) if s else (k, np.float32) for k, s in
                   (("position", 3), ("velocity", 3), ("rotation", 4), ("angular_velocity", 3), ("initial_scale", 0),
                    ("scale", 0), ("age", 0), ("lifetime", 0), ("base_color", 4), ("emissive_color", 4))])
    out = np.zeros(n, dtype=dt)
    for k in dt.names:
        out[k] = g[pre + k]
    return out


def run_scenario(sc, make_target, check):
    """drive any implementation over a scenario of tests/golden/scenarios.py; `make_target()` returns an object with
    step(dt) taking np.float32; `check(frame)` is called at every checkpoint"""
    step = make_target()
    for fr in range(sc["frames"]):
        step(np.float32(sc["dts"][fr % len(sc["dts"])]))
        if fr in sc["checkpoints"]:
            check(fr)
