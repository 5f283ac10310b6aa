"""Synthetic workloads = BASELINE.json `configs`.  Settings only."""
from __future__ import annotations

import math
from typing import List, Tuple

from .settings import (
    Collider, EmissionMode, EmissionPacing, EmissionSettings, EmissionShape, FireworkCurve, FireworkGradient,
    ParticleCollisionSettings, ParticleSettings, ParticleSpawner, RandF32, RandVec3, Transform,
)

DT_60 = 1.0 / 60.0
SEED = 0x00C0FFEE

STRESS_GRADIENT = [  # examples/stress_test.rs:100-106
    (0.0, (10.0, 7.0, 1.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.8, (1.0, 0.3, 0.3, 1.0)),
    (0.9, (0.3, 0.3, 0.3, 1.0)), (1.0, (0.1, 0.1, 0.1, 0.0)),
]


def stress_test(rate: float = 160000.0) -> Tuple[ParticleSpawner, Transform]:
    """configs[0]: examples/stress_test.rs:92-129 (1 emitter, Circle, cone velocity, 5-key gradient)."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(1.0), initial_scale=RandF32(0.02, 0.08), scale_curve=FireworkCurve.constant(1.0),
        base_color=FireworkGradient.uneven_samples(STRESS_GRADIENT), linear_drag=0.1, pbr=False,
    )
    es = EmissionSettings(
        emission_pacing=EmissionPacing.rate(rate), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 0.3),
        inherit_parent_velocity=True,
        initial_velocity=RandVec3(RandF32(0.0, 10.0), (0.0, 1.0, 0.0), 30.0 / 180.0 * math.pi),
    )
    return ParticleSpawner([ps], [es]), Transform((0.0, 0.1, 0.0))


def one_million(rate: float = 1.0e6, lifetime: float = 1.0, capacity: int = 0) -> Tuple[ParticleSpawner, Transform]:
    """configs[1]: 1 emitter, ~1M live (rate 1e6/s x 1 s -> 983 333 live at dt = 1/60: one frame of emission is
    lost per cycle wrap), Point emission, linear 2-key scale / colour curves."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(lifetime), initial_scale=RandF32(0.02, 0.08),
        scale_curve=FireworkCurve.even_samples([1.0, 2.0]),
        base_color=FireworkGradient.even_samples([(1.0, 1.0, 1.0, 1.0), (0.0, 0.0, 0.0, 0.0)]),
        emissive_color=FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]),
        capacity=capacity,
    )
    es = EmissionSettings(
        emission_pacing=EmissionPacing.rate(rate), emission_shape=EmissionShape.Point(),
        initial_velocity=RandVec3(RandF32(0.0, 10.0), (0.0, 1.0, 0.0), 0.5),
        initial_angular_velocity=RandVec3(RandF32(0.0, 5.0), (0.0, 1.0, 0.0), 0.0),
    )
    return ParticleSpawner([ps], [es]), Transform((0.0, 0.1, 0.0))


def many_emitters(n_emitters: int = 256, live_per_emitter: int = 65536) -> List[Tuple[ParticleSpawner, Transform]]:
    """configs[2] / configs[4]: Sphere emission + radial velocity, lifetimes in [0.8, 1.2], distinct per-emitter
    constants (acceleration, drag, gradient) so the constants really come from the per-type tables."""
    out = []
    side = max(1, int(math.ceil(math.sqrt(n_emitters))))
    for e in range(n_emitters):
        k = e % 7
        grad = FireworkGradient.uneven_samples([
            (0.0, (4.0 + k, 2.0, 0.5 * k, 1.0)), (0.5 + 0.05 * k, (1.0, 0.5 + 0.1 * k, 0.2, 1.0)),
            (1.0, (0.1, 0.1, 0.1, 0.0)),
        ])
        ps = ParticleSettings(
            lifetime=RandF32(0.8, 1.2), initial_scale=RandF32(0.02, 0.06),
            scale_curve=FireworkCurve.even_samples([1.0, 1.5 + 0.1 * k, 0.2]),
            acceleration=(0.1 * k, -9.81 + 0.5 * k, -0.05 * k), linear_drag=0.1 + 0.02 * k, base_color=grad,
            emissive_color=FireworkGradient.even_samples([(2.0, 1.0 + 0.1 * k, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)]),
        )
        es = EmissionSettings(
            emission_pacing=EmissionPacing.rate(float(live_per_emitter)),  # mean lifetime 1.0 s
            emission_shape=EmissionShape.Sphere(1.0), initial_velocity=RandVec3.constant((0.0, 0.0, 0.0)),
            initial_velocity_radial=RandF32(1.0, 4.0),
        )
        out.append((ParticleSpawner([ps], [es]), Transform((3.0 * (e % side), 0.0, 3.0 * (e // side)))))
    return out


def nested(spark_rate: float = 100000.0, smoke_per_spark: float = 20.0) -> Tuple[ParticleSpawner, Transform]:
    """configs[3]: sparks (Global) -> smoke (Nested on sparks), modelled on examples/textures.rs:124-163."""
    sparks = ParticleSettings(
        lifetime=RandF32.constant(2.0), initial_scale=RandF32(0.01, 0.03), linear_drag=0.3,
        base_color=FireworkGradient.even_samples([(8.0, 4.0, 1.0, 1.0), (1.0, 0.2, 0.0, 0.0)]),
    )
    smoke = ParticleSettings(
        lifetime=RandF32.constant(2.0), initial_scale=RandF32(0.05, 0.1), acceleration=(0.0, 0.5, 0.0),
        linear_drag=0.7, scale_curve=FireworkCurve.even_samples([1.0, 3.0]),
        base_color=FireworkGradient.uneven_samples([(0.0, (0.1, 0.1, 0.1, 0.0)), (0.1, (0.1, 0.1, 0.1, 0.15)),
                                                    (1.0, (0.1, 0.1, 0.1, 0.0))]),
        pbr=True,
    )
    e_sparks = EmissionSettings(
        particle_index=0, emission_pacing=EmissionPacing.rate(spark_rate),
        initial_velocity=RandVec3(RandF32(2.0, 5.0), (0.0, 1.0, 0.0), 0.4),
        initial_angular_velocity=RandVec3(RandF32(5.0, 15.0), (0.0, -1.0, 0.0), 0.0),
    )
    e_smoke = EmissionSettings(
        particle_index=1, emission_mode=EmissionMode.Nested(0),
        emission_pacing=EmissionPacing.CountOverDuration(smoke_per_spark, 0.0, 0.0, 0.5),
        inherit_parent_velocity=False,
    )
    return ParticleSpawner([sparks, smoke], [e_sparks, e_smoke]), Transform((-2.0, 2.0, 0.0))


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


STRESS_COLLISION_GRADIENT = [  # examples/stress_test_collision.rs:101-107
    (0.0, (100.0, 70.0, 10.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.8, (1.0, 0.3, 0.3, 1.0)),
    (0.9, (0.3, 0.3, 0.3, 1.0)), (1.0, (0.1, 0.1, 0.1, 0.0)),
]


def stress_test_collision(rate: float = 80000.0):
    """examples/stress_test_collision.rs:68-151: one emitter (Circle, cone velocity 6-8, lifetime 2 s, linear_drag 0.15) whose
    particles bounce (restitution 0.6, friction 0.2, destroy_on_collision false) off a ground slab and an angled cube.
    Returns (spawner, transform, colliders): the two `Collider::cuboid`s of the example as this backend's analytic boxes
    (cuboid(x, y, z) takes full extents: half extents here)."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(2.0), initial_scale=RandF32(0.02, 0.08), scale_curve=FireworkCurve.constant(1.0),
        linear_drag=0.15, base_color=FireworkGradient.uneven_samples(STRESS_COLLISION_GRADIENT), pbr=False,
        collision_settings=ParticleCollisionSettings(restitution=0.6, friction=0.2, destroy_on_collision=False),
    )
    es = EmissionSettings(
        emission_pacing=EmissionPacing.rate(rate), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 0.3),
        initial_velocity=RandVec3(RandF32(6.0, 8.0), (0.0, 1.0, 0.0), 30.0 / 180.0 * math.pi), inherit_parent_velocity=True,
    )
    h = math.pi / 8.0  # half of PI / 4
    tf = Transform((5.0, 0.5, 0.0), (0.0, 0.0, math.sin(h), math.cos(h)))  # Quat::from_rotation_z(PI / 4.)
    cube_rot = _quat_mul((math.sin(h), 0.0, 0.0, math.cos(h)), (0.0, math.sin(h), 0.0, math.cos(h)))  # rotation_x * rotation_y
    colliders = [
        Collider.Box((0.0, -0.5, 0.0), (4.0, 0.5, 4.0)),               # Collider::cuboid(8., 1., 8.) at (0, -0.5, 0)
        Collider.Box((0.0, 0.5, 0.0), (0.5, 0.5, 0.5), cube_rot),      # Collider::cuboid(1., 1., 1.), the angled cube
    ]
    return ParticleSpawner([ps], [es]), tf, colliders


# ---- the reference's remaining examples (examples/*.rs), settings only: what a user of the crate actually runs ---------------
SPARKS_GRADIENT = [  # examples/sparks.rs:58-64, examples/on_demand.rs:62-68
    (0.0, (150.0, 100.0, 15.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.8, (1.0, 0.3, 0.3, 1.0)),
    (0.9, (0.3, 0.3, 0.3, 1.0)), (1.0, (0.1, 0.1, 0.1, 0.0)),
]
SMOKE_GRADIENT = [(0.0, (0.6, 0.3, 0.0, 0.0)), (0.1, (0.6, 0.3, 0.0, 0.35)), (1.0, (0.6, 0.3, 0.0, 0.0))]  # pbr.rs:58-62, one_shot.rs:101-105


def _quat_from_arc(a, b):
    """glam Quat::from_rotation_arc for two unit vectors that are not (anti)parallel: (a x b, 1 + a.b), normalised"""
    cx, cy, cz = a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]
    w = 1.0 + a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
    n = math.sqrt(cx * cx + cy * cy + cz * cz + w * w)
    return (cx / n, cy / n, cz / n, w / n)


def example_sparks(pacing: EmissionPacing = None, lifetime: float = 0.75) -> Tuple[ParticleSpawner, Transform]:
    """examples/sparks.rs:48-85 (rate 1000/s, lifetime 0.75 s); examples/on_demand.rs:56-96 is the same spawner with
    EmissionPacing::OnDemand (`pacing`), fed one particle per mouse click (on_demand.rs:130-141)."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(lifetime), initial_scale=RandF32(0.02, 0.08), scale_curve=FireworkCurve.constant(1.0),
        base_color=FireworkGradient.uneven_samples(SPARKS_GRADIENT), linear_drag=0.1, pbr=False,
    )
    es = EmissionSettings(
        emission_pacing=pacing or EmissionPacing.rate(1000.0), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 0.3),
        inherit_parent_velocity=True,
        initial_velocity=RandVec3(RandF32(0.0, 10.0), (0.0, 1.0, 0.0), 30.0 / 180.0 * math.pi),
    )
    return ParticleSpawner([ps], [es]), Transform((0.0, 0.1, 0.0))


def example_on_demand() -> Tuple[ParticleSpawner, Transform]:
    return example_sparks(EmissionPacing.OnDemand())


def example_pbr() -> Tuple[ParticleSpawner, Transform]:
    """examples/pbr.rs:49-84: slow smoke -- rate 150/s, lifetime 5 s, a wide Circle, NO initial velocity, a small upward
    acceleration against drag 0.7, a 2-key scale curve and a 3-key alpha gradient."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(5.0), scale_curve=FireworkCurve.even_samples([1.0, 2.0]), initial_scale=RandF32(0.5, 1.3),
        acceleration=(0.0, 0.3, 0.0), linear_drag=0.7, base_color=FireworkGradient.uneven_samples(SMOKE_GRADIENT),
        emissive_color=FireworkGradient.constant((0.0, 0.0, 0.0, 1.0)), fade_scene=3.5, pbr=True,
    )
    es = EmissionSettings(
        particle_index=0, emission_pacing=EmissionPacing.rate(150.0), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 3.5),
        initial_velocity=RandVec3.constant((0.0, 0.0, 0.0)), initial_velocity_radial=RandF32.constant(0.0),
        inherit_parent_velocity=True,
    )
    return ParticleSpawner([ps], [es]), Transform((0.0, 0.1, 0.0))


def example_collision():
    """examples/collision.rs:43-112: rate 100/s, lifetime 6.75 s, a 3-key uneven scale curve, a constant base colour and a 4-key
    emissive gradient, bouncing (restitution 0.6, friction 0.2) off the ground slab and the angled cube -- the scene
    stress_test_collision scales up.  Returns (spawner, transform, colliders)."""
    ps = ParticleSettings(
        lifetime=RandF32.constant(6.75), scale_curve=FireworkCurve.uneven_samples([(0.0, 1.0), (0.8, 1.0), (1.0, 0.0)]),
        initial_scale=RandF32(0.02, 0.08), linear_drag=0.15, base_color=FireworkGradient.constant((0.1, 0.1, 0.1, 1.0)),
        emissive_color=FireworkGradient.uneven_samples([
            (0.0, (30.0, 21.0, 1.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.75, (1.0, 0.3, 0.3, 1.0)), (0.8, (0.0, 0.0, 0.0, 1.0))]),
        pbr=True, collision_settings=ParticleCollisionSettings(restitution=0.6, friction=0.2, destroy_on_collision=False),
    )
    es = EmissionSettings(
        particle_index=0, emission_pacing=EmissionPacing.rate(100.0), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 0.3),
        initial_velocity=RandVec3(RandF32(6.0, 8.0), (0.0, 1.0, 0.0), 30.0 / 180.0 * math.pi), inherit_parent_velocity=True,
    )
    _, tf, colliders = stress_test_collision()  # (the same emitter pose, slab and cube: collision.rs:45-49, 95-112)
    return ParticleSpawner([ps], [es]), tf, colliders


def example_one_shot(impulse: float = 4.0, normal=(0.0, 1.0, 0.0), translation=(0.0, -2.0, 0.0)) -> Tuple[ParticleSpawner, Transform]:
    """examples/one_shot.rs:91-136: the dust puff a bouncing ball leaves at a contact -- OneShot(20) in
    SpawnTransformMode::Local, the scale range derived from the contact impulse, the emitter rotated onto the contact
    normal; the spawner entity is despawned on ParticleSpawnerFinished (one_shot.rs:138-142)."""
    from .settings import SpawnTransformMode

    ps = ParticleSettings(
        lifetime=RandF32.constant(2.5), initial_scale=RandF32(max(impulse / 10.0 - 0.1, 0.0), min(impulse / 10.0 + 0.1, 1.0)),
        scale_curve=FireworkCurve.even_samples([1.0, 2.0]), base_color=FireworkGradient.uneven_samples(SMOKE_GRADIENT),
        linear_drag=0.7, pbr=True, acceleration=(0.0, -1.5, 0.0), fade_scene=3.5,
    )
    es = EmissionSettings(
        emission_pacing=EmissionPacing.OneShot(20), emission_shape=EmissionShape.Circle((0.0, 1.0, 0.0), 0.4),
        inherit_parent_velocity=True, initial_velocity=RandVec3(RandF32(0.0, 2.0), (0.0, 1.0, 0.0), 0.0),
        initial_velocity_radial=RandF32(0.0, 2.5),
    )
    n = math.sqrt(sum(c * c for c in normal))
    normal = tuple(c / n for c in normal)
    rot = (0.0, 0.0, 0.0, 1.0) if normal == (0.0, 1.0, 0.0) else _quat_from_arc((0.0, 1.0, 0.0), normal)
    return ParticleSpawner([ps], [es], spawn_transform_mode=SpawnTransformMode.Local), Transform(tuple(translation), rot)


def example_textures(with_world: bool = True):
    """examples/textures.rs:53-173: bullet cases (rate 12/s, lifetime 5 s, initial rotation + a spin that angular_drag 0.85
    slows, bouncing with restitution 0.4 / friction 0.35) that each leave six smoke puffs in the first tenth of their life
    (a Nested CountOverDuration entry; its `duration` 0 is never read: core.rs:474-479), SpawnTransformMode::Local, the
    emitter turned from +Y onto +X.  Returns (spawner, transform, colliders): the example's own world -- the circular base, an avian
    Collider::cylinder(4., 0.2) at the origin (textures.rs:191-196), and a Collider::cone(0.5, 1.) at (0, 0.5, 0)
    (textures.rs:198-212) -- as this backend's analytic cylinder and cone (round 5; stand-ins until then)."""
    from .settings import SpawnTransformMode

    s = math.sin(math.pi / 4.0)
    cases = ParticleSettings(
        lifetime=RandF32.constant(5.0), scale_curve=FireworkCurve.constant(1.0), initial_scale=RandF32.constant(0.3),
        linear_drag=0.3, angular_drag=0.85,
        base_color=FireworkGradient.uneven_samples([(0.0, (1.0, 1.0, 1.0, 1.0)), (0.9, (1.0, 1.0, 1.0, 1.0)), (1.0, (1.0, 1.0, 1.0, 0.0))]),
        emissive_color=FireworkGradient.constant((0.0, 0.0, 0.0, 1.0)), fade_scene=0.0, fade_edge=0.0, pbr=True,
        collision_settings=ParticleCollisionSettings(restitution=0.4, friction=0.35, destroy_on_collision=False),
    )
    smoke = ParticleSettings(
        lifetime=RandF32.constant(2.0), scale_curve=FireworkCurve.even_samples([1.0, 2.0]), initial_scale=RandF32(0.5, 0.8),
        acceleration=(0.0, 0.3, 0.0), linear_drag=0.7,
        base_color=FireworkGradient.uneven_samples([(0.0, (0.1, 0.1, 0.1, 0.0)), (0.1, (0.1, 0.1, 0.1, 0.15)), (1.0, (0.1, 0.1, 0.1, 0.0))]),
        emissive_color=FireworkGradient.constant((0.0, 0.0, 0.0, 1.0)), fade_scene=3.5, pbr=True,
    )
    e_cases = EmissionSettings(
        particle_index=0, emission_mode=EmissionMode.Global(), emission_pacing=EmissionPacing.rate(12.0),
        emission_shape=EmissionShape.Point(), initial_velocity=RandVec3(RandF32(2.0, 5.0), (0.0, 1.0, 0.0), 0.4),
        initial_velocity_radial=RandF32.constant(0.0), inherit_parent_velocity=True,
        initial_rotation=(0.0, s, 0.0, s),  # Quat::from_rotation_y(FRAC_PI_2)
        initial_angular_velocity=RandVec3(RandF32(5.0, 15.0), (0.0, -1.0, 0.0), 0.0),
    )
    e_smoke = EmissionSettings(
        particle_index=1, emission_mode=EmissionMode.Nested(0),
        emission_pacing=EmissionPacing.CountOverDuration(6.0, 0.0, 0.0, 0.1), emission_shape=EmissionShape.Point(),
        initial_velocity=RandVec3.constant((0.0, 0.0, 0.0)), initial_velocity_radial=RandF32.constant(0.0),
        inherit_parent_velocity=False, initial_angular_velocity=RandVec3.constant((0.0, 0.0, 0.0)),
    )
    tf = Transform((-2.0, 2.0, 0.0), _quat_from_arc((0.0, 1.0, 0.0), (1.0, 0.0, 0.0)))
    colliders = [Collider.Cylinder((0.0, 0.0, 0.0), 4.0, 0.2), Collider.Cone((0.0, 0.5, 0.0), 0.5, 1.0)] if with_world else []
    return ParticleSpawner([cases, smoke], [e_cases, e_smoke], spawn_transform_mode=SpawnTransformMode.Local), tf, colliders
