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
