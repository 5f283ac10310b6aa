"""Inputs of the multi-frame golden trajectories (tests/golden/trajectories.npz).

Settings, seeds and dt sequences only -- shared by make_golden.py (which runs the numpy restatement
np_sim.py over them and stores the particle state at the checkpoint frames) and by the tests (which run
the C oracle and the HIP backend over the same inputs and compare with the stored state).
"""
from __future__ import annotations

import math

from bevy_firework_amd import settings as S

SEED = 0x5EED1234

STRESS_GRADIENT = [  # examples/stress_test.rs:100-106
    (0.0, (10.0, 7.0, 1.0, 1.0)), (0.7, (3.0, 1.0, 1.0, 1.0)), (0.8, (1.0, 0.3, 0.3, 1.0)),
    (0.9, (0.3, 0.3, 0.3, 1.0)), (1.0, (0.1, 0.1, 0.1, 0.0)),
]

_IRREGULAR = [1 / 60] * 6 + [1 / 144, 1 / 30, 0.0, 1 / 60, 0.021, 0.0043, 1 / 60, 1 / 60, 0.033, 1 / 90]


def rotation_cone_sphere():
    """quaternion spin (from_scaled_axis + Hamilton product), RandVec3 cones, Sphere shape, rotated origin,
    parent velocity, EffectModifier, uneven 5-key gradient, 3-key even scale curve, irregular dt"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.3, 2.0), initial_scale=S.RandF32(0.5, 1.5),
                            angular_acceleration=(0.1, 0.0, -0.2), angular_drag=0.3, linear_drag=0.15,
                            scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5]),
                            base_color=S.FireworkGradient.uneven_samples(STRESS_GRADIENT),
                            emissive_color=S.FireworkGradient.even_samples([(4.0, 2.0, 0.0, 1.0), (0.5, 0.0, 3.0, 1.0),
                                                                           (0.0, 0.0, 0.0, 1.0)]))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(800.0), emission_shape=S.EmissionShape.Sphere(0.5),
                            initial_rotation=(0.0, math.sin(0.4), 0.0, math.cos(0.4)),
                            initial_velocity=S.RandVec3(S.RandF32(0.5, 2.0), (0.6, 0.8, 0.0), 0.7),
                            initial_velocity_radial=S.RandF32(0.5, 1.5),
                            initial_angular_velocity=S.RandVec3(S.RandF32(1.0, 9.0), (0.0, 0.6, 0.8), 0.5))
    return dict(spawner=S.ParticleSpawner([ps], [es]),
                transform=S.Transform((0.0, 1.0, 0.0), (math.sin(0.3), 0.0, 0.0, math.cos(0.3))),
                modifier=S.EffectModifier(scale=2.0, speed=0.5), parent_velocity=(0.5, 0.0, -0.25), uid=5,
                dts=_IRREGULAR, frames=200, checkpoints=[0, 1, 9, 57, 130, 199])


def two_types_circle_oneshot():
    """two particle types, three entries (Circle, CountOverDuration with an offset window, OneShot Sphere),
    uneven f32 scale curve, constant dt"""
    p0 = S.ParticleSettings(lifetime=S.RandF32(0.5, 0.7), linear_drag=0.5,
                            base_color=S.FireworkGradient.even_samples([(1.0, 1.0, 1.0, 1.0), (0.0, 0.0, 0.0, 0.0)]))
    p1 = S.ParticleSettings(lifetime=S.RandF32.constant(0.25), acceleration=(0.0, 1.0, 0.0),
                            scale_curve=S.FireworkCurve.uneven_samples([(0.0, 1.0), (0.8, 1.2), (1.0, 0.0)]))
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(900.0),
                            emission_shape=S.EmissionShape.Circle((0.0, 0.0, 1.0), 2.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.0, 3.0), (0.0, 1.0, 0.0), 30.0 / 180.0 * math.pi))
    e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.CountOverDuration(500.0, 0.5, 0.2, 0.9),
                            initial_velocity=S.RandVec3.constant((0.0, 2.0, 0.0)))
    e2 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.OneShot(300),
                            initial_velocity_radial=S.RandF32(1.0, 2.0), emission_shape=S.EmissionShape.Sphere(1.0),
                            inherit_parent_velocity=False)
    return dict(spawner=S.ParticleSpawner([p0, p1], [e0, e1, e2]), transform=S.Transform((1.0, -2.0, 0.5)),
                modifier=None, parent_velocity=(0.0, 0.75, 0.0), uid=3, dts=[1 / 60], frames=120,
                checkpoints=[0, 7, 40, 119])


def nested_sparks_smoke():
    """Nested emission (core.rs:471-546): per-parent counts, parent-major child order, last_emitted_age; the smoke
    inherits nothing, the embers (a second Nested entry, on the smoke) inherit their parent's velocity"""
    sparks = S.ParticleSettings(lifetime=S.RandF32(1.0, 2.0), initial_scale=S.RandF32(0.01, 0.03), linear_drag=0.3,
                                base_color=S.FireworkGradient.even_samples([(8.0, 4.0, 1.0, 1.0), (1.0, 0.2, 0.0, 0.0)]))
    smoke = S.ParticleSettings(lifetime=S.RandF32(0.8, 1.4), initial_scale=S.RandF32(0.05, 0.1),
                               acceleration=(0.0, 0.5, 0.0), linear_drag=0.7,
                               scale_curve=S.FireworkCurve.even_samples([1.0, 3.0]),
                               base_color=S.FireworkGradient.uneven_samples([(0.0, (0.1, 0.1, 0.1, 0.0)),
                                                                            (0.1, (0.1, 0.1, 0.1, 0.15)),
                                                                            (1.0, (0.1, 0.1, 0.1, 0.0))]), pbr=True)
    embers = S.ParticleSettings(lifetime=S.RandF32(0.1, 0.3), initial_scale=S.RandF32.constant(0.01))
    e_sparks = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(70.0),
                                  initial_velocity=S.RandVec3(S.RandF32(2.0, 5.0), (0.0, 1.0, 0.0), 0.4),
                                  initial_angular_velocity=S.RandVec3(S.RandF32(5.0, 15.0), (0.0, -1.0, 0.0), 0.0))
    e_smoke = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                                 emission_pacing=S.EmissionPacing.CountOverDuration(20.0, 0.0, 0.0, 0.5),
                                 inherit_parent_velocity=False)
    e_embers = S.EmissionSettings(particle_index=2, emission_mode=S.EmissionMode.Nested(1),
                                  emission_pacing=S.EmissionPacing.CountOverDuration(2.0, 0.0, 0.3, 0.9),
                                  emission_shape=S.EmissionShape.Sphere(0.05),
                                  initial_velocity=S.RandVec3(S.RandF32(0.1, 0.4), (0.0, 1.0, 0.0), 1.0),
                                  inherit_parent_velocity=True)
    return dict(spawner=S.ParticleSpawner([sparks, smoke, embers], [e_sparks, e_smoke, e_embers]),
                transform=S.Transform((-2.0, 2.0, 0.0)), modifier=None, parent_velocity=(0.0, 0.0, 0.0), uid=11,
                dts=[1 / 60] * 50 + [1 / 45, 1 / 60, 1 / 75], frames=170, checkpoints=[0, 3, 45, 110, 169])


def deaths_everywhere():
    """no trigonometry anywhere (Point, zero spread, zero spin): every implementation must agree bit for bit;
    lifetimes spread so that every frame removes particles throughout the array (stable compaction order)"""
    ps = S.ParticleSettings(lifetime=S.RandF32(0.05, 1.5), linear_drag=0.3, initial_scale=S.RandF32(0.5, 2.0),
                            scale_curve=S.FireworkCurve.even_samples([1.0, 2.0, 0.5, 0.25]),
                            base_color=S.FireworkGradient.uneven_samples(STRESS_GRADIENT))
    es = S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(3000.0),
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (0.0, 1.0, 0.0), 0.0),
                            initial_velocity_radial=S.RandF32(0.0, 1.0))
    return dict(spawner=S.ParticleSpawner([ps], [es]), transform=S.Transform((1.0, 2.0, 3.0)), modifier=None,
                parent_velocity=(0.0, 0.0, 0.0), uid=9, dts=[1 / 60] * 20 + [1 / 50, 1 / 61.5, 1 / 58.7], frames=150,
                checkpoints=[0, 30, 95, 149], exact=True)


def bouncing_colliders():
    """particle_collision (core.rs:607-624, 744-800) against the analytic collider set: a tilted ground plane, a sphere,
    a rotated box, a tilted cylinder and a cone (the two kinds examples/textures.rs:195, 211 bounces its cases off; round 5);
    type 0 bounces (restitution, friction), type 1 is destroyed on contact, type 2 only collides
    with layer 2 (the small sphere high up); fast particles take several bounce sub-steps per frame.  No libm call
    anywhere in the collision arithmetic (sqrt and division are correctly rounded everywhere): bit-exact."""
    bounce = S.ParticleSettings(lifetime=S.RandF32(1.5, 3.0), linear_drag=0.05, initial_scale=S.RandF32(0.02, 0.05),
                                collision_settings=S.ParticleCollisionSettings(restitution=0.6, friction=0.3))
    fragile = S.ParticleSettings(lifetime=S.RandF32(2.0, 3.0), linear_drag=0.0,
                                 collision_settings=S.ParticleCollisionSettings(0.2, 0.9, destroy_on_collision=True))
    picky = S.ParticleSettings(lifetime=S.RandF32.constant(2.5), acceleration=(0.0, 2.0, 0.0), linear_drag=0.0,
                               collision_settings=S.ParticleCollisionSettings(1.0, 0.0, False, filter_mask=2))
    e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(600.0),
                            emission_shape=S.EmissionShape.Point(),
                            initial_velocity=S.RandVec3(S.RandF32(2.0, 14.0), (0.3, -1.0, 0.1), 0.0),
                            initial_velocity_radial=S.RandF32(0.0, 0.0))
    e1 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.rate(400.0),
                            initial_velocity=S.RandVec3(S.RandF32(1.0, 6.0), (-0.5, -1.0, 0.2), 0.0))
    e2 = S.EmissionSettings(particle_index=2, emission_pacing=S.EmissionPacing.rate(150.0),
                            initial_velocity=S.RandVec3(S.RandF32(0.5, 3.0), (0.0, 1.0, 0.0), 0.0))
    colliders = [S.Collider.Plane((0.0, -1.0, 0.0), (0.1, 1.0, 0.05)), S.Collider.Sphere((1.0, 0.0, 0.2), 0.8),
                 S.Collider.Box((-1.2, -0.2, 0.3), (0.7, 0.4, 0.9), (0.0, math.sin(0.35), 0.0, math.cos(0.35))),
                 S.Collider.Sphere((0.0, 4.5, 0.0), 0.6, layers=2),
                 S.Collider.Cylinder((0.3, 0.2, -0.4), 0.6, 0.5, (math.sin(0.2), 0.0, 0.0, math.cos(0.2))),
                 S.Collider.Cone((-0.3, 0.6, 0.5), 0.5, 1.0)]
    return dict(spawner=S.ParticleSpawner([bounce, fragile, picky], [e0, e1, e2]), transform=S.Transform((0.0, 2.0, 0.0)),
                modifier=None, parent_velocity=(0.0, 0.0, 0.0), uid=21, dts=[1 / 60] * 30 + [1 / 30, 1 / 120, 0.05],
                frames=160, checkpoints=[0, 25, 80, 159], exact=True, colliders=colliders)


ALL = {f.__name__: f for f in (rotation_cone_sphere, two_types_circle_oneshot, nested_sparks_smoke, deaths_everywhere,
                               bouncing_colliders)}
