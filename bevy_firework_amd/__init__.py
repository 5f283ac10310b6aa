"""bevy_firework_amd -- MI355X (gfx950) backend for bevy_firework's particle simulation path.

Settings types mirror the reference (``settings``); ``ParticleSystem`` drives the
HIP kernels through the C ABI in include/firework_hip.h.  Importing the settings
never needs a GPU; constructing a ``ParticleSystem`` does (no CPU fallback).
"""
from .settings import (  # noqa: F401
    BLACK, WHITE, EffectModifier, EmissionMode, EmissionPacing, EmissionSettings, EmissionShape, FireworkCurve,
    FireworkGradient, ParticleSettings, ParticleSpawner, RandF32, RandVec3, SpawnTransformMode, Transform,
    INSTANCE_DTYPE, PARTICLE_DTYPE,
)


def __getattr__(name):
    if name in ("ParticleSystem", "SpawnerData", "FwError", "compute_emission_count"):
        from . import system

        return getattr(system, name)
    raise AttributeError(name)
