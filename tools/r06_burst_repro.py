"""round 6: the scenario of tests/test_gpu_golden.py::test_global_burst_into_a_nested_fed_type_stays_in_bounds, in a loop, to be run by
several processes at once (tools/r06_contention_repro.sh found `check 2 of an update kernel failed` in ~1 of 8 runs when six
processes share the GPU, never alone).  With the `ab` build and FW_TRACE=1 the tail of the trace is kept when a run fails."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_firework_amd import settings as S  # noqa: E402
from bevy_firework_amd.system import FwError, ParticleSystem  # noqa: E402

DT = np.float32(1.0 / 60.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for rep in range(reps):
    with ParticleSystem(device=0, seed=1234) as system:
        sparks = S.ParticleSettings(lifetime=S.RandF32.constant(1.0))
        mixed = S.ParticleSettings(lifetime=S.RandF32.constant(0.5), capacity=4096)
        e0 = S.EmissionSettings(particle_index=0, emission_pacing=S.EmissionPacing.rate(600.0))
        e1 = S.EmissionSettings(particle_index=1, emission_mode=S.EmissionMode.Nested(0),
                                emission_pacing=S.EmissionPacing.CountOverDuration(4.0, 0.0, 0.0, 1.0))
        e2 = S.EmissionSettings(particle_index=1, emission_pacing=S.EmissionPacing.OnDemand())
        h = system.spawn(S.ParticleSpawner([sparks, mixed], [e0, e1, e2]), uid=1)
        nb = system.spawn(S.ParticleSpawner([S.ParticleSettings(lifetime=S.RandF32.constant(2.0))],
                                            [S.EmissionSettings(emission_pacing=S.EmissionPacing.rate(20000.0))]), uid=2)
        try:
            for fr in range(80):
                if fr in (5, 6, 20):
                    h.queue_particles(30000)
                sys.stderr.write(f"[repro] rep {rep} frame {fr}\n")
                system.update(DT)
            try:
                h.counts()
            except FwError as e:
                if e.status != -4:
                    raise
            system.synchronize()
        except FwError as e:
            bad += 1
            sys.stderr.write(f"[repro] FAILED rep {rep}: {e}\n")
            try:
                system.synchronize()
            except FwError as e2:
                sys.stderr.write(f"[repro] at sync: {e2}\n")
            print(f"FAILED rep {rep}: {e}", flush=True)
            break
print(f"{bad} failures in {reps} reps", flush=True)
