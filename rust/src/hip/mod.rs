//! The HIP backend behind bevy_firework's ECS surface (UNVERIFIED SOURCE: no Rust toolchain in the build image).
//!
//! `ParticleSystemPlugin`, `ParticleSpawner`, `ParticleSpawnerData`, `ParticleData`, `EffectModifier` and
//! `ParticleSpawnerFinished` stay what they are; the systems `sync_spawner_data`, `spawn_particles`, `update_particles`
//! and `notify_finished_particle_spawners` (src/plugin.rs:46-60) are replaced by the three below, which call
//! `libfirework_hip.so` through `ffi.rs`.  Particle state lives on the GPU; `ParticleSpawnerData::particles` is filled on
//! demand (`HipSpawner::read_particles`) instead of every frame.
//!
//! Two accessors are needed in src/curve.rs, whose sample vectors are private to that module:
//! ```ignore
//! impl<T: Clone> FireworkCurve<T> {      // -> (kind 0 constant / 1 even / 2 uneven, times, values)
//!     pub fn hip_samples(&self) -> (i32, Vec<f32>, Vec<T>) { match self {
//!         FireworkCurve::Constant(c) => (0, vec![], vec![c.sample_unchecked(0.)]),
//!         FireworkCurve::SampleAuto(c) => (1, vec![], c.core.samples.clone()),
//!         FireworkCurve::UnevenSampleAuto(c) => (2, c.core.times.clone(), c.core.samples.clone()) } }
//! }
//! impl FireworkGradient<LinearRgba> { pub fn hip_samples(&self) -> (i32, Vec<f32>, Vec<LinearRgba>) { /* the same over its cores */ } }
//! ```
pub mod ffi;
#[cfg(feature = "physics_avian")]
pub mod colliders;
pub mod render; // the renderer's half of the hand-off (src/render.rs:368-423, 568-584, 677-703, 922-926)

use crate::core::*;
use crate::emission_shape::EmissionShape;
use bevy::prelude::*;
use ffi::*;
use std::collections::HashMap;
use std::ffi::CStr;

/// One context per GPU.  `!Send`: insert with `insert_non_send_resource` (calls on one context are serialised by its owner).
/// A world of thousands of small emitters keeps a few of these on one GPU, spawner `e` on backend `e % n`, stepped from `n`
/// tasks: contexts share nothing.
pub struct HipBackend {
    pub ctx: *mut fw_ctx,
    next_uid: u32,
    /// Entity -> device spawner, kept HERE and not only in the `HipSpawner` component: when an entity is despawned -- the normal
    /// end of a one-shot effect, `ParticleSpawnerFinished` then `despawn` (examples/one_shot.rs:137-141) -- its components are gone
    /// by the time `RemovedComponents<ParticleSpawner>` is read, and a lookup through the removed entity finds nothing: every
    /// despawned emitter would leak its device segments.
    spawners: HashMap<Entity, fw_spawner>,
}

/// The handle of a spawner's device-resident state, next to its `ParticleSpawnerData`.
#[derive(Component)]
pub struct HipSpawner(pub fw_spawner);

#[derive(Debug)]
pub struct HipError(pub i32, pub String);

impl HipBackend {
    pub fn new(device: i32, seed: u32) -> Result<Self, HipError> {
        let mut ctx = std::ptr::null_mut();
        let st = unsafe { fw_ctx_create(device, seed, std::ptr::null_mut(), &mut ctx) };
        if st != 0 {
            let msg = unsafe { CStr::from_ptr(fw_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(HipError(st, msg)); // FW_ENODEV: there is no CPU fallback in the library; the caller keeps the CPU systems
        }
        assert_eq!(unsafe { fw_abi_version() }, 5);
        Ok(Self { ctx, next_uid: 0, spawners: HashMap::new() })
    }
    pub fn check(&self, st: i32) -> Result<(), HipError> {
        if st == 0 { return Ok(()); }
        Err(HipError(st, unsafe { CStr::from_ptr(fw_last_error(self.ctx)) }.to_string_lossy().into_owned()))
    }
}
impl Drop for HipBackend {
    fn drop(&mut self) { unsafe { fw_ctx_destroy(self.ctx); } }
}

/// ParticleSpawner -> fw_spawner_desc.  The descriptor is copied by the call: the vectors only have to outlive it.
struct Desc {
    ps: Vec<fw_particle_settings>,
    es: Vec<fw_emission_settings>,
    _keep: Vec<Vec<f32>>, // curve sample storage the descriptors point into
}

fn rand_f32(r: &bevy_utilitarian::prelude::RandF32) -> fw_rand_f32 { fw_rand_f32 { min: r.min, max: r.max } }
fn rand_vec3(r: &bevy_utilitarian::prelude::RandVec3) -> fw_rand_vec3 {
    fw_rand_vec3 { magnitude: rand_f32(&r.magnitude), direction: r.direction.to_array(), spread: r.spread }
}

fn build_desc(s: &ParticleSpawner) -> Desc {
    let mut keep: Vec<Vec<f32>> = Vec::new();
    let mut hold = |v: Vec<f32>| -> *const f32 { keep.push(v); keep.last().unwrap().as_ptr() };
    let ps = s.particle_settings.iter().map(|p| {
        let (sk, st, sv) = p.scale_curve.hip_samples();
        let (bk, bt, bv) = p.base_color.hip_samples();
        let (ek, et, ev) = p.emissive_color.hip_samples();
        let rgba = |v: &[LinearRgba]| v.iter().flat_map(|c| [c.red, c.green, c.blue, c.alpha]).collect::<Vec<f32>>();
        #[cfg(feature = "physics_avian")]
        let collision = p.collision_settings.as_ref().map_or(fw_collision_settings::default(), |c| fw_collision_settings {
            enabled: 1, restitution: c.restitution, friction: c.friction, destroy_on_collision: c.destroy_on_collision as i32,
            filter_mask: c.filter.mask.0, // SpatialQueryFilter -> 32-bit layer mask (Collisions)
        });
        #[cfg(not(feature = "physics_avian"))]
        let collision = fw_collision_settings::default();
        fw_particle_settings {
            lifetime: rand_f32(&p.lifetime),
            scale_curve: fw_curve { kind: sk, n: sv.len() as i32, times: hold(st), values: hold(sv) },
            initial_scale: rand_f32(&p.initial_scale),
            acceleration: p.acceleration.to_array(), angular_acceleration: p.angular_acceleration.to_array(),
            linear_drag: p.linear_drag, angular_drag: p.angular_drag,
            base_color: fw_gradient { kind: bk, n: bv.len() as i32, times: hold(bt), rgba: hold(rgba(&bv)) },
            emissive_color: fw_gradient { kind: ek, n: ev.len() as i32, times: hold(et), rgba: hold(rgba(&ev)) },
            pbr: p.pbr as i32, report_destroyed: p.event_handlers.particles_destroyed.is_some() as i32,
            capacity: 0, // derived from the emitters; Vec-like growth on the device
            collision,
        }
    }).collect();
    let es = s.emission_settings.iter().map(|e| {
        let (pacing_kind, oneshot_count, count, duration, offset_start, offset_end) = match e.emission_pacing {
            EmissionPacing::OneShot(n) => (0, n as u64, 0., 0., 0., 0.),
            EmissionPacing::OnDemand => (1, 0, 0., 0., 0., 0.),
            EmissionPacing::CountOverDuration { count, duration, offset_start, offset_end } => (2, 0, count, duration, offset_start, offset_end),
        };
        let (mode, target) = match e.emission_mode {
            EmissionMode::Global => (0, 0),
            EmissionMode::Nested { target_particle_type } => (1, target_particle_type as i32),
        };
        let (shape_kind, shape_radius, shape_normal) = match e.emission_shape {
            EmissionShape::Point => (0, 0., [0., 1., 0.]),
            EmissionShape::Sphere(r) => (1, r, [0., 1., 0.]),
            EmissionShape::Circle { normal, radius } => (2, radius, normal.to_array()),
        };
        fw_emission_settings {
            particle_index: e.particle_index as i32, pacing_kind, oneshot_count, count, duration, offset_start, offset_end,
            mode, target_particle_type: target, shape_kind, shape_radius, shape_normal,
            initial_velocity: rand_vec3(&e.initial_velocity), initial_velocity_radial: rand_f32(&e.initial_velocity_radial),
            inherit_parent_velocity: e.inherit_parent_velocity as i32, initial_rotation: e.initial_rotation.to_array(),
            initial_angular_velocity: rand_vec3(&e.initial_angular_velocity),
        }
    }).collect();
    Desc { ps, es, _keep: keep }
}

/// sync_spawner_data (core.rs:343-365): `Changed<ParticleSpawner>` (insertion included) builds or rebuilds the device state --
/// emission state reset, all particles dropped, as in the reference.
pub fn hip_sync_spawner_data(
    mut backend: NonSendMut<HipBackend>, mut commands: Commands,
    spawners: Query<(Entity, &ParticleSpawner, Option<&HipSpawner>), Changed<ParticleSpawner>>,
    mut removed: RemovedComponents<ParticleSpawner>,
) {
    // (component removed OR entity despawned: the handle comes from the backend's own map, see HipBackend::spawners)
    for e in removed.read() {
        if let Some(h) = backend.spawners.remove(&e) {
            unsafe { fw_spawner_destroy(backend.ctx, h); }
            if let Ok(mut ec) = commands.get_entity(e) { ec.remove::<(HipSpawner, render::HipInstances)>(); }
        }
    }
    for (entity, settings, handle) in &spawners {
        let d = build_desc(settings);
        let uid = backend.next_uid; // RNG stream id: keep it stable across GPUs when sharding (desc.uid = global emitter index)
        let desc = fw_spawner_desc {
            particle_settings: d.ps.as_ptr(), n_particle_settings: d.ps.len() as u32,
            emission_settings: d.es.as_ptr(), n_emission_settings: d.es.len() as u32,
            starts_enabled: settings.starts_enabled as i32, uid,
        };
        match handle {
            Some(h) => { let _ = backend.check(unsafe { fw_spawner_update_settings(backend.ctx, h.0, &desc) }); }
            None => {
                let mut h: fw_spawner = -1;
                if backend.check(unsafe { fw_spawner_create(backend.ctx, &desc, &mut h) }).is_ok() {
                    backend.next_uid += 1;
                    backend.spawners.insert(entity, h);
                    commands.entity(entity).insert((HipSpawner(h), render::HipInstances::default()));
                } // FW_EINVAL mirrors the reference's panics (0-sample curves, indices out of range): log and skip
            }
        }
    }
}

/// spawn_particles + update_particles (core.rs:367-670) for EVERY spawner: the per-frame inputs in one FFI call each, then one
/// asynchronous `fw_step`.
pub fn hip_frame(
    backend: NonSend<HipBackend>, time: Res<Time>, mut commands: Commands,
    mut q: Query<(&Transform, &GlobalTransform, &ParticleSpawner, &mut ParticleSpawnerData, &HipSpawner, Option<&EffectModifier>)>,
) {
    let (mut hs, mut ts, mut rs) = (Vec::<fw_spawner>::new(), Vec::<f32>::new(), Vec::<f32>::new());
    let (mut vs, mut scales, mut speeds) = (Vec::<f32>::new(), Vec::<f32>::new(), Vec::<f32>::new());
    let (mut qh, mut qn) = (Vec::<fw_spawner>::new(), Vec::<u64>::new());
    for (t, gt, settings, mut data, h, m) in &mut q {
        let origin = match settings.spawn_transform_mode { // core.rs:432-435
            SpawnTransformMode::Global => gt.compute_transform(),
            SpawnTransformMode::Local => *t,
        };
        hs.push(h.0);
        ts.extend_from_slice(&origin.translation.to_array());
        rs.extend_from_slice(&origin.rotation.to_array());
        vs.extend_from_slice(&data.parent_velocity.to_array()); // core.rs:276, written by sync_parent_velocity
        let m = m.copied().unwrap_or_default();
        scales.push(m.scale);
        speeds.push(m.speed);
        let queued = std::mem::take(&mut data.manual_queued_count) as u64; // core.rs:284-286
        if queued != 0 { qh.push(h.0); qn.push(queued); }
    }
    unsafe {
        let n = hs.len() as u32;
        let _ = backend.check(fw_ctx_set_origins(backend.ctx, n, hs.as_ptr(), ts.as_ptr(), rs.as_ptr()));
        let _ = backend.check(fw_ctx_set_parent_velocities(backend.ctx, n, hs.as_ptr(), vs.as_ptr()));
        let _ = backend.check(fw_ctx_set_modifiers(backend.ctx, n, hs.as_ptr(), scales.as_ptr(), speeds.as_ptr()));
        let _ = backend.check(fw_ctx_queue(backend.ctx, qh.len() as u32, qh.as_ptr(), qn.as_ptr()));
        // asynchronous: enqueues the frame's launch(es).  FW_EHIP with "internal error" in fw_last_error: the spawner named there
        // is INVALID until it is rebuilt (re-insert its ParticleSpawner) or despawned -- INTEGRATION.md section 3, Errors
        if let Err(e) = backend.check(fw_step(backend.ctx, time.delta_secs())) { error!("fw_step: {} {}", e.0, e.1); }
    }
    // particles_destroyed handlers (core.rs:660-667): `run_system_with(handler, destroyed)` for every particle type that registered
    // one and lost particles in this step.  Reading the records waits for the frame -- only spawners with a handler pay for it
    // (`report_destroyed` was set from `event_handlers.particles_destroyed.is_some()` in build_desc); the records hold what the
    // reference's `destroyed` vector holds: age already >= lifetime, pose of the previous frame for an age death (core.rs:596-599),
    // of this frame for a collision death (core.rs:636-639).
    let mut buf = Vec::<fw_particle>::new();
    for (_, _, settings, _, h, _) in &q {
        for (ty, p) in settings.particle_settings.iter().enumerate() {
            let Some(handler) = p.event_handlers.particles_destroyed else { continue };
            let mut n = 0u64;
            unsafe { fw_spawner_read_destroyed(backend.ctx, h.0, ty as u32, std::ptr::null_mut(), 0, &mut n); } // size query
            if n == 0 { continue; }
            buf.clear();
            buf.reserve(n as usize);
            let st = unsafe { fw_spawner_read_destroyed(backend.ctx, h.0, ty as u32, buf.as_mut_ptr(), n, &mut n) };
            if backend.check(st).is_err() { continue; }
            unsafe { buf.set_len(n as usize); }
            let n_emissions = settings.emission_settings.len();
            let destroyed: Vec<ParticleData> = buf.iter().map(|r| particle_data(r, n_emissions)).collect();
            commands.run_system_with(handler, destroyed); // deferred to the next sync point, like ParallelCommands in the reference
        }
    }
}

/// fw_particle (104-byte record of the ABI) -> ParticleData (core.rs:305-321).  `last_emitted_age` is not part of the record: a
/// handler that needs it reads `fw_spawner_read_last_emitted`; the vector has the reference's length and its initial value
/// (core.rs:467).
pub fn particle_data(r: &fw_particle, n_emissions: usize) -> ParticleData {
    ParticleData {
        position: Vec3::from_array(r.position), velocity: Vec3::from_array(r.velocity), rotation: Quat::from_array(r.rotation),
        angular_velocity: Vec3::from_array(r.angular_velocity), initial_scale: r.initial_scale, scale: r.scale, age: r.age,
        lifetime: r.lifetime,
        base_color: LinearRgba::new(r.base_color[0], r.base_color[1], r.base_color[2], r.base_color[3]),
        emissive_color: LinearRgba::new(r.emissive_color[0], r.emissive_color[1], r.emissive_color[2], r.emissive_color[3]),
        pbr: r.pbr != 0, last_emitted_age: vec![f32::MIN; n_emissions],
    }
}

/// notify_finished_particle_spawners (core.rs:674-688).  Reads device counts: this is the one call of the frame that WAITS for
/// the GPU (INTEGRATION.md "when does it pay"): a host that does not need the event every frame polls every few frames.
pub fn hip_notify_finished(backend: NonSend<HipBackend>, mut commands: Commands, q: Query<(Entity, &HipSpawner)>) {
    for (entity, h) in &q {
        let mut fin = 0i32;
        if unsafe { fw_spawner_poll_finished(backend.ctx, h.0, &mut fin) } == 0 && fin != 0 {
            commands.trigger(ParticleSpawnerFinished { entity });
        }
    }
}

impl HipSpawner {
    /// `ParticleSpawnerData::particles[ty]` on demand (it stays a public field for users who read it)
    pub fn read_particles(&self, backend: &HipBackend, ty: u32) -> Vec<fw_particle> {
        let mut n = 0u64;
        unsafe { fw_spawner_read_particles(backend.ctx, self.0, ty, std::ptr::null_mut(), 0, &mut n); }
        let mut v = Vec::<fw_particle>::with_capacity(n as usize);
        unsafe {
            fw_spawner_read_particles(backend.ctx, self.0, ty, v.as_mut_ptr(), n, &mut n);
            v.set_len(n as usize);
        }
        v
    }
}
