//! `extern "C"` binding of `libfirework_hip.so` -- one item per entry point of `include/firework_hip.h` (ABI version 5).
//!
//! UNVERIFIED SOURCE: the image this backend was built in has no Rust toolchain.  `tests/test_abi_cpu.py` checks that the
//! functions declared here are exactly the ones the header declares and the library exports; the struct layouts follow the
//! `#[repr(C)]` mirrors the Python / C++ hosts are tested with (`bevy_firework_amd/_ffi.py`, `include/firework.hpp`).
//! The same text is section 2 of INTEGRATION.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct fw_ctx { _private: [u8; 0] }
pub type fw_spawner = i32;

#[repr(C)] #[derive(Clone, Copy)] pub struct fw_rand_f32 { pub min: f32, pub max: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct fw_rand_vec3 { pub magnitude: fw_rand_f32, pub direction: [f32; 3], pub spread: f32 }
#[repr(C)] pub struct fw_curve    { pub kind: i32, pub n: i32, pub times: *const f32, pub values: *const f32 }
#[repr(C)] pub struct fw_gradient { pub kind: i32, pub n: i32, pub times: *const f32, pub rgba: *const f32 }

#[repr(C)] pub struct fw_particle_settings {          // ParticleSettings, core.rs:99-142
    pub lifetime: fw_rand_f32, pub scale_curve: fw_curve, pub initial_scale: fw_rand_f32,
    pub acceleration: [f32; 3], pub angular_acceleration: [f32; 3], pub linear_drag: f32, pub angular_drag: f32,
    pub base_color: fw_gradient, pub emissive_color: fw_gradient,
    pub pbr: i32, pub report_destroyed: i32, pub capacity: u32,
    pub collision: fw_collision_settings,             // collision_settings: Option<..>, core.rs:137-138
}
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct fw_collision_settings {   // ParticleCollisionSettings, core.rs:240-248
    pub enabled: i32, pub restitution: f32, pub friction: f32, pub destroy_on_collision: i32, pub filter_mask: u32,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct fw_collider {   // one analytic collider of the device-resident world
    pub kind: i32, pub layers: u32, pub position: [f32; 3], pub rotation: [f32; 4], pub normal: [f32; 3],
    pub radius: f32, pub half_extents: [f32; 3],
}
#[repr(C)] pub struct fw_emission_settings {          // EmissionSettings, core.rs:144-162
    pub particle_index: i32, pub pacing_kind: i32, pub oneshot_count: u64,
    pub count: f32, pub duration: f32, pub offset_start: f32, pub offset_end: f32,
    pub mode: i32, pub target_particle_type: i32,
    pub shape_kind: i32, pub shape_radius: f32, pub shape_normal: [f32; 3],
    pub initial_velocity: fw_rand_vec3, pub initial_velocity_radial: fw_rand_f32,
    pub inherit_parent_velocity: i32, pub initial_rotation: [f32; 4], pub initial_angular_velocity: fw_rand_vec3,
}
#[repr(C)] pub struct fw_spawner_desc {               // ParticleSpawner, core.rs:178-185
    pub particle_settings: *const fw_particle_settings, pub n_particle_settings: u32,
    pub emission_settings: *const fw_emission_settings, pub n_emission_settings: u32,
    pub starts_enabled: i32, pub uid: u32,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct fw_particle {   // ParticleData, core.rs:305-321
    pub position: [f32; 3], pub velocity: [f32; 3], pub rotation: [f32; 4], pub angular_velocity: [f32; 3],
    pub initial_scale: f32, pub scale: f32, pub age: f32, pub lifetime: f32,
    pub base_color: [f32; 4], pub emissive_color: [f32; 4], pub pbr: i32,
}
#[repr(C)] #[derive(Clone, Copy, bytemuck::Pod, bytemuck::Zeroable)]
pub struct fw_particle_instance {                      // == render::ParticleInstance, render.rs:95-103
    pub position: [f32; 3], pub scale: f32, pub rotation: [f32; 4], pub base_color: [f32; 4], pub emissive_color: [f32; 4],
}

extern "C" {
    pub fn fw_abi_version() -> c_int;
    pub fn fw_ctx_create(device: c_int, seed: u32, stream: *mut c_void, out: *mut *mut fw_ctx) -> c_int;
    pub fn fw_ctx_destroy(ctx: *mut fw_ctx) -> c_int;
    pub fn fw_last_error(ctx: *const fw_ctx) -> *const c_char;
    pub fn fw_ctx_stream(ctx: *const fw_ctx) -> *mut c_void;
    pub fn fw_ctx_synchronize(ctx: *mut fw_ctx) -> c_int;
    pub fn fw_ctx_set_colliders(ctx: *mut fw_ctx, colliders: *const fw_collider, n: u32) -> c_int;
    pub fn fw_spawner_create(ctx: *mut fw_ctx, desc: *const fw_spawner_desc, out: *mut fw_spawner) -> c_int;
    pub fn fw_spawner_update_settings(ctx: *mut fw_ctx, h: fw_spawner, desc: *const fw_spawner_desc) -> c_int;
    pub fn fw_spawner_destroy(ctx: *mut fw_ctx, h: fw_spawner) -> c_int;
    pub fn fw_spawner_set_origin(ctx: *mut fw_ctx, h: fw_spawner, t: *const f32, r_xyzw: *const f32) -> c_int;
    pub fn fw_ctx_set_origins(ctx: *mut fw_ctx, n: u32, handles: *const fw_spawner, translations: *const f32, rotations_xyzw: *const f32) -> c_int;
    pub fn fw_spawner_set_parent_velocity(ctx: *mut fw_ctx, h: fw_spawner, v: *const f32) -> c_int;
    pub fn fw_spawner_set_modifier(ctx: *mut fw_ctx, h: fw_spawner, scale: f32, speed: f32) -> c_int;
    pub fn fw_spawner_queue(ctx: *mut fw_ctx, h: fw_spawner, count: u64) -> c_int;
    pub fn fw_ctx_set_parent_velocities(ctx: *mut fw_ctx, n: u32, handles: *const fw_spawner, velocities: *const f32) -> c_int;
    pub fn fw_ctx_set_modifiers(ctx: *mut fw_ctx, n: u32, handles: *const fw_spawner, scales: *const f32, speeds: *const f32) -> c_int;
    pub fn fw_ctx_queue(ctx: *mut fw_ctx, n: u32, handles: *const fw_spawner, counts: *const u64) -> c_int;
    pub fn fw_step(ctx: *mut fw_ctx, dt: f32) -> c_int;
    pub fn fw_spawner_counts(ctx: *mut fw_ctx, h: fw_spawner, per_type: *mut u32, n_types: u32) -> c_int;
    pub fn fw_spawner_active(ctx: *mut fw_ctx, h: fw_spawner, out: *mut i32) -> c_int;
    pub fn fw_spawner_poll_finished(ctx: *mut fw_ctx, h: fw_spawner, out: *mut i32) -> c_int;
    pub fn fw_spawner_read_particles(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, out: *mut fw_particle, cap: u64, n: *mut u64) -> c_int;
    pub fn fw_spawner_read_last_emitted(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, emission: u32, out: *mut f32, cap: u64, n: *mut u64) -> c_int;
    pub fn fw_spawner_write_particles(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, input: *const fw_particle, n: u64) -> c_int;
    pub fn fw_spawner_write_last_emitted(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, emission: u32, input: *const f32, n: u64) -> c_int;
    pub fn fw_spawner_read_destroyed(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, out: *mut fw_particle, cap: u64, n: *mut u64) -> c_int;
    pub fn fw_spawner_pack_instances(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, out: *mut fw_particle_instance, cap: u64, n: *mut u64) -> c_int;
    pub fn fw_spawner_pack_instances_device(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, d_out: *mut c_void, cap: u64, n_ub: *mut u64) -> c_int;
    pub fn fw_spawner_attach_instances(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, d_out: *mut c_void, cap: u64) -> c_int;
    pub fn fw_spawner_attach_instances_window(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, d_out: *mut c_void, cap: u64) -> c_int;
    pub fn fw_spawner_instance_window(ctx: *mut fw_ctx, h: fw_spawner, ty: u32, first: *mut u64, count: *mut u64) -> c_int;
    pub fn fw_spawner_aabb(ctx: *mut fw_ctx, h: fw_spawner, min: *mut f32, max: *mut f32, any: *mut i32) -> c_int;
    pub fn fw_ctx_track_aabbs(ctx: *mut fw_ctx, enable: i32) -> c_int;
    pub fn fw_ctx_live_count(ctx: *mut fw_ctx, out: *mut u64) -> c_int;
    pub fn fw_ctx_live_count_device(ctx: *mut fw_ctx, d_out_u64: *mut c_void) -> c_int;
    pub fn fw_ctx_live_count_ring(ctx: *mut fw_ctx, d_ring_u64: *mut c_void, n_slots: u32) -> c_int;
    pub fn fw_ctx_last_step_updated(ctx: *mut fw_ctx, out: *mut u64) -> c_int;
    pub fn fw_compute_emission_count(t: f32, last: f32, dur: f32, start: f32, end: f32, per_cycle: f32, next: *mut f32) -> u64;
}
