//! The renderer's half of the hand-off with the HIP backend (UNVERIFIED SOURCE: no Rust toolchain in the build image).
//!
//! In the reference the render world reads `ParticleSpawnerData::particles` directly:
//!   * `extract_component` (src/render.rs:368-423) turns every non-empty `particles[i]` into a `ParticleMaterialData`
//!     (`Vec<ParticleInstance>`, one CPU pass per frame: `particles.iter().map(|p| p.into()).collect()`, src/render.rs:403),
//!   * `prepare_instance_buffers` (src/render.rs:568-584) uploads it into a fresh vertex buffer,
//!   * `DrawFirework` (src/render.rs:905-926) draws `0..6, 0..length`,
//!   * `update_aabbs` (src/render.rs:677-703) folds `position -/+ scale` over every particle of the spawner, twice.
//! With the backend the particles live on the GPU and `particles` stays empty, so an unchanged renderer would draw nothing.
//! This file is what replaces those four items under the `hip_backend` feature.  Two forms of the hand-off:
//!
//! **A. Through host memory (works with any wgpu backend; what this file implements in full).**  A main-world system,
//! `hip_fill_instances`, runs after `hip_frame` and packs the records of every VISIBLE spawner
//! (`fw_spawner_pack_instances`: the device packs -- `fw_k_pack`, 64 B per particle -- into a staging buffer and copies it
//! out) into the `HipInstances` component; `extract_component` moves the vectors into `ParticleMaterialData` exactly where
//! the reference builds them from `ParticleData`.  `prepare_instance_buffers` and the draw are unchanged.  Cost: one
//! device -> host -> device round trip of 64 B per particle per frame over PCIe instead of the reference's CPU pass.
//!
//! **B. Through device memory (zero-copy; needs the vertex buffer's allocation visible to HIP).**  When wgpu runs on Vulkan
//! and the instance buffer is created from an exportable allocation (`VK_KHR_external_memory_fd` -> `hipImportExternalMemory`
//! -> `hipExternalMemoryGetMappedBuffer`), the mapped pointer is attached ONCE with `fw_spawner_attach_instances_window`: the
//! update kernel then writes the records itself every frame (no packing pass) and the extract only asks for
//! `(first, count)` (`fw_spawner_instance_window`).  The draw becomes the sub-range below.  `HipInstanceWindow` carries the
//! range through the render world; the interop calls themselves are outside this crate (wgpu-hal `Device::buffer_from_raw`).
//!
//! Both forms leave `ParticleInstance`'s layout (src/render.rs:95-103) and the vertex attributes (src/render.rs:737-766)
//! untouched: `fw_particle_instance` is that struct.
use super::ffi::*;
use super::{HipBackend, HipSpawner};
use crate::core::*;
use bevy::camera::primitives::Aabb;
use bevy::prelude::*;

/// Per spawner, per particle type: the packed `ParticleInstance` records of this frame (form A).  Filled by
/// `hip_fill_instances`, consumed by the extract (below).  `fw_particle_instance` is `#[repr(C)]`-identical to
/// `render::ParticleInstance` (checked by `const _: () = assert!(size_of::<ParticleInstance>() == 64)` in render.rs).
#[derive(Component, Default)]
pub struct HipInstances {
    pub per_type: Vec<Vec<fw_particle_instance>>,
}

/// Form B: the live records of an attached device buffer are `buffer[first .. first + count]` (a particle type
/// with a lifetime range numbers its records from the particles the step destroyed; 0 on every other update path).
#[derive(Component, Clone, Copy, Default)]
pub struct HipInstanceWindow {
    pub first: u32,
    pub count: u32,
}

/// Runs in the update schedule right after `hip_frame` (src/plugin.patch.rs).  Only spawners the renderer will extract:
/// `ViewVisibility` is what `extract_component`'s caller filters on (src/render.rs:382-403 runs per visible entity).
pub fn hip_fill_instances(
    backend: NonSend<HipBackend>,
    mut q: Query<(&ParticleSpawner, &HipSpawner, &mut HipInstances, &ViewVisibility)>,
) {
    for (settings, h, mut inst, vis) in &mut q {
        let n_types = settings.particle_settings.len();
        inst.per_type.resize_with(n_types, Vec::new);
        if !vis.get() {
            for v in &mut inst.per_type { v.clear(); }
            continue;
        }
        // one call for the counts of every type (synchronises with the frame's launch), then one packing pass per non-empty type
        let mut counts = vec![0u32; n_types];
        if backend.check(unsafe { fw_spawner_counts(backend.ctx, h.0, counts.as_mut_ptr(), n_types as u32) }).is_err() { continue; }
        for (ty, out) in inst.per_type.iter_mut().enumerate() {
            out.clear();
            let n = counts[ty] as usize;
            if n == 0 { continue; }
            out.reserve(n);
            let mut got = 0u64;
            let st = unsafe { fw_spawner_pack_instances(backend.ctx, h.0, ty as u32, out.as_mut_ptr(), n as u64, &mut got) };
            if backend.check(st).is_ok() { unsafe { out.set_len(got.min(n as u64) as usize); } }
        }
    }
}

// ---- src/render.rs:368-423 with the backend.  The ExtractComponent query gains `&HipInstances`; everything but the source of the
// records is the reference's code (flags, FireworkUniform, FireworkImages, RenderLayers), elided here as in the original:
//
// fn extract_component(item: (&ParticleSpawnerData, &ParticleSpawner, Option<&RenderLayers>, &HipInstances)) -> Vec<(..)> {
//     let (_data, settings, render_layers, inst) = item;
//     inst.per_type.iter().enumerate()
//         .filter(|(_, records)| !records.is_empty())                         // was: data.particles ... !particles.is_empty()
//         .map(|(index, records)| {
//             let particle_settings = &settings.particle_settings[index];
//             /* flags, as in src/render.rs:388-398 */
//             (FireworkRenderEntityMarker,
//              ParticleMaterialData {
//                  particles: bytemuck::cast_slice::<fw_particle_instance, ParticleInstance>(records).to_vec(),   // was :403
//                  alpha_mode: particle_settings.blend_mode.into() },
//              /* FireworkUniform, FireworkImages, render layers: unchanged, src/render.rs:406-420 */)
//         }).collect()
// }
//
// Form B replaces `particles: Vec<ParticleInstance>` by the shared buffer + `HipInstanceWindow`:

/// Form B, main world, after `hip_frame`: where the live records of every attached type are this frame.
pub fn hip_instance_windows(
    backend: NonSend<HipBackend>, mut commands: Commands,
    q: Query<(Entity, &HipSpawner, &ViewVisibility), With<SharedInstanceBuffer>>,
) {
    for (e, h, vis) in &q {
        if !vis.get() { continue; }
        let (mut first, mut count) = (0u64, 0u64);
        // (type 0 of the spawner; a spawner with several particle types keeps one window component per type)
        if backend.check(unsafe { fw_spawner_instance_window(backend.ctx, h.0, 0, &mut first, &mut count) }).is_ok() {
            commands.entity(e).insert(HipInstanceWindow { first: first as u32, count: count as u32 });
        }
    }
}

/// Marker + the HIP-visible pointer of a vertex buffer shared with wgpu (form B); attached once when the buffer is created:
/// `fw_spawner_attach_instances_window(ctx, h, ty, ptr, capacity)`.
#[derive(Component)]
pub struct SharedInstanceBuffer {
    pub device_ptr: *mut std::ffi::c_void,
    pub capacity: u64,
}

// ---- src/render.rs:922-926 for form B: the sub-range draw (every graphics API has firstInstance):
//
//     let w = instance_window.unwrap();                       // ItemQuery gains Read<HipInstanceWindow>
//     pass.set_vertex_buffer(0, instance_buffer.unwrap().buffer.slice(..));
//     pass.draw(0..6, w.first..w.first + w.count);            // was: pass.draw(0..6, 0..buffer_length)

/// update_aabbs (src/render.rs:677-703) through `fw_spawner_aabb`: min / max of `position -/+ scale` over every particle of every
/// type of the spawner come from the device -- fused into the update when `fw_ctx_track_aabbs(ctx, 1)` was called once after
/// `HipBackend::new` (a few hundred 32-byte tile boxes folded per query instead of two passes over the particles; same bits,
/// min / max are exact) -- and the centre is brought into the entity's local space as in the reference.
pub fn hip_update_aabbs(backend: NonSend<HipBackend>, mut q: Query<(&mut Aabb, &GlobalTransform, &HipSpawner)>) {
    for (mut aabb, global_transform, h) in &mut q {
        let (mut mn, mut mx, mut any) = ([0f32; 3], [0f32; 3], 0i32);
        if backend.check(unsafe { fw_spawner_aabb(backend.ctx, h.0, mn.as_mut_ptr(), mx.as_mut_ptr(), &mut any) }).is_err() || any == 0 {
            continue; // (the reference `continue`s on a spawner without particle vectors, src/render.rs:679-681)
        }
        let (min, max) = (Vec3::from_array(mn), Vec3::from_array(mx));
        let center = (min + max) / 2.;
        let half_extents = (max - min) / 2.;
        aabb.center = global_transform.to_matrix().inverse().transform_point3(center).into(); // src/render.rs:696-700
        aabb.half_extents = half_extents.into();
    }
}

// ---- registration (src/render.rs `CustomMaterialPlugin::build`): `.add_systems(Last, update_aabbs)` (src/render.rs:45) becomes
// `.add_systems(Last, hip_update_aabbs)`, and in the update schedule (src/plugin.patch.rs) `hip_fill_instances` (form A) or
// `hip_instance_windows` (form B) right after `hip_frame`.
