//! The world `particle_collision` casts its rays into (core.rs:744-800), mirrored into the backend's device-resident set of
//! analytic colliders (UNVERIFIED SOURCE: no Rust toolchain in the build image).
//!
//! The reference asks avian's `SpatialQuery` (arbitrary parry shapes behind a CPU broadphase, core.rs:756-765).  The backend
//! keeps planes, spheres, oriented boxes, cylinders and cones on the GPU (`fw_collider`; ray-cast semantics in
//! `include/firework_hip.h`).  Entities opt in with the `ParticleCollider` marker; the set is replaced EVERY frame -- the call
//! does not wait for the frames in flight, the new set travels as one small copy in the context's stream -- so moving bodies
//! cost what the reference's per-frame query costs.  Shapes without an analytic counterpart (meshes, compounds, capsules) are
//! skipped: particles do not collide with them on this path.
use super::ffi::*;
use super::HipBackend;
use avian3d::prelude::*;
use bevy::prelude::*;

/// "particles bounce off this collider"
#[derive(Component, Default)]
pub struct ParticleCollider;

/// `SpatialQueryFilter::excluded_entities` (core.rs:247, 764) has no counterpart on the device (include/firework_hip.h): entities
/// listed here are sent with NO membership bit, so no particle type's mask selects them -- the one exclusion the backend can honour,
/// and it holds for every particle type of the context.  A filter that excludes an entity for ONE type only keeps that spawner on
/// the CPU systems.
#[derive(Resource, Default)]
pub struct ParticleColliderExclusions(pub bevy::platform::collections::HashSet<Entity>);

pub fn hip_sync_colliders(
    backend: NonSend<HipBackend>, excluded: Option<Res<ParticleColliderExclusions>>,
    q: Query<(Entity, &Collider, &GlobalTransform, Option<&CollisionLayers>), With<ParticleCollider>>,
) {
    let mut set = Vec::<fw_collider>::new();
    for (entity, collider, gt, layers) in &q {
        let t = gt.compute_transform();
        let out = excluded.as_ref().is_some_and(|x| x.0.contains(&entity));
        let base = fw_collider {
            kind: 0, layers: if out { 0 } else { layers.map_or(1, |l| l.memberships.0) }, position: t.translation.to_array(), rotation: t.rotation.to_array(),
            normal: [0., 1., 0.], radius: 0., half_extents: [0.; 3],
        };
        let shape = collider.shape_scaled();
        if let Some(b) = shape.as_ball() {
            set.push(fw_collider { kind: 1, radius: b.radius, ..base });                                          // Collider::sphere
        } else if let Some(c) = shape.as_cuboid() {
            set.push(fw_collider { kind: 2, half_extents: [c.half_extents.x, c.half_extents.y, c.half_extents.z], ..base }); // ::cuboid
        } else if let Some(c) = shape.as_cylinder() {
            set.push(fw_collider { kind: 3, radius: c.radius, half_extents: [0., c.half_height, 0.], ..base });   // ::cylinder (textures.rs:195)
        } else if let Some(c) = shape.as_cone() {
            set.push(fw_collider { kind: 4, radius: c.radius, half_extents: [0., c.half_height, 0.], ..base });   // ::cone (textures.rs:211)
        } else if let Some(h) = shape.as_halfspace() {
            let n = t.rotation * Vec3::new(h.normal.x, h.normal.y, h.normal.z);
            set.push(fw_collider { kind: 0, normal: n.to_array(), ..base });                                       // ::half_space
        }
    }
    unsafe { let _ = backend.check(fw_ctx_set_colliders(backend.ctx, set.as_ptr(), set.len() as u32)); }
}
