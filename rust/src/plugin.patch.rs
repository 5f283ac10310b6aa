// What replaces src/plugin.rs:46-60 when the `hip_backend` feature is on (UNVERIFIED SOURCE).
//
//        app.register_type::<ParticleSpawner>()
//            .add_plugins(render::CustomMaterialPlugin)
#[cfg(feature = "hip_backend")]
{
    use crate::hip::{hip_frame, hip_notify_finished, hip_sync_spawner_data, HipBackend};
    app.insert_non_send_resource(HipBackend::new(/* device */ 0, /* seed */ rand::random()).expect("libfirework_hip: no usable gfx950 device"))
        .add_systems(
            self.update_schedule,
            (
                ApplyDeferred,
                propagate_particle_spawner_modifier, // unchanged (core.rs:690-703)
                ApplyDeferred,
                hip_sync_spawner_data, // Changed<ParticleSpawner> -> fw_spawner_create / fw_spawner_update_settings (core.rs:343-365)
                #[cfg(feature = "physics_avian")]
                sync_parent_velocity, // unchanged (core.rs:706-736); its result travels in fw_ctx_set_parent_velocities
                #[cfg(feature = "physics_avian")]
                crate::hip::colliders::hip_sync_colliders, // the world particle_collision casts its rays into (core.rs:756-765)
                hip_frame,            // spawn_particles + update_particles (core.rs:367-670): ONE asynchronous call; dispatches the
                                      // particles_destroyed handlers (core.rs:660-667)
                crate::hip::render::hip_fill_instances, // the records the render extract reads (render.rs:368-423, 403); form B of
                                      // src/hip/render.rs uses hip_instance_windows here instead
                hip_notify_finished,  // notify_finished_particle_spawners (core.rs:674-688)
            )
                .chain(),
        );
}
#[cfg(not(feature = "hip_backend"))]
{
    // ... the reference's chain, unchanged (plugin.rs:46-60)
}
