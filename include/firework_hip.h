/*
 * firework_hip.h -- C ABI of libfirework_hip.so, the MI355X (gfx950) backend for
 * bevy_firework's per-frame particle simulation path.
 *
 * The reference has no FFI: the path sits behind Bevy's system registration
 *   (sync_spawner_data, spawn_particles, update_particles,
 *    notify_finished_particle_spawners).chain()        reference src/plugin.rs:46-60
 * operating on the components ParticleSpawner (src/core.rs:178-185, user-owned
 * settings) and ParticleSpawnerData (src/core.rs:269-303, plugin-owned state).
 * This header is what a Rust shim crate would bind (INTEGRATION.md shows the
 * `extern "C"` block and the exclusive system that replaces the two CPU systems).
 * Each entry point cites the reference item it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; all input descriptors are copied at the call.
 *  - every function returns fw_status (0 = ok, negative = error) unless noted;
 *    fw_last_error() gives a message.  The library never aborts the process; the
 *    reference's panics (zero-key curve curve.rs:45,61,211,227; out-of-range
 *    particle_index / target_particle_type core.rs:392,453,488) become FW_EINVAL
 *    at create time.
 *  - calls on ONE context must be serialised by the caller (the reference chain is sequential too).  Contexts share no
 *    state: different contexts -- on one GPU or on several -- may be driven from different threads at the same time.
 *    One context per GPU is the normal arrangement; a host with thousands of small emitters, whose frame is bound by the
 *    host half of fw_step (~30 ns per emitter on one thread), spreads them over a few contexts on the same GPU, one per
 *    worker thread -- the counterpart of the reference's par_iter_mut over spawners (core.rs:583-585); the device runs
 *    their launches side by side (examples/many_contexts.cpp).  fw_last_error(NULL) is per calling thread.
 *  - fw_step only ENQUEUES work on the
 *    context's HIP stream; readers synchronise that stream.  A context created on a CALLER-SUPPLIED stream keeps the
 *    whole frame on that stream: work the caller orders behind fw_step on it (or hipStreamSynchronize of it) covers
 *    the frame.  A context that owns its stream (stream = NULL at fw_ctx_create) may run part of a frame on a second,
 *    internal stream (the in-place update of ring segments next to the compacting launch): everything the library
 *    enqueues on the context's stream that looks at particle data -- attached instance buffers, the *_device entry
 *    points -- is ordered after it by the library itself, and fw_ctx_synchronize waits for both; wait for a frame of
 *    such a context with fw_ctx_synchronize, not hipStreamSynchronize(fw_ctx_stream(ctx)).
 *  - there is NO CPU fallback: without a usable HIP device fw_ctx_create fails
 *    with FW_ENODEV.
 */
#ifndef FIREWORK_HIP_H
#define FIREWORK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_ABI_VERSION 5
/* (no FW_MAX_TYPES / FW_MAX_EMISSIONS / FW_MAX_KEYS / FW_MAX_COLLIDERS: the reference's Vec<ParticleSettings>,
 * Vec<EmissionSettings> (core.rs:178-185), curve sample vectors (curve.rs:40-75) and collider world are unbounded, and so
 * are the descriptors below -- FW_EINVAL is for input the reference itself rejects.  Curves and gradients of up to
 * FW_FAST_KEYS samples are staged in LDS; longer ones are read from device memory by the feature kernels.) */
#define FW_FAST_KEYS 32

typedef enum fw_status {
    FW_OK = 0,
    FW_EINVAL = -1,    /* bad argument / descriptor (mirrors the reference's panics) */
    FW_ENOMEM = -2,    /* host or device allocation failed */
    FW_EHIP = -3,      /* a HIP call failed, or an internal consistency check of an update kernel did (fw_last_error) */
    FW_ECAPACITY = -4, /* a particle type overflowed its device capacity (nested emission) */
    FW_ENODEV = -5,    /* no usable HIP device / kernels not loadable */
    FW_ESMALL = -6     /* output buffer too small; required size reported */
} fw_status;

/* Internal errors are STICKY PER SPAWNER.  The update kernels check the host's bookkeeping against the particles they load
 * (which particles a step destroys, live counts, look-back waits).  A failed check means the library's own state is wrong;
 * where the type is updated in place (ring paths) the frame has overwritten its input and cannot be redone.  The library
 * then does not guess: the spawner the particle type belongs to is marked invalid, fw_step returns FW_EHIP without
 * enqueuing anything (for any spawner: the frame is all-or-nothing) and so does every call that reads or writes that
 * spawner's particles, until fw_spawner_update_settings rebuilds it -- sync_spawner_data (core.rs:343-365): emission state
 * reset, all particles dropped; the rebuilt spawner stays off the in-place paths -- or fw_spawner_destroy removes it.  Other
 * spawners keep their state. */
typedef struct fw_ctx fw_ctx;
typedef int32_t fw_spawner; /* handle, >= 0 */

/* bevy_utilitarian RandF32 / RandVec3 as used by core.rs:102,107,155,157,161 */
typedef struct fw_rand_f32 { float min, max; } fw_rand_f32;
typedef struct fw_rand_vec3 { fw_rand_f32 magnitude; float direction[3]; float spread; } fw_rand_vec3;

/* FireworkCurve<f32> (curve.rs:8-12) and FireworkGradient<LinearRgba> (curve.rs:171-175) */
enum { FW_CURVE_CONSTANT = 0, FW_CURVE_EVEN = 1, FW_CURVE_UNEVEN = 2 };
typedef struct fw_curve { int32_t kind; int32_t n; const float *times; const float *values; } fw_curve;
typedef struct fw_gradient { int32_t kind; int32_t n; const float *times; const float *rgba; } fw_gradient;

/* ParticleCollisionSettings (core.rs:240-248, feature physics_avian): `enabled` = the Option is Some.
 * `filter_mask` stands in for SpatialQueryFilter (core.rs:247, passed to cast_ray at core.rs:764): a collider takes part when
 * (filter_mask & collider.layers) != 0 -- avian's `mask` against the collider's `memberships`.
 * NOT SUPPORTED: `SpatialQueryFilter::excluded_entities`.  The device-resident set has no entity identity: a host that needs
 * an exclusion keeps the excluded colliders out of the set it sends (fw_ctx_set_colliders; the set is per context, so this
 * excludes them for every particle type of the context) or gives them a membership bit no particle type's mask contains
 * (rust/src/hip/colliders.rs does the latter for entities listed in a `ParticleColliderExclusions` resource). */
typedef struct fw_collision_settings {
    int32_t enabled;
    float restitution, friction;
    int32_t destroy_on_collision;
    uint32_t filter_mask;
} fw_collision_settings;

/* ParticleSettings (core.rs:99-142), simulation-relevant fields only; textures,
 * fade_*, blend_mode stay on the host (they never enter update_particles). */
typedef struct fw_particle_settings {
    fw_rand_f32 lifetime;
    fw_curve scale_curve;
    fw_rand_f32 initial_scale;
    float acceleration[3];
    float angular_acceleration[3];
    float linear_drag, angular_drag;
    fw_gradient base_color, emissive_color;
    int32_t pbr;
    int32_t report_destroyed; /* event_handlers.particles_destroyed.is_some() (core.rs:164-167) */
    uint32_t capacity;        /* device slots for this type; 0 = derive from the emitters */
    fw_collision_settings collision; /* collision_settings: Option<ParticleCollisionSettings> (core.rs:137-138) */
} fw_particle_settings;

/* The world particle_collision (core.rs:744-800) casts its rays into.  The reference asks avian's SpatialQuery
 * (arbitrary colliders, CPU broadphase); this backend keeps a DEVICE-RESIDENT set of analytic colliders instead and
 * casts against them inside the update.  Ray-cast semantics (ours, modelled on parry's `solid = true` casts):
 *   PLANE   the half-space n.(x - position) <= 0 is solid; `normal` must be a unit vector
 *   SPHERE  |x - position| <= radius is solid
 *   BOX     |R^-1 (x - position)|_i <= half_extents_i is solid (R = rotation, xyzw)
 *   CYLINDER  avian's Collider::cylinder(radius, height) (examples/textures.rs:195): in the collider's frame (R, position) the
 *           axis is Y; |y| <= half_extents[1] (= height / 2) and x^2 + z^2 <= radius^2 is solid; a hit on a cap reports +-Y, a hit
 *           on the lateral surface the radial direction (both rotated by R)
 *   CONE    avian's Collider::cone(radius, height) (examples/textures.rs:211): base disc of `radius` at y = -half_extents[1], apex
 *           at y = +half_extents[1]; solid between them; the base reports -Y, the lateral surface its outward normal
 *   a ray that starts inside a solid hits it at distance 0 with a ZERO normal (core.rs:762-771 handles that case);
 *   otherwise the hit is the entry point, its normal the outward surface normal; the nearest hit over all colliders
 *   that pass the filter wins (lowest index on ties). */
enum { FW_COLLIDER_PLANE = 0, FW_COLLIDER_SPHERE = 1, FW_COLLIDER_BOX = 2, FW_COLLIDER_CYLINDER = 3, FW_COLLIDER_CONE = 4 };
typedef struct fw_collider {
    int32_t kind;
    uint32_t layers;        /* collision layers (membership bits) */
    float position[3];
    float rotation[4];      /* xyzw; BOX, CYLINDER, CONE */
    float normal[3];        /* PLANE only */
    float radius;           /* SPHERE, CYLINDER, CONE */
    float half_extents[3];  /* BOX; [1] = half the height of a CYLINDER / CONE */
} fw_collider;

enum { FW_PACING_ONESHOT = 0, FW_PACING_ONDEMAND = 1, FW_PACING_COUNT_OVER_DURATION = 2 }; /* core.rs:12-29 */
enum { FW_MODE_GLOBAL = 0, FW_MODE_NESTED = 1 };                                           /* core.rs:47-54 */
enum { FW_SHAPE_POINT = 0, FW_SHAPE_SPHERE = 1, FW_SHAPE_CIRCLE = 2 };                     /* emission_shape.rs:7-15 */

/* EmissionSettings (core.rs:144-162) */
typedef struct fw_emission_settings {
    int32_t particle_index;
    int32_t pacing_kind;
    uint64_t oneshot_count;
    float count, duration, offset_start, offset_end;
    int32_t mode;
    int32_t target_particle_type;
    int32_t shape_kind;
    float shape_radius;
    float shape_normal[3];
    fw_rand_vec3 initial_velocity;
    fw_rand_f32 initial_velocity_radial;
    int32_t inherit_parent_velocity;
    float initial_rotation[4]; /* xyzw */
    fw_rand_vec3 initial_angular_velocity;
} fw_emission_settings;

/* ParticleSpawner (core.rs:178-185).  spawn_transform_mode is resolved by the host:
 * it passes the chosen transform to fw_spawner_set_origin (core.rs:432-435). */
typedef struct fw_spawner_desc {
    const fw_particle_settings *particle_settings;
    uint32_t n_particle_settings;
    const fw_emission_settings *emission_settings;
    uint32_t n_emission_settings;
    int32_t starts_enabled;
    uint32_t uid; /* RNG stream id; keep it stable across GPUs when sharding */
} fw_spawner_desc;

/* ParticleData (core.rs:305-321) as an AoS record for readback / upload.
 * last_emitted_age is read separately (fw_spawner_read_last_emitted). */
typedef struct fw_particle {
    float position[3];
    float velocity[3];
    float rotation[4];
    float angular_velocity[3];
    float initial_scale, scale, age, lifetime;
    float base_color[4];
    float emissive_color[4];
    int32_t pbr;
} fw_particle;

/* ParticleInstance (render.rs:95-103): 64 B */
typedef struct fw_particle_instance {
    float position[3];
    float scale;
    float rotation[4];
    float base_color[4];
    float emissive_color[4];
} fw_particle_instance;

/* ---- context ---------------------------------------------------------------- */
/* `stream` = an existing hipStream_t to enqueue on (e.g. torch's current stream),
 * or NULL to let the context create its own. */
int fw_abi_version(void);
fw_status fw_ctx_create(int device, uint32_t seed, void *stream, fw_ctx **out);
fw_status fw_ctx_destroy(fw_ctx *ctx);
const char *fw_last_error(const fw_ctx *ctx); /* ctx may be NULL: last create error */
void *fw_ctx_stream(const fw_ctx *ctx);       /* the hipStream_t in use */
fw_status fw_ctx_synchronize(fw_ctx *ctx);

/* replaces the context's collider set (copied; any n; n = 0 clears it).  Takes effect at the next fw_step.  Does NOT
 * synchronise: the set travels as one copy in the context's stream, behind the frames that read the old one (the
 * reference queries the live physics world every frame, core.rs:756-765 -- moving colliders cost one small copy per
 * frame).  Only a set larger than any before reallocates the device table, which waits for the frames in flight. */
fw_status fw_ctx_set_colliders(fw_ctx *ctx, const fw_collider *colliders, uint32_t n);

/* ---- spawners ----------------------------------------------------------------- */
/* ParticleSpawner insertion + first sync_spawner_data (core.rs:343-365) */
fw_status fw_spawner_create(fw_ctx *ctx, const fw_spawner_desc *desc, fw_spawner *out);
/* Changed<ParticleSpawner>: sync_spawner_data again -- resets emission state, drops all particles */
fw_status fw_spawner_update_settings(fw_ctx *ctx, fw_spawner h, const fw_spawner_desc *desc);
fw_status fw_spawner_destroy(fw_ctx *ctx, fw_spawner h);

/* per-frame inputs the ECS owns */
fw_status fw_spawner_set_origin(fw_ctx *ctx, fw_spawner h, const float translation[3], const float rotation_xyzw[4]);
/* ... for ALL spawners in one call: spawn_particles walks every spawner entity in one system (core.rs:377), and a host with
 * thousands of them would otherwise cross the FFI once per spawner per frame.  handles[n], translations[n][3],
 * rotations_xyzw[n][4]; all-or-nothing: one invalid handle -> FW_EINVAL and no origin changes. */
fw_status fw_ctx_set_origins(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *translations,
                             const float *rotations_xyzw);
fw_status fw_spawner_set_parent_velocity(fw_ctx *ctx, fw_spawner h, const float v[3]); /* core.rs:276,444-448 */
fw_status fw_spawner_set_modifier(fw_ctx *ctx, fw_spawner h, float scale, float speed); /* EffectModifier core.rs:323-327 */
fw_status fw_spawner_queue(fw_ctx *ctx, fw_spawner h, uint64_t count);                  /* queue_particles core.rs:284-286 */
/* ... and the same three for MANY spawners in one call each (ABI 5).  The reference rewrites these inputs for whole sets of
 * spawners every frame -- sync_parent_velocity walks every spawner under a rigid body (core.rs:706-736),
 * propagate_particle_spawner_modifier every spawner under an EffectModifier (core.rs:690-703), a gameplay system queues
 * particles on every OnDemand spawner it owns -- and a host with thousands of emitters would cross the FFI once per spawner
 * per frame for each.  handles[n]; velocities[n][3]; scales[n], speeds[n]; counts[n] (added to what is queued, core.rs:284-286).
 * All-or-nothing like fw_ctx_set_origins: one invalid handle -> FW_EINVAL and nothing changes. */
fw_status fw_ctx_set_parent_velocities(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *velocities);
fw_status fw_ctx_set_modifiers(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *scales, const float *speeds);
fw_status fw_ctx_queue(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const uint64_t *counts);

/* ---- the frame: spawn_particles then update_particles for every spawner -------- */
fw_status fw_step(fw_ctx *ctx, float dt); /* core.rs:367-551 + 577-670 */

/* ---- outputs (synchronise the stream) ------------------------------------------- */
/* particles[i].len() for every type (core.rs:274) */
fw_status fw_spawner_counts(fw_ctx *ctx, fw_spawner h, uint32_t *per_type, uint32_t n_types);
/* ParticleSpawnerData::active (core.rs:288-302): *out = 0/1 */
fw_status fw_spawner_active(fw_ctx *ctx, fw_spawner h, int32_t *out);
/* notify_finished_particle_spawners (core.rs:674-688): *out = 1 exactly once */
fw_status fw_spawner_poll_finished(fw_ctx *ctx, fw_spawner h, int32_t *out);
/* copies min(count, cap) records; *n_out = count.  Order = reference Vec order. */
fw_status fw_spawner_read_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out);
fw_status fw_spawner_read_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index, float *out,
                                       uint64_t cap, uint64_t *n_out);
/* replaces the particle vector of `type` (`particles` is a pub field in the reference) */
fw_status fw_spawner_write_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, const fw_particle *in, uint64_t n);
fw_status fw_spawner_write_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index,
                                        const float *in, uint64_t n);
/* particles destroyed by the last fw_step for a type with report_destroyed (core.rs:588,596-599,660-667) */
fw_status fw_spawner_read_destroyed(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out);
/* ParticleInstance packing (render.rs:105-115,403) into a HOST buffer */
fw_status fw_spawner_pack_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle_instance *out, uint64_t cap,
                                    uint64_t *n_out);
/* same, into a DEVICE buffer on the context's stream (no sync): the render hand-off */
fw_status fw_spawner_pack_instances_device(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap,
                                           uint64_t *n_upper_bound);
/* Render hand-off fused into the update (render.rs:403 builds these records on the CPU every frame): from the next
 * fw_step on, the update kernel itself also writes the ParticleInstance record of every particle of (spawner, type) that
 * survives the step into d_out[0 .. live count) -- device memory, `cap` records, particle order -- so the frame needs
 * no packing pass.  Records beyond `cap` are dropped.  Types that receive Nested children work too: children are spawned
 * before the update of the same frame (plugin.rs:46-60), so they are among the records.
 * d_out = NULL detaches; fw_spawner_update_settings (which rebuilds the particle types) detaches too.
 * Synchronises the DEVICE once (the segment record changes; and whatever the caller enqueued on its own streams to
 * initialise the buffer has completed before a frame writes into it -- the same holds for fw_ctx_live_count_ring). */
fw_status fw_spawner_attach_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap);
/* The same hand-off for a host that can draw an instance SUB-RANGE (every graphics API can: firstInstance): the records of
 * the particles that survive a step are d_out[first, first + count), particle order, with `first` and `count` reported by
 * fw_spawner_instance_window after the step (one readback: the count has to be read anyway).  `first` is 0 on most update
 * paths; a particle type with a lifetime RANGE that the library keeps in a ring numbers its records from the particles the
 * step destroys (first = their number): with the plain attach above such a type is moved to the compacting path, with this
 * one it keeps its in-place update.  The buffer is indexed from 0, not from `first`: `cap` must cover first + count --
 * a buffer of the particle type's capacity always does (first + count never exceeds the live count before the step);
 * records at an index >= cap are dropped, so a buffer sized for the live count alone loses its last `first` records.
 * d_out = NULL detaches. */
fw_status fw_spawner_attach_instances_window(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap);
fw_status fw_spawner_instance_window(fw_ctx *ctx, fw_spawner h, uint32_t type, uint64_t *first, uint64_t *count);
/* update_aabbs reduction (render.rs:677-703), world space; *any = 0 when no particles */
fw_status fw_spawner_aabb(fw_ctx *ctx, fw_spawner h, float out_min[3], float out_max[3], int32_t *any);
/* AABB fused into the update: from the next fw_step on, every tile of the update kernel also leaves the box of
 * position -/+ scale of the survivors it stored (no extra pass over the particles, no extra launch in the frame);
 * fw_spawner_aabb then folds a few hundred 32-byte tile boxes instead of re-reading every particle.  Same result bit for
 * bit.  Frames with colliding particle types, and queries after the state was touched outside fw_step, fall back to the
 * two-pass reduction. */
fw_status fw_ctx_track_aabbs(fw_ctx *ctx, int32_t enable);

/* ---- whole-context statistics ---------------------------------------------------- */
/* total live particles over all spawners (host value; synchronises) */
fw_status fw_ctx_live_count(fw_ctx *ctx, uint64_t *out);
/* enqueue a write of the total live count into a caller-owned DEVICE uint64 (no
 * sync): feed for the RCCL all-reduce of live counts across GPUs */
fw_status fw_ctx_live_count_device(fw_ctx *ctx, void *d_out_u64);
/* register a caller-owned DEVICE ring of n_slots (>= 2) uint64: from now on every fw_step leaves the total live
 * count of its frame in slot (k mod n_slots), k = frames stepped since registration, at no extra launch
 * (written by the update kernel itself).  NULL unregisters.  Bucketed RCCL all-reduce feed. */
fw_status fw_ctx_live_count_ring(fw_ctx *ctx, void *d_ring_u64, uint32_t n_slots);
/* running total of particles that entered update_particles (after spawn) since the context was created */
fw_status fw_ctx_last_step_updated(fw_ctx *ctx, uint64_t *out);

/* (measurement and debugging hooks -- kernel timing, the copy-bandwidth probe, in-kernel timestamps, which update path a
 * particle type is on -- are declared in firework_hip_debug.h: exported by the same library, not part of the surface a
 * host binds) */

/* ---- pure host helpers (no GPU needed; the bit-exact count arithmetic) ------------ */
/* compute_emission_count (core.rs:553-575) exactly as fw_step's host side evaluates it */
uint64_t fw_compute_emission_count(float time_passed_in_cycle, float last_emission, float cycle_duration,
                                   float offset_start, float offset_end, float particles_per_cycle,
                                   float *next_last_emission);

#ifdef __cplusplus
}
#endif
#endif /* FIREWORK_HIP_H */
