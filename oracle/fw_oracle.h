/*
 * fw_oracle.h -- CPU ORACLE for the bevy_firework particle hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The shipped
 * library (libfirework_hip.so) never links, includes or calls anything here.
 *
 * It is a plain-C restatement of the reference algorithm, in the reference's
 * own data shape (array-of-structs, one heap vector of `last_emitted_age` per
 * particle, clone -> integrate -> filter-collect into a fresh vector):
 *
 *   compute_emission_count   /root/reference/src/core.rs:553-575
 *   spawn_particles          /root/reference/src/core.rs:367-551
 *   update_particles         /root/reference/src/core.rs:577-670  (non-avian arm)
 *   sync_spawner_data        /root/reference/src/core.rs:343-365
 *   active()                 /root/reference/src/core.rs:288-302
 *   notify_finished          /root/reference/src/core.rs:674-688
 *   FireworkCurve/Gradient   /root/reference/src/curve.rs:14-33,101-120,146-164,177-239
 *   EmissionShape            /root/reference/src/emission_shape.rs:18-39
 *
 * Third-party arithmetic that is NOT under /root/reference (Cargo.lock pins;
 * sources unavailable offline) is restated from the published algorithms:
 *   glam 0.32.1 (scalar Vec3/Quat paths), bevy_math 0.19.0 (EvenCore /
 *   UnevenCore::sample_with, Curve::sample_clamped, f32 lerp), bevy_color
 *   0.19.0 (Mix for LinearRgba), bevy_utilitarian 0.10.0 (RandF32, RandVec3,
 *   PitchYaw), rand 0.9.4 (f32 = (u32 >> 8) * 2^-24).
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - compute_emission_count: pinned by the reference's own unit test
 *     core.rs:806-834 (total in {22,23}) and cross-checked bit-for-bit against
 *     an independent numpy-float32 restatement (tests/golden/).
 *   - even gradient + Mix: pinned by curve.rs:246-258.
 *   - update_particles, uneven cores, f32 curves (VectorSpace::lerp:
 *     a * (1 - t) + b * t), quaternion path, particle_collision
 *     (core.rs:744-800 over this backend's analytic ray cast): the reference
 *     holds no test for them -> "parity unpinned"; anchored on the reference
 *     source lines above, on hand-derived KATs and on a second, independent
 *     array-oriented numpy restatement of the whole path
 *     (tests/golden/np_sim.py) whose multi-frame trajectories this oracle
 *     reproduces (tests/test_oracle_golden.py) and which the HIP path is
 *     compared with directly (tests/test_gpu_golden.py).
 *   - spawn ATTRIBUTES (RandF32/RandVec3/PitchYaw/shapes): the reference uses
 *     an unseeded thread-local RNG (rand::random), so they are unpinnable in
 *     principle; this oracle defines a Philox4x32-10 counter stream with the
 *     reference's draw ORDER, and the RandVec3/PitchYaw formulas are our
 *     reading of the published crate -> "parity unpinned (distributional)".
 *   The reference itself cannot be compiled here (Rust, no cargo, no network).
 */
#ifndef FW_ORACLE_H
#define FW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- settings (flattened mirror of core.rs:99-162) ------------------------ */

typedef struct { float min, max; } fwo_randf32;
typedef struct { fwo_randf32 magnitude; float direction[3]; float spread; } fwo_randvec3;

enum { FWO_CURVE_CONSTANT = 0, FWO_CURVE_EVEN = 1, FWO_CURVE_UNEVEN = 2 };
/* FireworkCurve<f32> (curve.rs:8-12): n keys; times used only when UNEVEN. */
typedef struct { int32_t kind; int32_t n; const float *times; const float *values; } fwo_curve;
/* FireworkGradient<LinearRgba> (curve.rs:171-175): values = n * rgba. */
typedef struct { int32_t kind; int32_t n; const float *times; const float *rgba; } fwo_gradient;

typedef struct {
    fwo_randf32 lifetime;
    fwo_curve scale_curve;
    fwo_randf32 initial_scale;
    float acceleration[3];
    float angular_acceleration[3];
    float linear_drag, angular_drag;
    fwo_gradient base_color, emissive_color;
    int32_t pbr;
    /* collision_settings: Option<ParticleCollisionSettings> (core.rs:137-138, 240-248; feature physics_avian) */
    int32_t coll_enabled;
    float coll_restitution, coll_friction;
    int32_t coll_destroy_on_collision;
    uint32_t coll_filter_mask;
} fwo_particle_settings;

/* the analytic collider set the oracle's ray cast runs against (stands in for avian's SpatialQuery; semantics in
 * include/firework_hip.h: fw_collider).  "Parity unpinned": the reference's world is arbitrary parry shapes. */
enum { FWO_COLLIDER_PLANE = 0, FWO_COLLIDER_SPHERE = 1, FWO_COLLIDER_BOX = 2, FWO_COLLIDER_CYLINDER = 3, FWO_COLLIDER_CONE = 4 };
typedef struct {
    int32_t kind;
    uint32_t layers;
    float position[3];
    float rotation[4];
    float normal[3];
    float radius;
    float half_extents[3];
} fwo_collider;

enum { FWO_PACING_ONESHOT = 0, FWO_PACING_ONDEMAND = 1, FWO_PACING_COUNT_OVER_DURATION = 2 };
enum { FWO_MODE_GLOBAL = 0, FWO_MODE_NESTED = 1 };
enum { FWO_SHAPE_POINT = 0, FWO_SHAPE_SPHERE = 1, FWO_SHAPE_CIRCLE = 2 };

typedef struct {
    int32_t particle_index;
    int32_t pacing_kind;
    uint64_t oneshot_count;
    float count, duration, offset_start, offset_end;
    int32_t mode;
    int32_t target_particle_type;
    int32_t shape_kind;
    float shape_radius;
    float shape_normal[3];
    fwo_randvec3 initial_velocity;
    fwo_randf32 initial_velocity_radial;
    int32_t inherit_parent_velocity;
    float initial_rotation[4]; /* xyzw */
    fwo_randvec3 initial_angular_velocity;
} fwo_emission_settings;

/* ---- one particle, as handed to callers (flat; core.rs:305-321) ----------- */
typedef struct {
    float position[3];
    float velocity[3];
    float rotation[4];
    float angular_velocity[3];
    float initial_scale, scale, age, lifetime;
    float base_color[4];
    float emissive_color[4];
    int32_t pbr;
} fwo_particle_flat; /* 26 x 4 B */

typedef struct fwo_spawner fwo_spawner;

/* ---- unit functions (KAT surface) ----------------------------------------- */
/* core.rs:553-575; returns the usize count saturated to uint64. */
uint64_t fwo_compute_emission_count(float time_passed_in_cycle, float last_emission, float cycle_duration,
                                    float offset_start, float offset_end, float particles_per_cycle,
                                    float *next_last_emission);
float fwo_rem_euclid(float a, float b);
float fwo_div_euclid(float a, float b);
/* UnevenCore::new normalisation (drop non-finite times, stable sort, dedup-keep-first); returns new n */
int32_t fwo_uneven_normalize(float *times, float *vals, int32_t n, int32_t stride);
float fwo_curve_sample_clamped(const fwo_curve *c, float t);
void fwo_gradient_sample_clamped(const fwo_gradient *g, float t, float out_rgba[4]);
void fwo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* the 12 uniforms of one spawned particle (stream definition in fw_oracle.c) */
void fwo_spawn_uniforms(uint32_t seed, uint32_t spawner_uid, uint32_t emission_index, uint64_t serial, float u[12]);
void fwo_shape_generate(const fwo_emission_settings *e, const float u[3], float out[3]);
void fwo_randvec3_generate(const fwo_randvec3 *r, float u_angle, float u_radius, float u_mag, float out[3]);
void fwo_quat_from_scaled_axis(const float v[3], float out[4]);
void fwo_quat_mul(const float a[4], const float b[4], float out[4]);
void fwo_quat_mul_vec3(const float q[4], const float v[3], float out[3]);
void fwo_quat_from_rotation_arc(const float from[3], const float to[3], float out[4]);

/* particle_collision (core.rs:744-800) against `n` colliders; pos / vel updated in place; returns should_destroy */
int32_t fwo_particle_collision(float pos[3], float vel[3], float delta, float restitution, float friction,
                               int32_t destroy_on_collision, uint32_t filter_mask, const fwo_collider *colliders,
                               int32_t n);

/* ---- spawner lifecycle ----------------------------------------------------- */
fwo_spawner *fwo_spawner_create(const fwo_particle_settings *ps, int32_t n_ps, const fwo_emission_settings *es,
                                int32_t n_es, int32_t starts_enabled, uint32_t seed, uint32_t uid);
void fwo_spawner_destroy(fwo_spawner *s);
/* sync_spawner_data (core.rs:343-365): reset emission state, drop all particles */
void fwo_spawner_reset(fwo_spawner *s);
void fwo_spawner_set_origin(fwo_spawner *s, const float translation[3], const float rotation[4]);
void fwo_spawner_set_parent_velocity(fwo_spawner *s, const float v[3]);
void fwo_spawner_set_modifier(fwo_spawner *s, float scale, float speed);
void fwo_spawner_queue(fwo_spawner *s, uint64_t n);
/* the world the spawner's particles collide with (copied) */
void fwo_spawner_set_colliders(fwo_spawner *s, const fwo_collider *colliders, int32_t n);
int32_t fwo_spawner_active(const fwo_spawner *s);
/* returns 1 exactly once, like notify_finished_particle_spawners (core.rs:674-688) */
int32_t fwo_spawner_poll_finished(fwo_spawner *s);

/* one frame: spawn_particles then update_particles (plugin.rs:46-60 order) */
void fwo_spawner_spawn(fwo_spawner *s, float dt);
void fwo_spawner_update(fwo_spawner *s, float dt);
void fwo_spawner_step(fwo_spawner *s, float dt);

uint64_t fwo_spawner_count(const fwo_spawner *s, int32_t type);
uint64_t fwo_spawner_read(const fwo_spawner *s, int32_t type, fwo_particle_flat *out, uint64_t cap);
/* last_emitted_age[emission_index] of every particle of `type` */
uint64_t fwo_spawner_read_last_emitted(const fwo_spawner *s, int32_t type, int32_t emission_index, float *out,
                                       uint64_t cap);
/* replace the particle vector of `type` (last_emitted_age = f32::MIN unless given) */
void fwo_spawner_write(fwo_spawner *s, int32_t type, const fwo_particle_flat *in, uint64_t n);
void fwo_spawner_write_last_emitted(fwo_spawner *s, int32_t type, int32_t emission_index, const float *in, uint64_t n);
/* particles destroyed by the LAST update of `type` (age already >= lifetime) */
uint64_t fwo_spawner_read_destroyed(const fwo_spawner *s, int32_t type, fwo_particle_flat *out, uint64_t cap);
/* update_aabbs reduction (render.rs:677-703) in world space: min/max of position -/+ scale */
int32_t fwo_spawner_aabb(const fwo_spawner *s, float out_min[3], float out_max[3]);

#ifdef __cplusplus
}
#endif
#endif
