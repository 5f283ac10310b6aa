// firework.hpp -- C++ host-side mirror of the reference's plugin surface over the C ABI
// (include/firework_hip.h).  The reference is compiled code (Rust); with no Rust toolchain in the
// build image this header plays the role of the shim crate: same type names, defaults, argument
// meaning and error behaviour (the reference's panics become exceptions carrying fw_last_error).
//
//   reference item                                  here
//   ParticleSystemPlugin       src/plugin.rs:22-61  firework::ParticleSystemPlugin (owns one GPU context;
//                                                   update(dt) = the chained per-frame systems)
//   ParticleSpawner            src/core.rs:178-185  firework::ParticleSpawner
//   ParticleSettings           src/core.rs:99-142   firework::ParticleSettings   (defaults core.rs:187-211)
//   EmissionSettings           src/core.rs:144-162  firework::EmissionSettings   (defaults core.rs:213-227)
//   EmissionPacing / Mode      src/core.rs:11-54    firework::EmissionPacing / EmissionMode
//   EmissionShape              src/emission_shape.rs firework::EmissionShape
//   FireworkCurve / Gradient   src/curve.rs         firework::FireworkCurve / FireworkGradient
//   ParticleSpawnerData        src/core.rs:269-303  firework::ParticleSpawnerData (queue_particles, active, particles)
//   ParticleData               src/core.rs:305-321  fw_particle
//   EffectModifier             src/core.rs:323-336  firework::EffectModifier
//   ParticleCollisionSettings  src/core.rs:240-248  firework::ParticleCollisionSettings (+ firework::Collider, set_colliders)
//
// Header-only; link with -lfirework_hip.  No simulation arithmetic lives here.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "firework_hip.h"
#include "firework_hip_debug.h"  // update_path(): which kernels a particle type runs on (diagnostics)

namespace firework {

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Quat { float x = 0, y = 0, z = 0, w = 1; };
struct LinearRgba {
    float red = 1, green = 1, blue = 1, alpha = 1;
    static LinearRgba WHITE() { return {1, 1, 1, 1}; }
    static LinearRgba BLACK() { return {0, 0, 0, 1}; }
};

struct Error : std::runtime_error {
    fw_status status;
    Error(fw_status s, const std::string &m) : std::runtime_error(m), status(s) {}
};

struct RandF32 {
    float min = 0, max = 0;
    static RandF32 constant(float v) { return {v, v}; }
};

struct RandVec3 {
    RandF32 magnitude;
    Vec3 direction;
    float spread = 0;
    static RandVec3 constant(Vec3 v) {  // bevy_utilitarian: direction = v.normalize_or_zero(), magnitude = |v|
        const float len = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
        const float rcp = 1.0f / len;
        Vec3 d{0, 0, 0};
        if (std::isfinite(rcp) && rcp > 0) d = {v.x * rcp, v.y * rcp, v.z * rcp};
        return {RandF32::constant(len), d, 0.0f};
    }
};

// FireworkCurve<f32> (curve.rs:8-75)
struct FireworkCurve {
    int32_t kind = FW_CURVE_CONSTANT;
    std::vector<float> times, values;
    static FireworkCurve constant(float v) { return {FW_CURVE_CONSTANT, {}, {v}}; }
    static FireworkCurve even_samples(std::vector<float> s) {
        if (s.empty()) throw Error(FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:61
        if (s.size() == 1) return constant(s[0]);
        return {FW_CURVE_EVEN, {}, std::move(s)};
    }
    static FireworkCurve uneven_samples(const std::vector<std::pair<float, float>> &s) {
        if (s.empty()) throw Error(FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:45
        if (s.size() == 1) return constant(s[0].second);
        FireworkCurve c{FW_CURVE_UNEVEN, {}, {}};
        for (auto &p : s) c.times.push_back(p.first), c.values.push_back(p.second);
        return c;
    }
};

// FireworkGradient<LinearRgba> (curve.rs:171-239)
struct FireworkGradient {
    int32_t kind = FW_CURVE_CONSTANT;
    std::vector<float> times, rgba;
    static FireworkGradient constant(LinearRgba c) { return {FW_CURVE_CONSTANT, {}, {c.red, c.green, c.blue, c.alpha}}; }
    static FireworkGradient even_samples(const std::vector<LinearRgba> &s) {
        if (s.empty()) throw Error(FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:227
        if (s.size() == 1) return constant(s[0]);
        FireworkGradient g{FW_CURVE_EVEN, {}, {}};
        for (auto &c : s) g.rgba.insert(g.rgba.end(), {c.red, c.green, c.blue, c.alpha});
        return g;
    }
    static FireworkGradient uneven_samples(const std::vector<std::pair<float, LinearRgba>> &s) {
        if (s.empty()) throw Error(FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:211
        if (s.size() == 1) return constant(s[0].second);
        FireworkGradient g{FW_CURVE_UNEVEN, {}, {}};
        for (auto &p : s) {
            g.times.push_back(p.first);
            g.rgba.insert(g.rgba.end(), {p.second.red, p.second.green, p.second.blue, p.second.alpha});
        }
        return g;
    }
};

// EmissionPacing (core.rs:12-44)
struct EmissionPacing {
    int32_t kind = FW_PACING_COUNT_OVER_DURATION;
    uint64_t one_shot = 0;
    float count = 5, duration = 1, offset_start = 0, offset_end = 1;
    static EmissionPacing OneShot(uint64_t n) { return {FW_PACING_ONESHOT, n, 0, 1, 0, 1}; }
    static EmissionPacing OnDemand() { return {FW_PACING_ONDEMAND, 0, 0, 1, 0, 1}; }
    static EmissionPacing CountOverDuration(float count, float duration, float start, float end) {
        return {FW_PACING_COUNT_OVER_DURATION, 0, count, duration, start, end};
    }
    static EmissionPacing rate(float r) { return CountOverDuration(r, 1.0f, 0.0f, 1.0f); }  // core.rs:36-43
    bool is_one_shot() const { return kind == FW_PACING_ONESHOT; }
};

// EmissionMode (core.rs:47-54)
struct EmissionMode {
    int32_t kind = FW_MODE_GLOBAL;
    int32_t target_particle_type = 0;
    static EmissionMode Global() { return {}; }
    static EmissionMode Nested(int32_t target) { return {FW_MODE_NESTED, target}; }
};

// EmissionShape (emission_shape.rs:6-15)
struct EmissionShape {
    int32_t kind = FW_SHAPE_POINT;
    float radius = 0;
    Vec3 normal{0, 1, 0};
    static EmissionShape Point() { return {}; }
    static EmissionShape Sphere(float r) { return {FW_SHAPE_SPHERE, r, {0, 1, 0}}; }
    static EmissionShape Circle(Vec3 normal, float r) { return {FW_SHAPE_CIRCLE, r, normal}; }
};

enum class SpawnTransformMode { Global, Local };  // core.rs:66-73

// ParticleCollisionSettings (core.rs:240-248, feature physics_avian).  filter_mask stands in for the
// SpatialQueryFilter: a collider takes part when (filter_mask & collider.layers) != 0.
struct ParticleCollisionSettings {
    float restitution = 0, friction = 0;
    bool destroy_on_collision = false;
    uint32_t filter_mask = 0xFFFFFFFFu;
};

// One analytic collider of the world particle_collision casts its rays into (core.rs:744-800): the backend keeps a
// device-resident set of these instead of asking avian's SpatialQuery (semantics: fw_collider in firework_hip.h).
struct Collider {
    int32_t kind = FW_COLLIDER_PLANE;
    Vec3 position{};
    Quat rotation{};
    Vec3 normal{0, 1, 0};
    float radius = 0;
    Vec3 half_extents{};
    uint32_t layers = 1;
    static Collider Plane(Vec3 point, Vec3 unit_normal, uint32_t layers = 1) {
        Collider c; c.kind = FW_COLLIDER_PLANE, c.position = point, c.normal = unit_normal, c.layers = layers; return c;
    }
    static Collider Sphere(Vec3 center, float radius, uint32_t layers = 1) {
        Collider c; c.kind = FW_COLLIDER_SPHERE, c.position = center, c.radius = radius, c.layers = layers; return c;
    }
    static Collider Box(Vec3 center, Vec3 half_extents, Quat rotation = {}, uint32_t layers = 1) {
        Collider c; c.kind = FW_COLLIDER_BOX, c.position = center, c.half_extents = half_extents, c.rotation = rotation;
        c.layers = layers; return c;
    }
    // avian's Collider::cylinder(radius, height) / Collider::cone(radius, height) (examples/textures.rs:195, 211): axis = local Y
    static Collider Cylinder(Vec3 center, float radius, float height, Quat rotation = {}, uint32_t layers = 1) {
        Collider c; c.kind = FW_COLLIDER_CYLINDER, c.position = center, c.radius = radius, c.half_extents = Vec3{0, height * 0.5f, 0};
        c.rotation = rotation, c.layers = layers; return c;
    }
    static Collider Cone(Vec3 center, float radius, float height, Quat rotation = {}, uint32_t layers = 1) {
        Collider c; c.kind = FW_COLLIDER_CONE, c.position = center, c.radius = radius, c.half_extents = Vec3{0, height * 0.5f, 0};
        c.rotation = rotation, c.layers = layers; return c;
    }
};

struct ParticleSettings {  // core.rs:99-142, defaults core.rs:187-211
    RandF32 lifetime = RandF32::constant(5.0f);
    FireworkCurve scale_curve = FireworkCurve::constant(1.0f);
    RandF32 initial_scale = RandF32::constant(1.0f);
    Vec3 acceleration{0.0f, -9.81f, 0.0f};
    Vec3 angular_acceleration{0, 0, 0};
    float linear_drag = 0.2f, angular_drag = 0.2f;
    FireworkGradient base_color = FireworkGradient::constant(LinearRgba::WHITE());
    FireworkGradient emissive_color = FireworkGradient::constant(LinearRgba::BLACK());
    float fade_edge = 0.7f, fade_scene = 1.0f;  // render-only, carried for parity
    bool pbr = false;
    // event_handlers.particles_destroyed (core.rs:164-167)
    std::function<void(const std::vector<fw_particle> &)> particles_destroyed;
    bool has_collision_settings = false;            // collision_settings: Option<..> (core.rs:137-138)
    ParticleCollisionSettings collision_settings{};
    uint32_t capacity = 0;  // backend knob
};

struct EmissionSettings {  // core.rs:144-162, defaults core.rs:213-227
    int32_t particle_index = 0;
    EmissionPacing emission_pacing = EmissionPacing::rate(5.0f);
    EmissionMode emission_mode = EmissionMode::Global();
    EmissionShape emission_shape = EmissionShape::Point();
    RandVec3 initial_velocity = RandVec3::constant({0, 0, 0});
    RandF32 initial_velocity_radial = RandF32::constant(0.0f);
    bool inherit_parent_velocity = true;
    Quat initial_rotation{};
    RandVec3 initial_angular_velocity = RandVec3::constant({0, 0, 0});
};

struct ParticleSpawner {  // core.rs:178-185, defaults core.rs:229-238
    std::vector<ParticleSettings> particle_settings{ParticleSettings{}};
    std::vector<EmissionSettings> emission_settings{EmissionSettings{}};
    bool starts_enabled = true;
    SpawnTransformMode spawn_transform_mode = SpawnTransformMode::Global;
};

struct EffectModifier { float scale = 1, speed = 1; };             // core.rs:323-336
struct Transform { Vec3 translation{}; Quat rotation{}; };          // the fields spawn_particles reads

class ParticleSystemPlugin;

// ParticleSpawnerData (core.rs:269-303): handle to the device-resident state of one spawner
class ParticleSpawnerData {
  public:
    void queue_particles(uint64_t count);  // core.rs:284-286
    bool active();                         // core.rs:288-302
    std::vector<uint32_t> counts();
    std::vector<fw_particle> particles(uint32_t particle_type);  // data.particles[i]
    std::vector<fw_particle> destroyed(uint32_t particle_type);
    std::vector<fw_particle_instance> instances(uint32_t particle_type);  // render.rs:95-115
    // render hand-off fused into the update (fw_spawner_attach_instances): device buffer of `cap` 64-byte records
    void attach_instances(void *device_buffer, uint64_t cap, uint32_t particle_type = 0);
    // ... for a renderer that draws an instance sub-range: the live records are buffer[first, first + count) (instance_window);
    // lets a lifetime-range type keep its in-place ring (fw_spawner_attach_instances_window)
    void attach_instances_window(void *device_buffer, uint64_t cap, uint32_t particle_type = 0) {
        check_(fw_spawner_attach_instances_window(raw_(), handle, particle_type, device_buffer, cap));
    }
    std::pair<uint64_t, uint64_t> instance_window(uint32_t particle_type = 0) {
        uint64_t first = 0, count = 0;
        check_(fw_spawner_instance_window(raw_(), handle, particle_type, &first, &count));
        return {first, count};
    }
    bool aabb(Vec3 &mn, Vec3 &mx);                                        // render.rs:677-703
    // which kernel family updates a particle type: true = in-place FIFO ring (types with one lifetime value)
    bool on_fifo_path(uint32_t particle_type = 0) {
        int32_t mode = 0;
        check_(fw_debug_update_path(raw_(), handle, particle_type, &mode, nullptr, nullptr));
        return mode != 0;
    }
    void set_transform(const Transform &local, const Transform *global = nullptr) {
        transform = local;
        has_global = global != nullptr;
        if (global) global_transform = *global;
    }
    void set_parent_velocity(Vec3 v);
    void set_modifier(EffectModifier m);
    std::function<void()> on_finished;  // observer of ParticleSpawnerFinished (core.rs:338-341)
    fw_spawner handle = -1;

  private:
    friend class ParticleSystemPlugin;
    fw_ctx *raw_();
    void check_(fw_status st);
    ParticleSystemPlugin *sys = nullptr;
    ParticleSpawner settings;
    Transform transform, global_transform;
    bool has_global = false;
};

class ParticleSystemPlugin {
  public:
    explicit ParticleSystemPlugin(int device = 0, uint32_t seed = 0, void *hip_stream = nullptr) {
        const fw_status st = fw_ctx_create(device, seed, hip_stream, &ctx_);
        if (st != FW_OK) throw Error(st, fw_last_error(nullptr));
    }
    ~ParticleSystemPlugin() {
        for (auto *d : spawners_) delete d;
        if (ctx_) fw_ctx_destroy(ctx_);
    }
    ParticleSystemPlugin(const ParticleSystemPlugin &) = delete;
    ParticleSystemPlugin &operator=(const ParticleSystemPlugin &) = delete;

    // commands.spawn((ParticleSpawner {..}, Transform))
    ParticleSpawnerData *spawn(const ParticleSpawner &s, const Transform &t = {}, uint32_t uid = UINT32_MAX) {
        std::vector<fw_particle_settings> ps(s.particle_settings.size());
        std::vector<fw_emission_settings> es(s.emission_settings.size());
        for (size_t i = 0; i < ps.size(); i++) {
            const ParticleSettings &p = s.particle_settings[i];
            fw_particle_settings &d = ps[i];
            d = fw_particle_settings{};
            d.lifetime = {p.lifetime.min, p.lifetime.max};
            d.scale_curve = {p.scale_curve.kind, (int32_t)p.scale_curve.values.size(),
                             p.scale_curve.times.empty() ? nullptr : p.scale_curve.times.data(), p.scale_curve.values.data()};
            d.initial_scale = {p.initial_scale.min, p.initial_scale.max};
            d.acceleration[0] = p.acceleration.x, d.acceleration[1] = p.acceleration.y, d.acceleration[2] = p.acceleration.z;
            d.angular_acceleration[0] = p.angular_acceleration.x, d.angular_acceleration[1] = p.angular_acceleration.y;
            d.angular_acceleration[2] = p.angular_acceleration.z;
            d.linear_drag = p.linear_drag, d.angular_drag = p.angular_drag;
            d.base_color = {p.base_color.kind, (int32_t)(p.base_color.rgba.size() / 4),
                            p.base_color.times.empty() ? nullptr : p.base_color.times.data(), p.base_color.rgba.data()};
            d.emissive_color = {p.emissive_color.kind, (int32_t)(p.emissive_color.rgba.size() / 4),
                                p.emissive_color.times.empty() ? nullptr : p.emissive_color.times.data(),
                                p.emissive_color.rgba.data()};
            d.pbr = p.pbr, d.report_destroyed = p.particles_destroyed ? 1 : 0, d.capacity = p.capacity;
            d.collision.enabled = p.has_collision_settings ? 1 : 0;
            d.collision.restitution = p.collision_settings.restitution, d.collision.friction = p.collision_settings.friction;
            d.collision.destroy_on_collision = p.collision_settings.destroy_on_collision ? 1 : 0;
            d.collision.filter_mask = p.collision_settings.filter_mask;
        }
        for (size_t i = 0; i < es.size(); i++) {
            const EmissionSettings &e = s.emission_settings[i];
            fw_emission_settings &d = es[i];
            d = fw_emission_settings{};
            d.particle_index = e.particle_index;
            d.pacing_kind = e.emission_pacing.kind, d.oneshot_count = e.emission_pacing.one_shot;
            d.count = e.emission_pacing.count, d.duration = e.emission_pacing.duration;
            d.offset_start = e.emission_pacing.offset_start, d.offset_end = e.emission_pacing.offset_end;
            d.mode = e.emission_mode.kind, d.target_particle_type = e.emission_mode.target_particle_type;
            d.shape_kind = e.emission_shape.kind, d.shape_radius = e.emission_shape.radius;
            d.shape_normal[0] = e.emission_shape.normal.x, d.shape_normal[1] = e.emission_shape.normal.y;
            d.shape_normal[2] = e.emission_shape.normal.z;
            auto rv = [](const RandVec3 &r) {
                fw_rand_vec3 o{};
                o.magnitude = {r.magnitude.min, r.magnitude.max};
                o.direction[0] = r.direction.x, o.direction[1] = r.direction.y, o.direction[2] = r.direction.z;
                o.spread = r.spread;
                return o;
            };
            d.initial_velocity = rv(e.initial_velocity);
            d.initial_velocity_radial = {e.initial_velocity_radial.min, e.initial_velocity_radial.max};
            d.inherit_parent_velocity = e.inherit_parent_velocity;
            d.initial_rotation[0] = e.initial_rotation.x, d.initial_rotation[1] = e.initial_rotation.y;
            d.initial_rotation[2] = e.initial_rotation.z, d.initial_rotation[3] = e.initial_rotation.w;
            d.initial_angular_velocity = rv(e.initial_angular_velocity);
        }
        fw_spawner_desc desc{ps.data(), (uint32_t)ps.size(), es.data(), (uint32_t)es.size(), s.starts_enabled ? 1 : 0,
                             uid == UINT32_MAX ? next_uid_ : uid};
        next_uid_ = desc.uid + 1;
        fw_spawner h = -1;
        check(fw_spawner_create(ctx_, &desc, &h));
        auto *d = new ParticleSpawnerData();
        d->handle = h, d->sys = this, d->settings = s, d->transform = t;
        spawners_.push_back(d);
        return d;
    }

    // the world particles collide with (stands in for avian's SpatialQuery; core.rs:581, 744-800)
    void set_colliders(const std::vector<Collider> &cs) {
        std::vector<fw_collider> v(cs.size());
        for (size_t i = 0; i < cs.size(); i++) {
            const Collider &c = cs[i];
            fw_collider &d = v[i];
            d = fw_collider{};
            d.kind = c.kind, d.layers = c.layers, d.radius = c.radius;
            d.position[0] = c.position.x, d.position[1] = c.position.y, d.position[2] = c.position.z;
            d.rotation[0] = c.rotation.x, d.rotation[1] = c.rotation.y, d.rotation[2] = c.rotation.z, d.rotation[3] = c.rotation.w;
            d.normal[0] = c.normal.x, d.normal[1] = c.normal.y, d.normal[2] = c.normal.z;
            d.half_extents[0] = c.half_extents.x, d.half_extents[1] = c.half_extents.y, d.half_extents[2] = c.half_extents.z;
        }
        check(fw_ctx_set_colliders(ctx_, v.data(), (uint32_t)v.size()));
    }

    // update_aabbs (render.rs:677-703) fused into the update: every frame leaves per-tile boxes, ParticleSpawnerData::aabb
    // folds them instead of re-reading the particles
    void track_aabbs(bool enable) { check(fw_ctx_track_aabbs(ctx_, enable ? 1 : 0)); }

    // one run of the chained systems (plugin.rs:46-60)
    void update(float dt) {
        // the transforms of all spawners in ONE call: spawn_particles walks every spawner of the query (core.rs:377)
        origin_h_.clear(), origin_t_.clear(), origin_r_.clear();
        for (auto *d : spawners_) {
            const Transform &t = (d->settings.spawn_transform_mode == SpawnTransformMode::Global && d->has_global)
                                     ? d->global_transform : d->transform;  // core.rs:432-435
            origin_h_.push_back(d->handle);
            origin_t_.insert(origin_t_.end(), {t.translation.x, t.translation.y, t.translation.z});
            origin_r_.insert(origin_r_.end(), {t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w});
        }
        check(fw_ctx_set_origins(ctx_, (uint32_t)origin_h_.size(), origin_h_.data(), origin_t_.data(), origin_r_.data()));
        check(fw_step(ctx_, dt));
        for (auto *d : spawners_) {
            for (size_t i = 0; i < d->settings.particle_settings.size(); i++)
                if (d->settings.particle_settings[i].particles_destroyed) {  // core.rs:660-667
                    auto dead = d->destroyed((uint32_t)i);
                    if (!dead.empty()) d->settings.particle_settings[i].particles_destroyed(dead);
                }
            if (d->on_finished) {
                int32_t fin = 0;
                check(fw_spawner_poll_finished(ctx_, d->handle, &fin));  // core.rs:674-688
                if (fin) d->on_finished();
            }
        }
    }
    void step(float dt) { check(fw_step(ctx_, dt)); }  // enqueue only: no transforms, no callbacks
    // the other per-frame inputs of MANY spawners in one FFI call each (ABI 5): sync_parent_velocity (core.rs:706-736),
    // propagate_particle_spawner_modifier (core.rs:690-703), queue_particles (core.rs:284-286) on a set of OnDemand spawners
    void set_parent_velocities(const std::vector<ParticleSpawnerData *> &ds, const std::vector<Vec3> &vs) {
        std::vector<fw_spawner> h;
        std::vector<float> v;
        for (size_t i = 0; i < ds.size(); i++) h.push_back(ds[i]->handle), v.insert(v.end(), {vs[i].x, vs[i].y, vs[i].z});
        check(fw_ctx_set_parent_velocities(ctx_, (uint32_t)h.size(), h.data(), v.data()));
    }
    void set_modifiers(const std::vector<ParticleSpawnerData *> &ds, const std::vector<EffectModifier> &ms) {
        std::vector<fw_spawner> h;
        std::vector<float> sc, sp;
        for (size_t i = 0; i < ds.size(); i++) h.push_back(ds[i]->handle), sc.push_back(ms[i].scale), sp.push_back(ms[i].speed);
        check(fw_ctx_set_modifiers(ctx_, (uint32_t)h.size(), h.data(), sc.data(), sp.data()));
    }
    void queue_particles(const std::vector<ParticleSpawnerData *> &ds, const std::vector<uint64_t> &counts) {
        std::vector<fw_spawner> h;
        for (auto *d : ds) h.push_back(d->handle);
        check(fw_ctx_queue(ctx_, (uint32_t)h.size(), h.data(), counts.data()));
    }
    void synchronize() { check(fw_ctx_synchronize(ctx_)); }
    uint64_t live_count() {
        uint64_t n = 0;
        check(fw_ctx_live_count(ctx_, &n));
        return n;
    }
    uint64_t updated_total() {
        uint64_t n = 0;
        check(fw_ctx_last_step_updated(ctx_, &n));
        return n;
    }
    fw_ctx *raw() { return ctx_; }
    void check(fw_status st) const {
        if (st != FW_OK) throw Error(st, fw_last_error(ctx_));
    }
    void check(int st) const { check((fw_status)st); }

  private:
    fw_ctx *ctx_ = nullptr;
    std::vector<ParticleSpawnerData *> spawners_;
    uint32_t next_uid_ = 0;
    std::vector<fw_spawner> origin_h_;  // scratch of update(): handles / translations / rotations of fw_ctx_set_origins
    std::vector<float> origin_t_, origin_r_;
};

inline fw_ctx *ParticleSpawnerData::raw_() { return sys->raw(); }
inline void ParticleSpawnerData::check_(fw_status st) { sys->check(st); }
inline void ParticleSpawnerData::queue_particles(uint64_t n) { sys->check(fw_spawner_queue(sys->raw(), handle, n)); }
inline bool ParticleSpawnerData::active() {
    int32_t a = 0;
    sys->check(fw_spawner_active(sys->raw(), handle, &a));
    return a != 0;
}
inline std::vector<uint32_t> ParticleSpawnerData::counts() {
    std::vector<uint32_t> c(settings.particle_settings.size());
    sys->check(fw_spawner_counts(sys->raw(), handle, c.data(), (uint32_t)c.size()));
    return c;
}
inline std::vector<fw_particle> ParticleSpawnerData::particles(uint32_t t) {
    uint64_t n = 0;
    sys->check(fw_spawner_read_particles(sys->raw(), handle, t, nullptr, 0, &n));
    std::vector<fw_particle> v(n);
    if (n) sys->check(fw_spawner_read_particles(sys->raw(), handle, t, v.data(), n, &n));
    return v;
}
inline std::vector<fw_particle> ParticleSpawnerData::destroyed(uint32_t t) {
    uint64_t n = 0;
    sys->check(fw_spawner_read_destroyed(sys->raw(), handle, t, nullptr, 0, &n));
    std::vector<fw_particle> v(n);
    if (n) sys->check(fw_spawner_read_destroyed(sys->raw(), handle, t, v.data(), n, &n));
    return v;
}
inline std::vector<fw_particle_instance> ParticleSpawnerData::instances(uint32_t t) {
    uint64_t n = 0;
    sys->check(fw_spawner_pack_instances(sys->raw(), handle, t, nullptr, 0, &n));
    std::vector<fw_particle_instance> v(n);
    if (n) sys->check(fw_spawner_pack_instances(sys->raw(), handle, t, v.data(), n, &n));
    return v;
}
inline void ParticleSpawnerData::attach_instances(void *device_buffer, uint64_t cap, uint32_t t) {
    sys->check(fw_spawner_attach_instances(sys->raw(), handle, t, device_buffer, cap));
}
inline bool ParticleSpawnerData::aabb(Vec3 &mn, Vec3 &mx) {
    float a[3], b[3];
    int32_t any = 0;
    sys->check(fw_spawner_aabb(sys->raw(), handle, a, b, &any));
    mn = {a[0], a[1], a[2]}, mx = {b[0], b[1], b[2]};
    return any != 0;
}
inline void ParticleSpawnerData::set_parent_velocity(Vec3 v) {
    const float a[3] = {v.x, v.y, v.z};
    sys->check(fw_spawner_set_parent_velocity(sys->raw(), handle, a));
}
inline void ParticleSpawnerData::set_modifier(EffectModifier m) {
    sys->check(fw_spawner_set_modifier(sys->raw(), handle, m.scale, m.speed));
}

}  // namespace firework
