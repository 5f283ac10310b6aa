// mirror_check.cpp -- a scenario that touches most of the C++ host mirror (include/firework.hpp): three particle types,
// Global / OnDemand / Nested entries, every curve kind, modifier, transforms, parent velocity, destroyed-particle
// handler, collisions against an analytic world, fused AABB tracking.  Prints, every tenth frame, the live counts and an
// FNV-1a digest of every particle record; tests/test_cpp_host.py runs the same scenario through the Python mirror and
// expects the same lines: both mirrors marshal the reference's settings into the C ABI the same way.
//
//   make -C examples && ./examples/mirror_check
#include <cstdio>
#include <cstring>

#include "firework.hpp"

using namespace firework;

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main() {
    try {
        ParticleSystemPlugin app(0, /*seed*/ 0x00C0FFEE);
        app.track_aabbs(true);
        app.set_colliders({Collider::Plane({0.0f, -1.0f, 0.0f}, {0.0f, 1.0f, 0.0f}), Collider::Sphere({1.0f, 0.5f, 0.0f}, 0.75f, 2u),
                           Collider::Box({-2.0f, 0.0f, 0.0f}, {0.5f, 1.0f, 0.5f}, Quat{0.0f, 0.38268343f, 0.0f, 0.92387953f})});
        ParticleSpawner sp;
        sp.particle_settings.resize(3);
        uint64_t destroyed_seen = 0;
        {
            ParticleSettings &p = sp.particle_settings[0];  // sparks: one lifetime value, reports its dead
            p.lifetime = RandF32::constant(0.4f);
            p.initial_scale = {0.5f, 2.0f};
            p.scale_curve = FireworkCurve::even_samples({1.0f, 2.0f, 0.5f});
            p.base_color = FireworkGradient::uneven_samples({{0.0f, {10, 7, 1, 1}}, {0.7f, {3, 1, 1, 1}}, {1.0f, {0.1f, 0.1f, 0.1f, 0}}});
            p.linear_drag = 0.3f;
            p.particles_destroyed = [&](const std::vector<fw_particle> &dead) { destroyed_seen += dead.size(); };
        }
        {
            ParticleSettings &p = sp.particle_settings[1];  // smoke: children of the sparks, lifetime range
            p.lifetime = {0.2f, 0.6f};
            p.acceleration = {0.0f, 0.5f, 0.0f};
            p.scale_curve = FireworkCurve::uneven_samples({{0.0f, 1.0f}, {0.8f, 1.2f}, {1.0f, 0.0f}});
            p.emissive_color = FireworkGradient::even_samples({{4, 2, 0, 1}, {0, 0, 0, 1}});
            p.angular_drag = 0.1f;
            p.angular_acceleration = {0.1f, 0.0f, -0.2f};
        }
        {
            ParticleSettings &p = sp.particle_settings[2];  // pebbles: bounce in the collider world
            p.lifetime = {0.5f, 0.9f};
            p.has_collision_settings = true;
            p.collision_settings = ParticleCollisionSettings{0.6f, 0.2f, false, 3u};
            p.pbr = true;
        }
        sp.emission_settings.resize(4);
        {
            EmissionSettings &e = sp.emission_settings[0];
            e.particle_index = 0;
            e.emission_pacing = EmissionPacing::rate(5000.0f);
            e.emission_shape = EmissionShape::Sphere(0.5f);
            e.initial_velocity = {{1.0f, 6.0f}, {0.0f, 1.0f, 0.0f}, 0.0f};
            e.initial_velocity_radial = {1.0f, 2.0f};
        }
        {
            EmissionSettings &e = sp.emission_settings[1];
            e.particle_index = 1;
            e.emission_pacing = EmissionPacing::CountOverDuration(8.0f, 1.0f, 0.1f, 0.9f);
            e.emission_mode = EmissionMode::Nested(0);
            e.inherit_parent_velocity = false;
        }
        {
            EmissionSettings &e = sp.emission_settings[2];
            e.particle_index = 2;
            e.emission_pacing = EmissionPacing::OnDemand();
            e.emission_shape = EmissionShape::Circle({0.0f, 0.0f, 1.0f}, 2.0f);
            e.initial_velocity = {{0.0f, 3.0f}, {0.0f, -1.0f, 0.0f}, 0.0f};
            e.initial_rotation = {0.0f, 0.38941834f, 0.0f, 0.92106099f};
        }
        {
            EmissionSettings &e = sp.emission_settings[3];
            e.particle_index = 2;
            e.emission_pacing = EmissionPacing::OneShot(700);
        }
        ParticleSpawnerData *d = app.spawn(sp, Transform{{0.0f, 1.0f, 0.0f}, {}}, 42u);
        d->set_modifier(EffectModifier{2.0f, 0.5f});
        d->set_parent_velocity({0.5f, 0.0f, -0.25f});
        const float dt = 1.0f / 60.0f;
        for (int fr = 0; fr < 60; fr++) {
            if (fr == 0 || fr == 7 || fr == 8 || fr == 31) d->queue_particles(500 + 10 * fr);
            if (fr == 20) d->set_transform(Transform{{1.0f, 2.0f, 3.0f}, Quat{0.0f, 0.0f, 0.38268343f, 0.92387953f}});
            app.update(dt);
            if (fr % 10 != 9) continue;
            const auto c = d->counts();
            std::printf("frame %d counts %u %u %u", fr, c[0], c[1], c[2]);
            for (uint32_t t = 0; t < 3; t++) {
                const auto ps = d->particles(t);
                std::printf(" %016llx", (unsigned long long)fnv(ps.data(), ps.size() * sizeof(fw_particle)));
            }
            Vec3 mn, mx;
            const bool any = d->aabb(mn, mx);
            const float box[6] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z};
            std::printf(" aabb %d %016llx active %d\n", any ? 1 : 0, (unsigned long long)fnv(box, sizeof box), d->active() ? 1 : 0);
        }
        std::printf("destroyed reported %llu\n", (unsigned long long)destroyed_seen);
    } catch (const Error &e) {
        std::fprintf(stderr, "firework error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}
