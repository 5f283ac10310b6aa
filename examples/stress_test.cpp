// stress_test.cpp -- headless counterpart of the reference's examples/stress_test.rs (lines 92-129: one
// spawner, Circle emission, cone velocity, 5-key uneven gradient, rate 160 000 / s, lifetime 1 s), driven
// through the C++ host mirror (include/firework.hpp) over the C ABI.  Prints what the example's DebugInfo
// overlay shows (stress_test.rs:178-201): particle count and frames per second.
//
//   make -C examples        (g++ -std=c++17 -Iinclude ... -lfirework_hip)
//   ./examples/stress_test [rate] [frames] [collision]
// With a third argument the particle type gets the collision settings of examples/stress_test_collision.rs:110-115
// (restitution 0.6, friction 0.2) and the world holds that example's 8 x 1 x 8 cuboid base (:85-90) as a box collider.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "firework.hpp"

using namespace firework;

int main(int argc, char **argv) {
    const float rate = argc > 1 ? (float)atof(argv[1]) : 160000.0f;
    const int frames = argc > 2 ? atoi(argv[2]) : 600;
    const bool collision = argc > 3;
    const float PI = 3.14159265358979f;
    try {
        ParticleSystemPlugin app(0, /*seed*/ 0x00C0FFEE);
        ParticleSpawner sp;
        ParticleSettings &ps = sp.particle_settings[0];
        ps.lifetime = RandF32::constant(1.0f);
        ps.initial_scale = {0.02f, 0.08f};
        ps.scale_curve = FireworkCurve::constant(1.0f);
        ps.base_color = FireworkGradient::uneven_samples({{0.0f, {10, 7, 1, 1}},
                                                          {0.7f, {3, 1, 1, 1}},
                                                          {0.8f, {1, 0.3f, 0.3f, 1}},
                                                          {0.9f, {0.3f, 0.3f, 0.3f, 1}},
                                                          {1.0f, {0.1f, 0.1f, 0.1f, 0}}});
        ps.linear_drag = 0.1f;
        if (collision) {
            ps.has_collision_settings = true;
            ps.collision_settings = ParticleCollisionSettings{0.6f, 0.2f, false, 0xFFFFFFFFu};
            app.set_colliders({Collider::Box({0.0f, -0.5f, 0.0f}, {4.0f, 0.5f, 4.0f})});
        }
        EmissionSettings &es = sp.emission_settings[0];
        es.emission_pacing = EmissionPacing::rate(rate);
        es.emission_shape = EmissionShape::Circle({0, 1, 0}, 0.3f);
        es.inherit_parent_velocity = true;
        es.initial_velocity = {{0.0f, 10.0f}, {0, 1, 0}, 30.0f / 180.0f * PI};
        ParticleSpawnerData *data = app.spawn(sp, Transform{{0.0f, 0.1f, 0.0f}, {}});

        const float dt = 1.0f / 60.0f;
        app.update(dt);
        for (int i = 0; i < 90; i++) app.step(dt);  // fill
        app.synchronize();
        const auto t0 = std::chrono::steady_clock::now();
        const uint64_t u0 = app.updated_total();
        for (int i = 0; i < frames; i++) app.step(dt);
        app.synchronize();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const uint64_t updated = app.updated_total() - u0;
        std::printf("Particles: %u\nSystems: 1\nFPS (simulation only): %.0f\nparticles updated/s: %.3e\n",
                    data->counts()[0], frames / sec, updated / sec);
        Vec3 mn, mx;
        if (data->aabb(mn, mx))
            std::printf("aabb: [%.3f %.3f %.3f] .. [%.3f %.3f %.3f]\n", mn.x, mn.y, mn.z, mx.x, mx.y, mx.z);
    } catch (const Error &e) {
        std::fprintf(stderr, "firework error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}
