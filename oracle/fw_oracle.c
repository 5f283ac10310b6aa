/*
 * fw_oracle.c -- CPU ORACLE (test infrastructure only; see fw_oracle.h header).
 *
 * Build: gcc -O2 -std=c11 -fPIC -shared -ffp-contract=off -fno-fast-math
 * (Rust never contracts a*b+c into an FMA; counts depend on that.)
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Nothing here is used by the shipped library.
 */
#include "fw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FWO_PI 3.14159265358979323846f /* std::f32::consts::PI */
#define FWO_F32_MIN (-3.40282347e+38f) /* f32::MIN (core.rs:467) */

/* ------------------------------------------------------------------------- */
/* Rust scalar semantics                                                      */
/* ------------------------------------------------------------------------- */

/* f32::rem_euclid (core.rs:412-414 call site) */
float fwo_rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return (r < 0.0f) ? r + fabsf(b) : r;
}

/* f32::div_euclid (core.rs:569 call site) */
float fwo_div_euclid(float a, float b) {
    float q = truncf(a / b);
    if (fmodf(a, b) < 0.0f) return (b > 0.0f) ? q - 1.0f : q + 1.0f;
    return q;
}

/* `x as usize`: saturating, NaN -> 0 (core.rs:570) */
static uint64_t f32_as_usize(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)x;
}

static float f32_clamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

/* core.rs:553-575 */
uint64_t fwo_compute_emission_count(float time_passed_in_cycle, float last_emission, float cycle_duration,
                                    float offset_start, float offset_end, float particles_per_cycle,
                                    float *next_last_emission) {
    float percent_passed = time_passed_in_cycle / cycle_duration;
    float last_emission_percent = last_emission / cycle_duration;
    float percent_passed_since_emission = fminf(percent_passed, offset_end) - fmaxf(last_emission_percent, offset_start);
    float percent_between_emissions = (offset_end - offset_start) / particles_per_cycle;
    float times_needed_to_emit = fwo_div_euclid(percent_passed_since_emission, percent_between_emissions);
    uint64_t n = f32_as_usize(times_needed_to_emit);
    float t0 = times_needed_to_emit * percent_between_emissions;
    float next_last_emission_percent = fmaxf(last_emission_percent, offset_start) + t0;
    if (next_last_emission) *next_last_emission = next_last_emission_percent * cycle_duration;
    return n;
}

/* ------------------------------------------------------------------------- */
/* bevy_math 0.19 curve cores (curve.rs:26-32,113,157 call sites)             */
/* ------------------------------------------------------------------------- */

/* cores::even_interp on domain [0,1]: returns 0 = exact/tail(idx in *lo), 1 = between */
static int even_interp(int n, float t, int *lo, float *s) {
    int subdivs = n - 1;
    float step = 1.0f / (float)subdivs; /* domain.length() / subdivs */
    float t_shifted = t - 0.0f;
    float steps_taken = t_shifted / step;
    if (steps_taken <= 0.0f) {
        *lo = 0;
        return 0;
    } else if (steps_taken >= (float)subdivs) {
        *lo = n - 1;
        return 0;
    } else {
        float fl = floorf(steps_taken);
        *lo = (int)f32_as_usize(fl); /* NaN -> 0 like `as usize` */
        *s = steps_taken - truncf(steps_taken); /* f32::fract */
        return 1;
    }
}

/* cores::uneven_interp: binary_search_by(partial_cmp) over sorted, deduped times */
static int uneven_interp(const float *times, int n, float t, int *lo, float *s) {
    int idx = 0; /* number of times < t == Err(insertion point) */
    while (idx < n && times[idx] < t) idx++;
    if (idx < n && times[idx] == t) {
        *lo = idx;
        return 0;
    }
    if (idx == 0) {
        *lo = 0;
        return 0;
    }
    if (idx >= n) {
        *lo = n - 1;
        return 0;
    }
    float t_lower = times[idx - 1], t_upper = times[idx];
    *s = (t - t_lower) / (t_upper - t_lower);
    *lo = idx - 1;
    return 1;
}

/* FireworkCurve<f32>::sample_clamped: Curve default = clamp to domain, then
 * sample_unchecked (curve.rs:26-32).  f32 interpolation: bevy_math 0.19 implements
 * StableInterpolate for every NormedVectorSpace as `self.lerp(*other, t)`, and inside that generic
 * impl `lerp` is VectorSpace::lerp = self * (1. - t) + rhs * t (NOT glam's FloatExt::lerp
 * a + (b - a) * t, which is not a candidate for a generic V).  Same form as Mix below. */
float fwo_curve_sample_clamped(const fwo_curve *c, float t) {
    int lo;
    float s = 0.0f;
    if (c->kind == FWO_CURVE_CONSTANT || c->n == 1) return c->values[0];
    if (c->kind == FWO_CURVE_EVEN) {
        t = f32_clamp(t, 0.0f, 1.0f);
        if (!even_interp(c->n, t, &lo, &s)) return c->values[lo];
    } else {
        t = f32_clamp(t, c->times[0], c->times[c->n - 1]);
        if (!uneven_interp(c->times, c->n, t, &lo, &s)) return c->values[lo];
    }
    float a = c->values[lo], b = c->values[lo + 1];
    return a * (1.0f - s) + b * s;
}

/* FireworkGradient<LinearRgba>::sample_clamped (curve.rs:111-114,156-158);
 * bevy_color Mix: a * (1 - f) + b * f per channel. */
void fwo_gradient_sample_clamped(const fwo_gradient *g, float t, float out[4]) {
    int lo, between;
    float s = 0.0f;
    if (g->kind == FWO_CURVE_CONSTANT || g->n == 1) {
        memcpy(out, g->rgba, 4 * sizeof(float));
        return;
    }
    if (g->kind == FWO_CURVE_EVEN)
        between = even_interp(g->n, t, &lo, &s);
    else
        between = uneven_interp(g->times, g->n, t, &lo, &s);
    const float *a = g->rgba + 4 * lo;
    if (!between) {
        memcpy(out, a, 4 * sizeof(float));
        return;
    }
    const float *b = a + 4;
    float nf = 1.0f - s;
    for (int k = 0; k < 4; k++) out[k] = a[k] * nf + b[k] * s;
}

/* ------------------------------------------------------------------------- */
/* glam 0.32.1 scalar Vec3 / Quat                                             */
/* ------------------------------------------------------------------------- */

static float v3_dot(const float a[3], const float b[3]) { return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]); }
static void v3_cross(const float a[3], const float b[3], float o[3]) {
    float x = a[1] * b[2] - b[1] * a[2];
    float y = a[2] * b[0] - b[2] * a[0];
    float z = a[0] * b[1] - b[0] * a[1];
    o[0] = x, o[1] = y, o[2] = z;
}
/* Vec3::normalize_or_zero (core.rs:442,512) */
static void v3_normalize_or_zero(const float a[3], float o[3]) {
    float rcp = 1.0f / sqrtf(v3_dot(a, a));
    if (isfinite(rcp) && rcp > 0.0f) {
        o[0] = a[0] * rcp, o[1] = a[1] * rcp, o[2] = a[2] * rcp;
    } else {
        o[0] = o[1] = o[2] = 0.0f;
    }
}

/* Quat::from_axis_angle */
static void quat_from_axis_angle(const float axis[3], float angle, float o[4]) {
    float h = angle * 0.5f;
    float s = sinf(h), c = cosf(h);
    o[0] = axis[0] * s, o[1] = axis[1] * s, o[2] = axis[2] * s, o[3] = c;
}

/* Quat::from_scaled_axis (core.rs:645-647): identity when |v| == 0 */
void fwo_quat_from_scaled_axis(const float v[3], float o[4]) {
    float len = sqrtf(v3_dot(v, v));
    if (len == 0.0f) {
        o[0] = o[1] = o[2] = 0.0f, o[3] = 1.0f;
        return;
    }
    float axis[3] = {v[0] / len, v[1] / len, v[2] / len};
    quat_from_axis_angle(axis, len, o);
}

/* Quat * Quat, scalar Hamilton product (core.rs:645-647) */
void fwo_quat_mul(const float a[4], const float b[4], float o[4]) {
    float x0 = a[0], y0 = a[1], z0 = a[2], w0 = a[3];
    float x1 = b[0], y1 = b[1], z1 = b[2], w1 = b[3];
    float x = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
    float y = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
    float z = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
    float w = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
    o[0] = x, o[1] = y, o[2] = z, o[3] = w;
}

/* Quat * Vec3, scalar path (core.rs:441,510) */
void fwo_quat_mul_vec3(const float q[4], const float v[3], float o[3]) {
    float w = q[3];
    float b[3] = {q[0], q[1], q[2]};
    float b2 = v3_dot(b, b);
    float k0 = w * w - b2;
    float k1 = v3_dot(v, b) * 2.0f;
    float c[3];
    v3_cross(b, v, c);
    float k2 = w * 2.0f;
    for (int i = 0; i < 3; i++) o[i] = (v[i] * k0 + b[i] * k1) + c[i] * k2;
}

/* Vec3::any_orthonormal_vector */
static void v3_any_orthonormal(const float v[3], float o[3]) {
    float sign = copysignf(1.0f, v[2]);
    float a = -1.0f / (sign + v[2]);
    float b = v[0] * v[1] * a;
    o[0] = b, o[1] = sign + v[1] * v[1] * a, o[2] = -v[1];
}

/* Quat::from_rotation_arc (emission_shape.rs:34) */
void fwo_quat_from_rotation_arc(const float from[3], const float to[3], float o[4]) {
    const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-7f;
    float d = v3_dot(from, to);
    if (d > one_minus_eps) {
        o[0] = o[1] = o[2] = 0.0f, o[3] = 1.0f;
    } else if (d < -one_minus_eps) {
        float ax[3];
        v3_any_orthonormal(from, ax);
        quat_from_axis_angle(ax, FWO_PI, o);
    } else {
        float c[3];
        v3_cross(from, to, c);
        float q[4] = {c[0], c[1], c[2], 1.0f + d};
        float len = sqrtf((q[0] * q[0]) + (q[1] * q[1]) + (q[2] * q[2]) + (q[3] * q[3]));
        float inv = 1.0f / len;
        for (int i = 0; i < 4; i++) o[i] = q[i] * inv;
    }
}

/* ------------------------------------------------------------------------- */
/* RNG: Philox4x32-10 counter stream (replaces the reference's unseeded        */
/* thread-local rand::random; draw ORDER follows core.rs:438-466)             */
/* ------------------------------------------------------------------------- */

void fwo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* rand 0.9 StandardUniform for f32: 24 high bits -> [0,1) */
static float u32_to_unit_f32(uint32_t x) { return (float)(x >> 8) * 5.9604645e-8f; /* 2^-24 */ }

/*
 * Stream definition (shared spec with the device kernels, implemented
 * independently there): key = (seed, spawner_uid); counter = (serial_lo,
 * serial_hi, emission_index, block) for block 0..2; `serial` counts particles
 * ever spawned by that emission entry.  Uniform slots, in the reference's
 * evaluation order for one particle (core.rs:438-466):
 *   u0..u2  emission_shape.generate_point()        (emission_shape.rs:22-26,33)
 *   u3..u5  initial_velocity.generate()            (angle, radius, magnitude)
 *   u6      initial_velocity_radial.generate()
 *   u7      initial_scale.generate()
 *   u8      lifetime.generate()
 *   u9..u11 initial_angular_velocity.generate()    (angle, radius, magnitude)
 * Slots are fixed (a Point shape simply does not consume u0..u2) so a
 * particle's attributes do not depend on launch geometry or on other entries.
 */
void fwo_spawn_uniforms(uint32_t seed, uint32_t uid, uint32_t emission_index, uint64_t serial, float u[12]) {
    uint32_t key[2] = {seed, uid};
    for (uint32_t b = 0; b < 3; b++) {
        uint32_t ctr[4] = {(uint32_t)serial, (uint32_t)(serial >> 32), emission_index, b};
        uint32_t o[4];
        fwo_philox4x32_10(ctr, key, o);
        for (int k = 0; k < 4; k++) u[4 * b + k] = u32_to_unit_f32(o[k]);
    }
}

/* bevy_utilitarian RandF32::generate: random::<f32>() * (max - min) + min */
static float randf32_generate(const fwo_randf32 *r, float u) { return u * (r->max - r->min) + r->min; }

/* bevy_utilitarian RandVec3::generate (our reading of the published crate,
 * unpinned): direction perturbed inside a cone of half-angle `spread`, times
 * magnitude.generate(). */
void fwo_randvec3_generate(const fwo_randvec3 *r, float u_angle, float u_radius, float u_mag, float out[3]) {
    float dir[3];
    if (r->spread > 0.0f) {
        float spread_angle = u_angle * 2.0f * FWO_PI;
        float spread_radius = u_radius * r->spread;
        float sr = sinf(spread_radius), cr = cosf(spread_radius);
        float local[3] = {sr * cosf(spread_angle), cr, sr * sinf(spread_angle)};
        const float y[3] = {0.0f, 1.0f, 0.0f};
        float q[4];
        fwo_quat_from_rotation_arc(y, r->direction, q);
        fwo_quat_mul_vec3(q, local, dir);
    } else {
        memcpy(dir, r->direction, sizeof dir);
    }
    float m = randf32_generate(&r->magnitude, u_mag);
    out[0] = dir[0] * m, out[1] = dir[1] * m, out[2] = dir[2] * m;
}

/* EmissionShape::generate_point (emission_shape.rs:18-39) */
void fwo_shape_generate(const fwo_emission_settings *e, const float u[3], float out[3]) {
    if (e->shape_kind == FWO_SHAPE_SPHERE) {
        float pitch = u[0] * 2.0f * FWO_PI, yaw = u[1] * FWO_PI, r = u[2];
        /* bevy_utilitarian PitchYaw::to_unit_vec (our reading, unpinned) */
        float cp = cosf(pitch), sp = sinf(pitch);
        float unit[3] = {cp * sinf(yaw), sp, cp * cosf(yaw)};
        for (int i = 0; i < 3; i++) out[i] = unit[i] * r * e->shape_radius;
    } else if (e->shape_kind == FWO_SHAPE_CIRCLE) {
        float ang = u[0] * 2.0f * FWO_PI, r = u[1];
        const float y[3] = {0.0f, 1.0f, 0.0f};
        float q1[4], q2[4], q[4];
        fwo_quat_from_rotation_arc(y, e->shape_normal, q1);
        float h = ang * 0.5f; /* Quat::from_rotation_y */
        q2[0] = 0.0f, q2[1] = sinf(h), q2[2] = 0.0f, q2[3] = cosf(h);
        fwo_quat_mul(q1, q2, q);
        float v[3] = {r * e->shape_radius, 0.0f, 0.0f};
        fwo_quat_mul_vec3(q, v, out);
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
}

/* ------------------------------------------------------------------------- */
/* particle_collision (core.rs:744-800) with an analytic ray cast               */
/* ------------------------------------------------------------------------- */

static float v3_length(const float a[3]) { return sqrtf(v3_dot(a, a)); }
/* glam Vec3::normalize = self * length_recip() */
static void v3_normalize(const float a[3], float o[3]) {
    float r = 1.0f / v3_length(a);
    o[0] = a[0] * r, o[1] = a[1] * r, o[2] = a[2] * r;
}
/* glam Vec3::project_onto(rhs) = rhs * self.dot(rhs) * rhs.dot(rhs).recip() */
static void v3_project_onto(const float a[3], const float rhs[3], float o[3]) {
    float rcp = 1.0f / v3_dot(rhs, rhs);
    float d = v3_dot(a, rhs);
    for (int i = 0; i < 3; i++) o[i] = (rhs[i] * d) * rcp;
}

/* one collider, `solid = true` semantics (include/firework_hip.h: fw_collider): inside -> distance 0, zero normal */
static int ray_collider(const fwo_collider *c, const float o[3], const float d[3], float max_distance, float *dist,
                        float normal[3]) {
    if (c->kind == FWO_COLLIDER_PLANE) {
        float dpos[3] = {c->position[0] - o[0], c->position[1] - o[1], c->position[2] - o[2]};
        float dot_normal_dpos = v3_dot(c->normal, dpos);
        if (dot_normal_dpos > 0.0f) {
            *dist = 0.0f, normal[0] = normal[1] = normal[2] = 0.0f;
            return 1;
        }
        float t = dot_normal_dpos / v3_dot(c->normal, d);
        if (t >= 0.0f && t <= max_distance) {
            *dist = t;
            memcpy(normal, c->normal, 3 * sizeof(float));
            return 1;
        }
        return 0;
    }
    if (c->kind == FWO_COLLIDER_SPHERE) {
        float dc[3] = {o[0] - c->position[0], o[1] - c->position[1], o[2] - c->position[2]};
        float a = v3_dot(d, d), b = v3_dot(dc, d);
        float cc = v3_dot(dc, dc) - c->radius * c->radius;
        if (cc <= 0.0f) {
            *dist = 0.0f, normal[0] = normal[1] = normal[2] = 0.0f;
            return 1;
        }
        if (b > 0.0f) return 0;
        float delta = b * b - a * cc;
        if (!(delta >= 0.0f)) return 0;
        float t = (-b - sqrtf(delta)) / a;
        if (!(t >= 0.0f && t <= max_distance)) return 0;
        float p[3];
        for (int i = 0; i < 3; i++) p[i] = (o[i] + d[i] * t) - c->position[i];
        v3_normalize(p, normal);
        *dist = t;
        return 1;
    }
    float qi[4] = {-c->rotation[0], -c->rotation[1], -c->rotation[2], c->rotation[3]};
    float rel[3] = {o[0] - c->position[0], o[1] - c->position[1], o[2] - c->position[2]}, ol[3], dl[3];
    fwo_quat_mul_vec3(qi, rel, ol);
    fwo_quat_mul_vec3(qi, d, dl);
    if (c->kind == FWO_COLLIDER_CYLINDER) {
        /* avian Collider::cylinder(radius, height) (examples/textures.rs:195): axis = local Y; the slab |y| <= half height
         * intersected with the infinite cylinder x^2 + z^2 <= r^2 */
        float hh = c->half_extents[1], rr = c->radius * c->radius;
        float c2 = (ol[0] * ol[0] + ol[2] * ol[2]) - rr;
        if (fabsf(ol[1]) <= hh && c2 <= 0.0f) {
            *dist = 0.0f, normal[0] = normal[1] = normal[2] = 0.0f;
            return 1;
        }
        float tnear = -INFINITY, tfar = INFINITY, sign = 0.0f;
        int side = 0; /* 0: a cap, 1: the lateral surface */
        if (dl[1] == 0.0f) {
            if (fabsf(ol[1]) > hh) return 0;
        } else {
            float inv = 1.0f / dl[1];
            float t1 = (-hh - ol[1]) * inv, t2 = (hh - ol[1]) * inv, sg = -1.0f;
            if (t1 > t2) {
                float tmp = t1;
                t1 = t2, t2 = tmp, sg = 1.0f;
            }
            if (t1 > tnear) tnear = t1, side = 0, sign = sg;
            if (t2 < tfar) tfar = t2;
            if (tnear > tfar) return 0;
        }
        float a = dl[0] * dl[0] + dl[2] * dl[2], b = ol[0] * dl[0] + ol[2] * dl[2];
        if (a == 0.0f) {
            if (c2 > 0.0f) return 0;
        } else {
            float disc = b * b - a * c2;
            if (!(disc >= 0.0f)) return 0;
            float sq = sqrtf(disc);
            float t1 = (-b - sq) / a, t2 = (-b + sq) / a;
            if (t1 > tnear) tnear = t1, side = 1;
            if (t2 < tfar) tfar = t2;
            if (tnear > tfar) return 0;
        }
        if (!(tnear >= 0.0f && tnear <= max_distance)) return 0;
        float nl[3] = {0.0f, sign, 0.0f};
        if (side == 1) {
            float p[3] = {ol[0] + dl[0] * tnear, 0.0f, ol[2] + dl[2] * tnear};
            v3_normalize(p, nl);
        }
        fwo_quat_mul_vec3(c->rotation, nl, normal);
        *dist = tnear;
        return 1;
    }
    if (c->kind == FWO_COLLIDER_CONE) {
        /* avian Collider::cone(radius, height) (examples/textures.rs:211): base disc at local y = -h/2, apex at y = +h/2.
         * w = p - apex; the solid is  w.y <= 0,  y >= -h/2,  w.x^2 + w.z^2 <= k^2 w.y^2  with k = radius / height */
        float hh = c->half_extents[1], rr = c->radius * c->radius;
        float k = c->radius / (hh + hh), k2 = k * k;
        float wy = ol[1] - hh;
        float cq = (ol[0] * ol[0] + ol[2] * ol[2]) - k2 * (wy * wy);
        if (ol[1] >= -hh && wy <= 0.0f && cq <= 0.0f) {
            *dist = 0.0f, normal[0] = normal[1] = normal[2] = 0.0f;
            return 1;
        }
        float best = INFINITY;
        int side = -1; /* 0: the base disc, 1: the lateral surface */
        if (dl[1] > 0.0f && ol[1] < -hh) {
            float tb = (-hh - ol[1]) / dl[1];
            float px = ol[0] + dl[0] * tb, pz = ol[2] + dl[2] * tb;
            if (px * px + pz * pz <= rr) best = tb, side = 0;
        }
        float a = (dl[0] * dl[0] + dl[2] * dl[2]) - k2 * (dl[1] * dl[1]);
        float b = (ol[0] * dl[0] + ol[2] * dl[2]) - k2 * (wy * dl[1]);
        float ta = INFINITY, tb2 = INFINITY;
        if (a == 0.0f) {
            if (b != 0.0f) ta = -cq / (b + b);
        } else {
            float disc = b * b - a * cq;
            if (disc >= 0.0f) {
                float sq = sqrtf(disc);
                ta = (-b - sq) / a, tb2 = (-b + sq) / a;
            }
        }
        {
            float ya = ol[1] + dl[1] * ta, yb = ol[1] + dl[1] * tb2;
            if (ta >= 0.0f && ta < INFINITY && ya >= -hh && ya <= hh && ta < best) best = ta, side = 1;
            if (tb2 >= 0.0f && tb2 < INFINITY && yb >= -hh && yb <= hh && tb2 < best) best = tb2, side = 1;
        }
        if (side < 0 || !(best <= max_distance)) return 0;
        float nl[3] = {0.0f, -1.0f, 0.0f};
        if (side == 1) {
            float w[3] = {ol[0] + dl[0] * best, (ol[1] + dl[1] * best) - hh, ol[2] + dl[2] * best};
            float g[3] = {w[0], -(k2 * w[1]), w[2]};
            if (g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f) nl[0] = 0.0f, nl[1] = 1.0f, nl[2] = 0.0f;
            else v3_normalize(g, nl);
        }
        fwo_quat_mul_vec3(c->rotation, nl, normal);
        *dist = best;
        return 1;
    }
    /* BOX: slab test in the box frame */
    int inside = 1;
    for (int i = 0; i < 3; i++) inside = inside && fabsf(ol[i]) <= c->half_extents[i];
    if (inside) {
        *dist = 0.0f, normal[0] = normal[1] = normal[2] = 0.0f;
        return 1;
    }
    float tnear = -INFINITY, tfar = INFINITY, sign = 0.0f;
    int axis = 0;
    for (int i = 0; i < 3; i++) {
        float h = c->half_extents[i];
        if (dl[i] == 0.0f) {
            if (fabsf(ol[i]) > h) return 0;
            continue;
        }
        float inv = 1.0f / dl[i];
        float t1 = (-h - ol[i]) * inv, t2 = (h - ol[i]) * inv, sg = -1.0f;
        if (t1 > t2) {
            float tmp = t1;
            t1 = t2, t2 = tmp, sg = 1.0f;
        }
        if (t1 > tnear) tnear = t1, axis = i, sign = sg;
        if (t2 < tfar) tfar = t2;
        if (tnear > tfar) return 0;
    }
    if (!(tnear >= 0.0f && tnear <= max_distance)) return 0;
    float nl[3] = {0.0f, 0.0f, 0.0f};
    nl[axis] = sign;
    fwo_quat_mul_vec3(c->rotation, nl, normal);
    *dist = tnear;
    return 1;
}

/* SpatialQuery::cast_ray(origin, dir, max_distance, true, filter): the nearest hit */
static int cast_ray(const fwo_collider *cs, int n, uint32_t mask, const float o[3], const float d[3], float max_distance,
                    float *dist, float normal[3]) {
    int any = 0;
    for (int i = 0; i < n; i++) {
        if (!(cs[i].layers & mask)) continue;
        float t, nn[3];
        if (ray_collider(&cs[i], o, d, max_distance, &t, nn) && (!any || t < *dist)) {
            *dist = t;
            memcpy(normal, nn, sizeof nn);
            any = 1;
        }
    }
    return any;
}

/* core.rs:744-800 */
int32_t fwo_particle_collision(float pos[3], float vel[3], float delta, float restitution, float friction,
                               int32_t destroy_on_collision, uint32_t filter_mask, const fwo_collider *colliders,
                               int32_t n) {
    const float orig_delta = delta;
    int n_steps = 0, should_destroy = 0;
    while (delta > 0.0f && n_steps < 4) {
        float len = v3_length(vel), dir[3] = {0.0f, 1.0f, 0.0f}; /* Dir3::try_from(vel) else Dir3::Y */
        if (isfinite(len) && len > 0.0f)
            for (int i = 0; i < 3; i++) dir[i] = vel[i] / len;
        float dist, normal[3];
        if (cast_ray(colliders, n, filter_mask, pos, dir, v3_length(vel) * delta, &dist, normal)) {
            if (dist == 0.0f) {
                if (normal[0] == 0.0f && normal[1] == 0.0f && normal[2] == 0.0f) {
                    if (vel[0] != 0.0f || vel[1] != 0.0f || vel[2] != 0.0f) {
                        v3_normalize(vel, normal);
                    } else {
                        normal[0] = 0.0f, normal[1] = 1.0f, normal[2] = 0.0f;
                    }
                }
                float k = fmaxf(v3_length(vel), 1.0f);
                for (int i = 0; i < 3; i++) pos[i] += (k * normal[i]) * delta;
            } else {
                float nv[3], rej[3], proj[3], nrej[3];
                v3_normalize_or_zero(vel, nv);
                for (int i = 0; i < 3; i++) pos[i] += nv[i] * dist;
                v3_project_onto(vel, normal, proj);
                for (int i = 0; i < 3; i++) rej[i] = vel[i] - proj[i]; /* reject_from */
                float friction_dv = fminf(v3_length(proj), v3_length(rej)) * friction;
                v3_normalize_or_zero(rej, nrej);
                for (int i = 0; i < 3; i++) vel[i] = (rej[i] - friction_dv * nrej[i]) - restitution * proj[i];
                for (int i = 0; i < 3; i++) pos[i] += normal[i] * 0.0001f;
                delta = f32_clamp(delta - dist, 0.0f, orig_delta);
            }
            should_destroy = destroy_on_collision;
            if (should_destroy) return 1;
        } else {
            for (int i = 0; i < 3; i++) pos[i] += vel[i] * delta;
            delta = 0.0f;
        }
        n_steps++;
    }
    return should_destroy;
}

/* ------------------------------------------------------------------------- */
/* Spawner state: AoS, one heap vector per particle, like the reference        */
/* ------------------------------------------------------------------------- */

typedef struct {
    float position[3], velocity[3], rotation[4], angular_velocity[3];
    float initial_scale, scale, age, lifetime;
    float base_color[4], emissive_color[4];
    int32_t pbr;
    float *last_emitted_age; /* Vec<f32>, len = n_es (core.rs:320) */
} particle;

typedef struct {
    particle *p;
    size_t n, cap;
} pvec;

typedef struct { /* EmissionData core.rs:261-267 */
    float last_emission, time_passed_in_cycle;
    int enabled, emits_on_other_particles;
    uint64_t serial; /* RNG stream position (oracle-defined) */
} emission_data;

struct fwo_spawner {
    int32_t n_ps, n_es;
    fwo_particle_settings *ps;
    fwo_emission_settings *es;
    int starts_enabled;
    uint32_t seed, uid;
    /* ParticleSpawnerData core.rs:269-281 */
    int initialized, finished_notified;
    pvec *particles;
    pvec *destroyed;
    emission_data *emission;
    float parent_velocity[3];
    uint64_t manual_queued_count;
    /* per-frame inputs the ECS would provide */
    float origin_translation[3], origin_rotation[4];
    float modifier_scale, modifier_speed;
    fwo_collider *colliders; /* the SpatialQuery world (analytic stand-in) */
    int32_t n_colliders;
};

static void pvec_push(pvec *v, const particle *p) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 4; /* Rust Vec growth */
        v->p = (particle *)realloc(v->p, nc * sizeof(particle));
        v->cap = nc;
    }
    v->p[v->n++] = *p;
}
static void pvec_free(pvec *v) {
    for (size_t i = 0; i < v->n; i++) free(v->p[i].last_emitted_age);
    free(v->p);
    v->p = NULL, v->n = v->cap = 0;
}
static float *lea_new(int n, float fill) {
    if (n <= 0) return NULL;
    float *a = (float *)malloc((size_t)n * sizeof(float));
    for (int i = 0; i < n; i++) a[i] = fill;
    return a;
}
static float *lea_clone(const float *src, int n) {
    if (n <= 0) return NULL;
    float *a = (float *)malloc((size_t)n * sizeof(float));
    memcpy(a, src, (size_t)n * sizeof(float));
    return a;
}

static float *dup_f32(const float *src, size_t n) {
    if (!src || !n) return NULL;
    float *d = (float *)malloc(n * sizeof(float));
    memcpy(d, src, n * sizeof(float));
    return d;
}

/* UnevenCore::new: drop non-finite times, stable sort by time, dedup keeping
 * the first of each run (bevy_math cores). Returns new n. */
int32_t fwo_uneven_normalize(float *times, float *vals, int32_t n, int32_t stride);
static int uneven_normalize(float *times, float *vals, int n, int stride) { return fwo_uneven_normalize(times, vals, n, stride); }
int32_t fwo_uneven_normalize(float *times, float *vals, int32_t n, int32_t stride) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (isfinite(times[i])) {
            times[m] = times[i];
            memmove(vals + (size_t)m * stride, vals + (size_t)i * stride, (size_t)stride * sizeof(float));
            m++;
        }
    }
    for (int i = 1; i < m; i++) { /* insertion sort = stable */
        float t = times[i], tmp[4];
        memcpy(tmp, vals + (size_t)i * stride, (size_t)stride * sizeof(float));
        int j = i - 1;
        while (j >= 0 && times[j] > t) {
            times[j + 1] = times[j];
            memcpy(vals + (size_t)(j + 1) * stride, vals + (size_t)j * stride, (size_t)stride * sizeof(float));
            j--;
        }
        times[j + 1] = t;
        memcpy(vals + (size_t)(j + 1) * stride, tmp, (size_t)stride * sizeof(float));
    }
    int k = 0;
    for (int i = 0; i < m; i++) {
        if (k > 0 && times[k - 1] == times[i]) continue;
        times[k] = times[i];
        memmove(vals + (size_t)k * stride, vals + (size_t)i * stride, (size_t)stride * sizeof(float));
        k++;
    }
    return k;
}

fwo_spawner *fwo_spawner_create(const fwo_particle_settings *ps, int32_t n_ps, const fwo_emission_settings *es,
                                int32_t n_es, int32_t starts_enabled, uint32_t seed, uint32_t uid) {
    /* mirror the reference's panics as NULL: zero-key curves (curve.rs:45,61,211,227),
     * out-of-range indices (core.rs:392,453,488) */
    for (int i = 0; i < n_ps; i++)
        if (ps[i].scale_curve.n < 1 || ps[i].base_color.n < 1 || ps[i].emissive_color.n < 1) return NULL;
    for (int i = 0; i < n_es; i++) {
        if (es[i].particle_index < 0 || es[i].particle_index >= n_ps) return NULL;
        if (es[i].mode == FWO_MODE_NESTED && (es[i].target_particle_type < 0 || es[i].target_particle_type >= n_ps))
            return NULL;
    }
    fwo_spawner *s = (fwo_spawner *)calloc(1, sizeof *s);
    s->n_ps = n_ps, s->n_es = n_es;
    s->ps = (fwo_particle_settings *)malloc(sizeof(*ps) * (size_t)(n_ps ? n_ps : 1));
    s->es = (fwo_emission_settings *)malloc(sizeof(*es) * (size_t)(n_es ? n_es : 1));
    memcpy(s->ps, ps, sizeof(*ps) * (size_t)n_ps);
    memcpy(s->es, es, sizeof(*es) * (size_t)n_es);
    for (int i = 0; i < n_ps; i++) { /* deep-copy key arrays; normalise uneven cores */
        fwo_particle_settings *p = &s->ps[i];
        p->scale_curve.times = dup_f32(ps[i].scale_curve.times, (size_t)ps[i].scale_curve.n);
        p->scale_curve.values = dup_f32(ps[i].scale_curve.values, (size_t)ps[i].scale_curve.n);
        if (p->scale_curve.kind == FWO_CURVE_UNEVEN && p->scale_curve.n >= 2)
            p->scale_curve.n = uneven_normalize((float *)p->scale_curve.times, (float *)p->scale_curve.values,
                                                p->scale_curve.n, 1);
        fwo_gradient *gs[2] = {&p->base_color, &p->emissive_color};
        const fwo_gradient *gi[2] = {&ps[i].base_color, &ps[i].emissive_color};
        for (int k = 0; k < 2; k++) {
            gs[k]->times = dup_f32(gi[k]->times, (size_t)gi[k]->n);
            gs[k]->rgba = dup_f32(gi[k]->rgba, (size_t)gi[k]->n * 4);
            if (gs[k]->kind == FWO_CURVE_UNEVEN && gs[k]->n >= 2)
                gs[k]->n = uneven_normalize((float *)gs[k]->times, (float *)gs[k]->rgba, gs[k]->n, 4);
        }
    }
    s->starts_enabled = starts_enabled;
    s->seed = seed, s->uid = uid;
    s->particles = (pvec *)calloc((size_t)(n_ps ? n_ps : 1), sizeof(pvec));
    s->destroyed = (pvec *)calloc((size_t)(n_ps ? n_ps : 1), sizeof(pvec));
    s->emission = (emission_data *)calloc((size_t)(n_es ? n_es : 1), sizeof(emission_data));
    s->origin_rotation[3] = 1.0f;
    s->modifier_scale = 1.0f, s->modifier_speed = 1.0f; /* EffectModifier::default core.rs:329-336 */
    fwo_spawner_reset(s); /* first Changed<ParticleSpawner> tick */
    return s;
}

void fwo_spawner_destroy(fwo_spawner *s) {
    if (!s) return;
    for (int i = 0; i < s->n_ps; i++) {
        pvec_free(&s->particles[i]);
        pvec_free(&s->destroyed[i]);
        free((void *)s->ps[i].scale_curve.times);
        free((void *)s->ps[i].scale_curve.values);
        free((void *)s->ps[i].base_color.times);
        free((void *)s->ps[i].base_color.rgba);
        free((void *)s->ps[i].emissive_color.times);
        free((void *)s->ps[i].emissive_color.rgba);
    }
    free(s->colliders);
    free(s->particles), free(s->destroyed), free(s->emission), free(s->ps), free(s->es), free(s);
}

/* sync_spawner_data core.rs:343-365 (RNG serials are NOT reset: a stream never replays) */
void fwo_spawner_reset(fwo_spawner *s) {
    for (int i = 0; i < s->n_es; i++) {
        s->emission[i].last_emission = 0.0f;
        s->emission[i].time_passed_in_cycle = 0.0f;
        s->emission[i].enabled = s->starts_enabled;
        s->emission[i].emits_on_other_particles = (s->es[i].mode == FWO_MODE_NESTED);
    }
    for (int i = 0; i < s->n_ps; i++) {
        pvec_free(&s->particles[i]);
        pvec_free(&s->destroyed[i]);
    }
    s->initialized = 1;
}

void fwo_spawner_set_origin(fwo_spawner *s, const float t[3], const float r[4]) {
    memcpy(s->origin_translation, t, 3 * sizeof(float));
    memcpy(s->origin_rotation, r, 4 * sizeof(float));
}
void fwo_spawner_set_parent_velocity(fwo_spawner *s, const float v[3]) { memcpy(s->parent_velocity, v, 3 * sizeof(float)); }
void fwo_spawner_set_modifier(fwo_spawner *s, float scale, float speed) { s->modifier_scale = scale, s->modifier_speed = speed; }
/* ParticleSpawnerData::queue_particles core.rs:284-286 */
void fwo_spawner_queue(fwo_spawner *s, uint64_t n) { s->manual_queued_count += n; }
void fwo_spawner_set_colliders(fwo_spawner *s, const fwo_collider *colliders, int32_t n) {
    free(s->colliders);
    s->colliders = NULL, s->n_colliders = 0;
    if (n > 0) {
        s->colliders = (fwo_collider *)malloc(sizeof(fwo_collider) * (size_t)n);
        memcpy(s->colliders, colliders, sizeof(fwo_collider) * (size_t)n);
        s->n_colliders = n;
    }
}

/* ParticleSpawnerData::active core.rs:288-302 */
int32_t fwo_spawner_active(const fwo_spawner *s) {
    int enabled = 0;
    for (int i = 0; i < s->n_es; i++) {
        if (s->emission[i].emits_on_other_particles) {
            int any = 0;
            for (int t = 0; t < s->n_ps; t++) any |= (s->particles[t].n != 0);
            enabled |= (s->emission[i].enabled && any);
        } else {
            enabled |= s->emission[i].enabled;
        }
    }
    return enabled;
}

/* notify_finished_particle_spawners core.rs:674-688 */
int32_t fwo_spawner_poll_finished(fwo_spawner *s) {
    int all_empty = 1;
    for (int t = 0; t < s->n_ps; t++) all_empty &= (s->particles[t].n == 0);
    if (all_empty && !fwo_spawner_active(s) && s->initialized && !s->finished_notified) {
        s->finished_notified = 1;
        return 1;
    }
    return 0;
}

/* one new ParticleData (core.rs:437-469 Global, 506-544 Nested) */
static void spawn_one(fwo_spawner *s, int ei, const float origin_pos[3], const float origin_rot[4],
                      const float inherit_vel[3]) {
    const fwo_emission_settings *e = &s->es[ei];
    const fwo_particle_settings *ps = &s->ps[e->particle_index];
    float u[12];
    fwo_spawn_uniforms(s->seed, s->uid, (uint32_t)ei, s->emission[ei].serial++, u);

    float spawn_offset[3];
    fwo_shape_generate(e, u, spawn_offset);

    float vr[3], rv[3], n[3];
    fwo_randvec3_generate(&e->initial_velocity, u[3], u[4], u[5], vr);
    fwo_quat_mul_vec3(origin_rot, vr, rv);
    v3_normalize_or_zero(spawn_offset, n);
    float radial = randf32_generate(&e->initial_velocity_radial, u[6]);

    particle p;
    memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; k++) {
        float inner = rv[k] + n[k] * radial;
        float inh = e->inherit_parent_velocity ? inherit_vel[k] : 0.0f;
        p.velocity[k] = s->modifier_speed * inner + inh;
        p.position[k] = origin_pos[k] + spawn_offset[k];
    }
    float initial_scale = randf32_generate(&ps->initial_scale, u[7]) * s->modifier_scale;
    p.lifetime = randf32_generate(&ps->lifetime, u[8]);
    p.initial_scale = initial_scale;
    p.scale = initial_scale;
    p.age = 0.0f;
    fwo_gradient_sample_clamped(&ps->base_color, 0.0f, p.base_color);
    fwo_gradient_sample_clamped(&ps->emissive_color, 0.0f, p.emissive_color);
    p.pbr = ps->pbr;
    memcpy(p.rotation, e->initial_rotation, 4 * sizeof(float));
    fwo_randvec3_generate(&e->initial_angular_velocity, u[9], u[10], u[11], p.angular_velocity);
    p.last_emitted_age = lea_new(s->n_es, FWO_F32_MIN);
    pvec_push(&s->particles[e->particle_index], &p);
}

/* spawn_particles core.rs:367-551 (one spawner) */
void fwo_spawner_spawn(fwo_spawner *s, float dt) {
    if (!fwo_spawner_active(s)) return;
    for (int i = 0; i < s->n_es; i++) {
        const fwo_emission_settings *e = &s->es[i];
        emission_data *ed = &s->emission[i];
        if (!ed->enabled) continue;
        if (e->mode == FWO_MODE_GLOBAL) {
            uint64_t n = 0;
            if (e->pacing_kind == FWO_PACING_ONESHOT) {
                ed->enabled = 0;
                n = e->oneshot_count;
            } else if (e->pacing_kind == FWO_PACING_ONDEMAND) {
                n = s->manual_queued_count;
                s->manual_queued_count = 0;
            } else {
                ed->time_passed_in_cycle = fwo_rem_euclid(ed->time_passed_in_cycle + dt, e->duration);
                float next;
                n = fwo_compute_emission_count(ed->time_passed_in_cycle, ed->last_emission, e->duration,
                                               e->offset_start, e->offset_end, e->count, &next);
                ed->last_emission = next;
            }
            for (uint64_t k = 0; k < n; k++)
                spawn_one(s, i, s->origin_translation, s->origin_rotation, s->parent_velocity);
        } else {
            if (e->pacing_kind != FWO_PACING_COUNT_OVER_DURATION) continue; /* warn_once + continue core.rs:474-485 */
            size_t n_parents = s->particles[e->target_particle_type].n; /* bound fixed once core.rs:488 */
            for (size_t pi = 0; pi < n_parents; pi++) {
                particle *op = &s->particles[e->target_particle_type].p[pi];
                float next;
                uint64_t n = fwo_compute_emission_count(op->age, op->last_emitted_age[i], op->lifetime,
                                                        e->offset_start, e->offset_end, e->count, &next);
                op->last_emitted_age[i] = next;
                float opos[3], orot[4], ovel[3];
                memcpy(opos, op->position, sizeof opos);
                memcpy(orot, op->rotation, sizeof orot);
                memcpy(ovel, op->velocity, sizeof ovel);
                for (uint64_t k = 0; k < n; k++) spawn_one(s, i, opos, orot, ovel); /* may realloc: op is dead after */
            }
        }
    }
}

/* update_particles core.rs:577-670; the physics_avian arm (core.rs:607-624) when the type has collision settings */
void fwo_spawner_update(fwo_spawner *s, float dt) {
    for (int i = 0; i < s->n_ps; i++) {
        const fwo_particle_settings *ps = &s->ps[i];
        pvec *src = &s->particles[i];
        pvec dst = {0}, destroyed = {0};
        for (size_t k = 0; k < src->n; k++) {
            particle p = src->p[k]; /* particle.clone(): deep-clones the Vec<f32> */
            p.last_emitted_age = lea_clone(src->p[k].last_emitted_age, s->n_es);

            p.age += dt;
            if (p.age >= p.lifetime) {
                pvec_push(&destroyed, &p);
                continue;
            }
            float age_percent = p.age / p.lifetime;
            float scale_factor = fwo_curve_sample_clamped(&ps->scale_curve, age_percent);
            p.scale = p.initial_scale * scale_factor;

            float sa[3];
            if (ps->coll_enabled) { /* core.rs:607-624 (feature physics_avian) */
                int destroy = fwo_particle_collision(p.position, p.velocity, dt, ps->coll_restitution, ps->coll_friction,
                                                     ps->coll_destroy_on_collision, ps->coll_filter_mask, s->colliders,
                                                     s->n_colliders);
                if (destroy) { /* core.rs:636-639: the record carries the new position / velocity / scale */
                    pvec_push(&destroyed, &p);
                    continue;
                }
            } else {
                for (int c = 0; c < 3; c++) p.position[c] = p.position[c] + p.velocity[c] * dt;
            }
            for (int c = 0; c < 3; c++) {
                float v = p.velocity[c];
                p.velocity[c] = v + (ps->acceleration[c] - v * ps->linear_drag) * dt;
                sa[c] = p.angular_velocity[c] * dt;
            }
            float dq[4], nr[4];
            fwo_quat_from_scaled_axis(sa, dq);
            fwo_quat_mul(dq, p.rotation, nr);
            memcpy(p.rotation, nr, sizeof nr);
            for (int c = 0; c < 3; c++) {
                float w = p.angular_velocity[c];
                p.angular_velocity[c] = w + (ps->angular_acceleration[c] - ps->angular_drag * w) * dt;
            }
            fwo_gradient_sample_clamped(&ps->base_color, age_percent, p.base_color);
            fwo_gradient_sample_clamped(&ps->emissive_color, age_percent, p.emissive_color);
            pvec_push(&dst, &p);
        }
        pvec_free(src); /* old Vec dropped */
        *src = dst;
        pvec_free(&s->destroyed[i]);
        s->destroyed[i] = destroyed;
    }
}

void fwo_spawner_step(fwo_spawner *s, float dt) {
    fwo_spawner_spawn(s, dt);
    fwo_spawner_update(s, dt);
}

/* ------------------------------------------------------------------------- */
/* accessors                                                                   */
/* ------------------------------------------------------------------------- */

static void flatten(const particle *p, fwo_particle_flat *o) {
    memcpy(o->position, p->position, sizeof o->position);
    memcpy(o->velocity, p->velocity, sizeof o->velocity);
    memcpy(o->rotation, p->rotation, sizeof o->rotation);
    memcpy(o->angular_velocity, p->angular_velocity, sizeof o->angular_velocity);
    o->initial_scale = p->initial_scale, o->scale = p->scale, o->age = p->age, o->lifetime = p->lifetime;
    memcpy(o->base_color, p->base_color, sizeof o->base_color);
    memcpy(o->emissive_color, p->emissive_color, sizeof o->emissive_color);
    o->pbr = p->pbr;
}

uint64_t fwo_spawner_count(const fwo_spawner *s, int32_t type) {
    return (type >= 0 && type < s->n_ps) ? s->particles[type].n : 0;
}

static uint64_t read_vec(const pvec *v, fwo_particle_flat *out, uint64_t cap) {
    uint64_t n = v->n < cap ? v->n : cap;
    for (uint64_t i = 0; i < n; i++) flatten(&v->p[i], &out[i]);
    return v->n;
}
uint64_t fwo_spawner_read(const fwo_spawner *s, int32_t type, fwo_particle_flat *out, uint64_t cap) {
    if (type < 0 || type >= s->n_ps) return 0;
    return read_vec(&s->particles[type], out, cap);
}
uint64_t fwo_spawner_read_destroyed(const fwo_spawner *s, int32_t type, fwo_particle_flat *out, uint64_t cap) {
    if (type < 0 || type >= s->n_ps) return 0;
    return read_vec(&s->destroyed[type], out, cap);
}
uint64_t fwo_spawner_read_last_emitted(const fwo_spawner *s, int32_t type, int32_t ei, float *out, uint64_t cap) {
    if (type < 0 || type >= s->n_ps || ei < 0 || ei >= s->n_es) return 0;
    const pvec *v = &s->particles[type];
    uint64_t n = v->n < cap ? v->n : cap;
    for (uint64_t i = 0; i < n; i++) out[i] = v->p[i].last_emitted_age[ei];
    return v->n;
}

void fwo_spawner_write(fwo_spawner *s, int32_t type, const fwo_particle_flat *in, uint64_t n) {
    if (type < 0 || type >= s->n_ps) return;
    pvec *v = &s->particles[type];
    pvec_free(v);
    for (uint64_t i = 0; i < n; i++) {
        particle p;
        memset(&p, 0, sizeof p);
        memcpy(p.position, in[i].position, sizeof p.position);
        memcpy(p.velocity, in[i].velocity, sizeof p.velocity);
        memcpy(p.rotation, in[i].rotation, sizeof p.rotation);
        memcpy(p.angular_velocity, in[i].angular_velocity, sizeof p.angular_velocity);
        p.initial_scale = in[i].initial_scale, p.scale = in[i].scale, p.age = in[i].age, p.lifetime = in[i].lifetime;
        memcpy(p.base_color, in[i].base_color, sizeof p.base_color);
        memcpy(p.emissive_color, in[i].emissive_color, sizeof p.emissive_color);
        p.pbr = in[i].pbr;
        p.last_emitted_age = lea_new(s->n_es, FWO_F32_MIN);
        pvec_push(v, &p);
    }
}
void fwo_spawner_write_last_emitted(fwo_spawner *s, int32_t type, int32_t ei, const float *in, uint64_t n) {
    if (type < 0 || type >= s->n_ps || ei < 0 || ei >= s->n_es) return;
    pvec *v = &s->particles[type];
    for (uint64_t i = 0; i < n && i < v->n; i++) v->p[i].last_emitted_age[ei] = in[i];
}

/* update_aabbs render.rs:677-703: fold(Vec3::MAX, min) / fold(Vec3::MIN, max) of position -/+ scale */
int32_t fwo_spawner_aabb(const fwo_spawner *s, float mn[3], float mx[3]) {
    for (int c = 0; c < 3; c++) mn[c] = 3.40282347e+38f, mx[c] = FWO_F32_MIN;
    int any = 0;
    for (int t = 0; t < s->n_ps; t++)
        for (size_t i = 0; i < s->particles[t].n; i++) {
            const particle *p = &s->particles[t].p[i];
            any = 1;
            for (int c = 0; c < 3; c++) {
                mn[c] = fminf(mn[c], p->position[c] - p->scale);
                mx[c] = fmaxf(mx[c], p->position[c] + p->scale);
            }
        }
    return any;
}
