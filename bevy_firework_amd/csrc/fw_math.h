// fw_math.h -- fp32 arithmetic of the particle path, shared by the host engine and
// the gfx950 kernels.  Compiled with -ffp-contract=off: the reference (Rust) never
// fuses a*b+c, and particle counts / ordering depend on unfused fp32 results.
// Operation order follows the reference line by line (cited per function); glam /
// bevy_math / bevy_color semantics are the published scalar algorithms.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define FW_HD __host__ __device__ __forceinline__

#define FW_PI 3.14159265358979323846f
#define FW_F32_MIN (-3.40282347e+38f)

struct fw_v3 { float x, y, z; };
struct fw_q4 { float x, y, z, w; };

FW_HD float fw_dot3(fw_v3 a, fw_v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
FW_HD fw_v3 fw_cross(fw_v3 a, fw_v3 b) {
    return fw_v3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}

// f32::div_euclid / rem_euclid (reference src/core.rs:412-414,569)
// (a % b carries the sign of a, so it can only be negative when a is: the remainder is not evaluated otherwise)
FW_HD float fw_div_euclid(float a, float b) {
    float q = truncf(a / b);
    if (a < 0.0f && fmodf(a, b) < 0.0f) return (b > 0.0f) ? q - 1.0f : q + 1.0f;
    return q;
}
// (the two common cases of the emission clock, 0 <= a < b and b <= a < 2b, have exact closed forms: a itself, and
// a - b, which fp32 subtracts exactly for b <= a <= 2b -- the same values fmodf returns, without its loop)
FW_HD float fw_rem_euclid(float a, float b) {
    if (b > 0.0f && a >= 0.0f) {
        if (a < b) return a;
        if (a - b < b) return a - b;
    }
    float r = fmodf(a, b);
    return (r < 0.0f) ? r + fabsf(b) : r;
}

// `x as usize` saturated to 64 bits; NaN and negatives -> 0 (src/core.rs:570)
FW_HD uint64_t fw_as_usize(float x) {
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return 0xFFFFFFFFFFFFFFFFull;
    return (uint64_t)x;
}

// compute_emission_count (src/core.rs:553-575)
FW_HD uint64_t fw_emission_count(float time_passed, float last_emission, float duration, float start, float end,
                                 float per_cycle, float *next_last) {
    float percent_passed = time_passed / duration;
    float last_percent = last_emission / duration;
    float base = fmaxf(last_percent, start);
    float since = fminf(percent_passed, end) - base;
    float between = (end - start) / per_cycle;
    float times = fw_div_euclid(since, between);
    float adv = times * between;
    float next_percent = base + adv;
    *next_last = next_percent * duration;
    return fw_as_usize(times);
}

// ... for an entry whose duration is exactly 1 (every EmissionPacing::rate, core.rs:36-43), with `between` = (end - start) /
// per_cycle evaluated once when the entry is built: x / 1 and x * 1 are x in IEEE arithmetic, so this returns the very bits
// fw_emission_count returns with one division instead of four (the host's spawner loop, thousands of emitters per frame)
FW_HD uint64_t fw_emission_count_unit(float time_passed, float last_emission, float start, float end, float between, float *next_last) {
    float base = fmaxf(last_emission, start);
    float since = fminf(time_passed, end) - base;
    float times = fw_div_euclid(since, between);
    float adv = times * between;
    *next_last = base + adv;
    return fw_as_usize(times);
}

// Vec3::normalize_or_zero (src/core.rs:442,512)
FW_HD fw_v3 fw_normalize_or_zero(fw_v3 a) {
    float rcp = 1.0f / sqrtf(fw_dot3(a, a));
    if (rcp > 0.0f && rcp < INFINITY) return fw_v3{a.x * rcp, a.y * rcp, a.z * rcp};
    return fw_v3{0.0f, 0.0f, 0.0f};
}

// Quat * Quat, scalar Hamilton product (src/core.rs:645-647)
FW_HD fw_q4 fw_quat_mul(fw_q4 a, fw_q4 b) {
    fw_q4 o;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return o;
}

// Quat * Vec3 (src/core.rs:441,510)
FW_HD fw_v3 fw_quat_mul_vec3(fw_q4 q, fw_v3 v) {
    fw_v3 b{q.x, q.y, q.z};
    float b2 = fw_dot3(b, b);
    float k0 = q.w * q.w - b2;
    float k1 = fw_dot3(v, b) * 2.0f;
    fw_v3 c = fw_cross(b, v);
    float k2 = q.w * 2.0f;
    return fw_v3{(v.x * k0 + b.x * k1) + c.x * k2, (v.y * k0 + b.y * k1) + c.y * k2,
                 (v.z * k0 + b.z * k1) + c.z * k2};
}

FW_HD fw_q4 fw_quat_from_axis_angle(fw_v3 axis, float angle) {
    float h = angle * 0.5f;
    float s, c;
    sincosf(h, &s, &c);
    return fw_q4{axis.x * s, axis.y * s, axis.z * s, c};
}

// Quat::from_scaled_axis (src/core.rs:645-647): identity when the axis is zero
FW_HD fw_q4 fw_quat_from_scaled_axis(fw_v3 v) {
    float len = sqrtf(fw_dot3(v, v));
    if (len == 0.0f) return fw_q4{0.0f, 0.0f, 0.0f, 1.0f};
    return fw_quat_from_axis_angle(fw_v3{v.x / len, v.y / len, v.z / len}, len);
}

// Quat::from_rotation_arc (emission_shape.rs:34; RandVec3 cone)
FW_HD fw_q4 fw_quat_from_rotation_arc(fw_v3 from, fw_v3 to) {
    const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-7f;
    float d = fw_dot3(from, to);
    if (d > one_minus_eps) return fw_q4{0.0f, 0.0f, 0.0f, 1.0f};
    if (d < -one_minus_eps) {
        float sign = copysignf(1.0f, from.z);  // Vec3::any_orthonormal_vector
        float a = -1.0f / (sign + from.z);
        float b = from.x * from.y * a;
        return fw_quat_from_axis_angle(fw_v3{b, sign + from.y * from.y * a, -from.y}, FW_PI);
    }
    fw_v3 c = fw_cross(from, to);
    float w = 1.0f + d;
    float len = sqrtf((c.x * c.x) + (c.y * c.y) + (c.z * c.z) + (w * w));
    float inv = 1.0f / len;
    return fw_q4{c.x * inv, c.y * inv, c.z * inv, w * inv};
}

// ---- Philox4x32-10 counter RNG (one stream per emission entry) ------------------
// key = (ctx seed, spawner uid); counter = (serial lo, serial hi, emission index, block)
struct fw_u4 { uint32_t x, y, z, w; };

FW_HD fw_u4 fw_philox4x32_10(fw_u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        fw_u4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// rand 0.9 f32 sampling: 24 high bits -> [0, 1)
FW_HD float fw_unit_f32(uint32_t x) { return (float)(x >> 8) * 5.9604645e-8f; }

// ---- curve cores (bevy_math 0.19 EvenCore / UnevenCore::sample_with) -------------
// returns true when the sample lies strictly between keys lo and lo+1 (fraction *s)
FW_HD bool fw_even_interp(int n, float t, int *lo, float *s) {
    int subdivs = n - 1;
    float steps_taken;
    if ((subdivs & (subdivs - 1)) == 0) {
        // step = 1/subdivs is a power of two: (t - 0) / step == t * subdivs exactly, no division needed
        steps_taken = (t - 0.0f) * (float)subdivs;
    } else {
        float step = 1.0f / (float)subdivs;
        steps_taken = (t - 0.0f) / step;
    }
    if (steps_taken <= 0.0f) {
        *lo = 0;
        return false;
    }
    if (steps_taken >= (float)subdivs) {
        *lo = n - 1;
        return false;
    }
    *lo = (int)fw_as_usize(floorf(steps_taken));
    *s = steps_taken - truncf(steps_taken);
    return true;
}

FW_HD bool fw_uneven_interp(const float *times, int n, float t, int *lo, float *s) {
    int idx = 0;
    while (idx < n && times[idx] < t) idx++;
    if (idx < n && times[idx] == t) {
        *lo = idx;
        return false;
    }
    if (idx == 0) {
        *lo = 0;
        return false;
    }
    if (idx >= n) {
        *lo = n - 1;
        return false;
    }
    float t_lower = times[idx - 1], t_upper = times[idx];
    *s = (t - t_lower) / (t_upper - t_lower);
    *lo = idx - 1;
    return true;
}

FW_HD float fw_clampf(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// FireworkCurve<f32>::sample_clamped (curve.rs:26-32): clamp to the domain, then VectorSpace::lerp a * (1 - s) + b * s
// (bevy_math's StableInterpolate for NormedVectorSpace types; not glam's a + (b - a) * s)
FW_HD float fw_curve_sample(int kind, int n, const float *times, const float *vals, float t) {
    if (kind == 0 || n == 1) return vals[0];
    if (kind == 1 && n == 2) {
        // two evenly spaced keys (the linear fade every example uses): EvenCore with one subdivision is
        // steps_taken = t, lo = 0, s = t - trunc(t) = t inside (0, 1) -- the same values without the index arithmetic
        // (the clamp to [0, 1] changes nothing here: outside (0, 1) a key is selected, inside the clamp is the identity,
        // and a NaN passes through both; without it `1 - t` is the same value the two-key gradients use)
        const float a = vals[0], b = vals[1];
        return t <= 0.0f ? a : (t >= 1.0f ? b : a * (1.0f - t) + b * t);
    }
    int lo;
    float s = 0.0f;
    bool between;
    if (kind == 1) {
        t = fw_clampf(t, 0.0f, 1.0f);
        between = fw_even_interp(n, t, &lo, &s);
    } else {
        t = fw_clampf(t, times[0], times[n - 1]);
        between = fw_uneven_interp(times, n, t, &lo, &s);
    }
    float a = vals[lo];
    if (!between) return a;
    float b = vals[lo + 1];
    return a * (1.0f - s) + b * s;
}

// FireworkGradient<LinearRgba>::sample_clamped (curve.rs:111-114,156-158); Mix: a*(1-f) + b*f
FW_HD void fw_gradient_sample(int kind, int n, const float *times, const float *rgba, float t, float out[4]) {
    if (kind == 1 && n == 2) {  // two evenly spaced keys: see fw_curve_sample (no clamp here: the ends select a key)
        const bool lo_end = t <= 0.0f, hi_end = t >= 1.0f;
        const float nf = 1.0f - t;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float a = rgba[c], b = rgba[4 + c];
            out[c] = lo_end ? a : (hi_end ? b : a * nf + b * t);
        }
        return;
    }
    int lo = 0;
    float s = 0.0f;
    bool between = false;
    if (!(kind == 0 || n == 1)) {
        between = (kind == 1) ? fw_even_interp(n, t, &lo, &s) : fw_uneven_interp(times, n, t, &lo, &s);
    }
    const float *a = rgba + 4 * lo;
    if (!between) {
        out[0] = a[0], out[1] = a[1], out[2] = a[2], out[3] = a[3];
        return;
    }
    const float *b = a + 4;
    float nf = 1.0f - s;
    out[0] = a[0] * nf + b[0] * s;
    out[1] = a[1] * nf + b[1] * s;
    out[2] = a[2] * nf + b[2] * s;
    out[3] = a[3] * nf + b[3] * s;
}
