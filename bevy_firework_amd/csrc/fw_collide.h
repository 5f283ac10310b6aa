// fw_collide.h -- particle_collision (reference src/core.rs:744-800, feature physics_avian) against the context's
// device-resident analytic colliders (include/firework_hip.h: fw_collider).  fp32, operation order of the reference
// line by line; built with -ffp-contract=off like everything else.
//
// The reference casts its rays through avian's SpatialQuery (parry shapes behind a CPU broadphase).  That world does
// not exist on the device; the ray cast below is this backend's own definition (header comment of fw_collider),
// modelled on parry's `solid = true` casts: a ray starting inside a solid reports distance 0 with a zero normal.
#pragma once
#include "fw_math.h"

struct alignas(16) FwCollider {
    int32_t kind;
    uint32_t layers;
    float radius;
    float bound;           // radius of a sphere around `position` that contains the collider (INFINITY for a plane): set by the host
                           // (fw_ctx_set_colliders); lets a wave skip a collider none of its rays can reach (fw_cast_ray)
    float position[4];
    float rotation[4];     // xyzw (BOX, CYLINDER, CONE)
    float normal[4];       // PLANE
    float half_extents[4]; // BOX; [1] = half the height of a CYLINDER / CONE (their axis is the local Y axis)
};

struct FwRayHit {
    float distance;
    fw_v3 normal;
};

FW_HD float fw_len3(fw_v3 a) { return sqrtf(fw_dot3(a, a)); }
FW_HD fw_v3 fw_scale3(fw_v3 a, float s) { return fw_v3{a.x * s, a.y * s, a.z * s}; }
FW_HD fw_v3 fw_add3(fw_v3 a, fw_v3 b) { return fw_v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
FW_HD fw_v3 fw_sub3(fw_v3 a, fw_v3 b) { return fw_v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
// glam Vec3::normalize: self * length_recip()
FW_HD fw_v3 fw_normalize3(fw_v3 a) { return fw_scale3(a, 1.0f / fw_len3(a)); }
// glam Vec3::project_onto(rhs): rhs * self.dot(rhs) * rhs.dot(rhs).recip()
FW_HD fw_v3 fw_project_onto(fw_v3 a, fw_v3 rhs) {
    const float rcp = 1.0f / fw_dot3(rhs, rhs);
    return fw_scale3(fw_scale3(rhs, fw_dot3(a, rhs)), rcp);
}

// one collider; returns true and fills *hit when the ray origin + t dir, t in [0, max_distance], meets it
FW_HD bool fw_ray_collider(const FwCollider &c, fw_v3 origin, fw_v3 dir, float max_distance, FwRayHit *hit) {
    const fw_v3 cpos{c.position[0], c.position[1], c.position[2]};
    if (c.kind == 0) {  // PLANE: the half-space n.(x - p) <= 0
        const fw_v3 n{c.normal[0], c.normal[1], c.normal[2]};
        const float dot_normal_dpos = fw_dot3(n, fw_sub3(cpos, origin));
        if (dot_normal_dpos > 0.0f) {  // inside the solid half-space
            *hit = FwRayHit{0.0f, fw_v3{0.0f, 0.0f, 0.0f}};
            return true;
        }
        const float t = dot_normal_dpos / fw_dot3(n, dir);
        if (t >= 0.0f && t <= max_distance) {
            *hit = FwRayHit{t, n};
            return true;
        }
        return false;
    }
    if (c.kind == 1) {  // SPHERE
        const fw_v3 dcenter = fw_sub3(origin, cpos);
        const float a = fw_dot3(dir, dir);
        const float b = fw_dot3(dcenter, dir);
        const float cc = fw_dot3(dcenter, dcenter) - c.radius * c.radius;
        if (cc <= 0.0f) {  // inside (or on) the ball
            *hit = FwRayHit{0.0f, fw_v3{0.0f, 0.0f, 0.0f}};
            return true;
        }
        if (b > 0.0f) return false;  // outside and moving away
        const float delta = b * b - a * cc;
        if (!(delta >= 0.0f)) return false;
        const float t = (-b - sqrtf(delta)) / a;
        if (!(t >= 0.0f && t <= max_distance)) return false;
        const fw_v3 p = fw_sub3(fw_add3(origin, fw_scale3(dir, t)), cpos);
        *hit = FwRayHit{t, fw_normalize3(p)};
        return true;
    }
    // BOX, CYLINDER, CONE: solids with a frame of their own.  The ray is taken into that frame ONCE, each kind finds where the ray
    // enters (distance + the surface normal there, in the frame), and the normal is taken back ONCE: three inlined copies of the
    // rotations took the colliding ring kernels from 113 to 141 VGPRs (4 -> 3 waves per SIMD).
    const fw_q4 q{c.rotation[0], c.rotation[1], c.rotation[2], c.rotation[3]};
    const fw_q4 qi{-q.x, -q.y, -q.z, q.w};  // conjugate = inverse of a unit quaternion
    // (an axis-aligned solid -- the identity rotation, e.g. the ground slab of examples/stress_test_collision.rs, the cylinder and
    // the cone of examples/textures.rs -- needs no rotations: Quat::IDENTITY * v is v itself up to the sign of a zero component,
    // which no comparison or quotient below depends on (a zero direction component takes the `== 0` arm); the results are the
    // general path's, numerically equal)
    const bool aligned = q.x == 0.0f && q.y == 0.0f && q.z == 0.0f && q.w == 1.0f;
    const fw_v3 ol = aligned ? fw_sub3(origin, cpos) : fw_quat_mul_vec3(qi, fw_sub3(origin, cpos));
    const fw_v3 dl = aligned ? dir : fw_quat_mul_vec3(qi, dir);
    float t_hit;
    fw_v3 nl{0.0f, 0.0f, 0.0f};
    if (c.kind == 3) {
        // CYLINDER (avian Collider::cylinder(radius, height): axis = local Y, examples/textures.rs:195): the slab |y| <= half height
        // intersected with the infinite cylinder x^2 + z^2 <= r^2
        const float hh = c.half_extents[1], rr = c.radius * c.radius;
        const float c2 = (ol.x * ol.x + ol.z * ol.z) - rr;
        if (fabsf(ol.y) <= hh && c2 <= 0.0f) {  // inside (or on) the solid
            *hit = FwRayHit{0.0f, fw_v3{0.0f, 0.0f, 0.0f}};
            return true;
        }
        float tnear = -INFINITY, tfar = INFINITY, sign = 0.0f;
        int side = 0;  // what the ray enters through: 0 a cap, 1 the lateral surface
        if (dl.y == 0.0f) {
            if (fabsf(ol.y) > hh) return false;
        } else {
            const float inv = 1.0f / dl.y;
            float t1 = (-hh - ol.y) * inv, t2 = (hh - ol.y) * inv, sg = -1.0f;
            if (t1 > t2) {
                const float tmp = t1;
                t1 = t2, t2 = tmp, sg = 1.0f;
            }
            if (t1 > tnear) tnear = t1, side = 0, sign = sg;
            if (t2 < tfar) tfar = t2;
            if (tnear > tfar) return false;
        }
        const float a = dl.x * dl.x + dl.z * dl.z, b = ol.x * dl.x + ol.z * dl.z;
        if (a == 0.0f) {
            if (c2 > 0.0f) return false;
        } else {
            const float disc = b * b - a * c2;
            if (!(disc >= 0.0f)) return false;
            const float sq = sqrtf(disc);
            const float t1 = (-b - sq) / a, t2 = (-b + sq) / a;
            if (t1 > tnear) tnear = t1, side = 1;
            if (t2 < tfar) tfar = t2;
            if (tnear > tfar) return false;
        }
        if (!(tnear >= 0.0f && tnear <= max_distance)) return false;
        t_hit = tnear;
        nl = side == 0 ? fw_v3{0.0f, sign, 0.0f} : fw_normalize3(fw_v3{ol.x + dl.x * tnear, 0.0f, ol.z + dl.z * tnear});
    } else if (c.kind == 4) {
        // CONE (avian Collider::cone(radius, height): base disc at local y = -h/2, apex at y = +h/2, examples/textures.rs:211):
        // w = p - apex; the solid is  w.y <= 0,  y >= -h/2,  w.x^2 + w.z^2 <= k^2 w.y^2  with k = radius / height
        const float hh = c.half_extents[1], rr = c.radius * c.radius;
        const float k = c.radius / (hh + hh), k2 = k * k;
        const float wy = ol.y - hh;
        const float cq = (ol.x * ol.x + ol.z * ol.z) - k2 * (wy * wy);
        if (ol.y >= -hh && wy <= 0.0f && cq <= 0.0f) {  // inside (or on) the solid
            *hit = FwRayHit{0.0f, fw_v3{0.0f, 0.0f, 0.0f}};
            return true;
        }
        // the boundary of the solid is the base disc and the part of the lower nappe between base and apex: the first crossing of
        // either along the ray is where a ray from outside enters
        float best = INFINITY;
        int side = -1;  // 0 the base disc, 1 the lateral surface
        if (dl.y > 0.0f && ol.y < -hh) {
            const float tb = (-hh - ol.y) / dl.y;
            const float px = ol.x + dl.x * tb, pz = ol.z + dl.z * tb;
            if (px * px + pz * pz <= rr) best = tb, side = 0;
        }
        const float a = (dl.x * dl.x + dl.z * dl.z) - k2 * (dl.y * dl.y);
        const float b = (ol.x * dl.x + ol.z * dl.z) - k2 * (wy * dl.y);
        float ta = INFINITY, tb2 = INFINITY;  // (INFINITY: no such root)
        if (a == 0.0f) {
            if (b != 0.0f) ta = -cq / (b + b);
        } else {
            const float disc = b * b - a * cq;
            if (disc >= 0.0f) {
                const float sq = sqrtf(disc);
                ta = (-b - sq) / a, tb2 = (-b + sq) / a;
            }
        }
        {
            const float ya = ol.y + dl.y * ta, yb = ol.y + dl.y * tb2;
            if (ta >= 0.0f && ta < INFINITY && ya >= -hh && ya <= hh && ta < best) best = ta, side = 1;
            if (tb2 >= 0.0f && tb2 < INFINITY && yb >= -hh && yb <= hh && tb2 < best) best = tb2, side = 1;
        }
        if (side < 0 || !(best <= max_distance)) return false;
        t_hit = best;
        nl = fw_v3{0.0f, -1.0f, 0.0f};
        if (side == 1) {
            const fw_v3 w{ol.x + dl.x * best, (ol.y + dl.y * best) - hh, ol.z + dl.z * best};
            const fw_v3 g{w.x, -(k2 * w.y), w.z};  // gradient of x^2 + z^2 - k^2 y^2: outward on the lower nappe
            nl = (g.x == 0.0f && g.y == 0.0f && g.z == 0.0f) ? fw_v3{0.0f, 1.0f, 0.0f} : fw_normalize3(g);
        }
    } else {
        // BOX: slabs in the box's own frame
        // (the three slabs written out one by one, in axis order: indexed arrays of three would live in scratch memory on the device)
        const float hx = c.half_extents[0], hy = c.half_extents[1], hz = c.half_extents[2];
        if (fabsf(ol.x) <= hx && fabsf(ol.y) <= hy && fabsf(ol.z) <= hz) {  // inside (or on) the box
            *hit = FwRayHit{0.0f, fw_v3{0.0f, 0.0f, 0.0f}};
            return true;
        }
        float tnear = -INFINITY, tfar = INFINITY;
        int axis = 0;
        float sign = 0.0f;
#define FW_SLAB(o, d, h, i)                                                              \
    if ((d) == 0.0f) {                                                                   \
        if (fabsf(o) > (h)) return false;                                                \
    } else {                                                                             \
        const float inv = 1.0f / (d);                                                    \
        float t1 = (-(h) - (o)) * inv, t2 = ((h) - (o)) * inv;                           \
        float s = -1.0f; /* entering through the -h face: outward normal -e_i */         \
        if (t1 > t2) {                                                                   \
            const float tmp = t1;                                                        \
            t1 = t2, t2 = tmp, s = 1.0f;                                                 \
        }                                                                                \
        if (t1 > tnear) tnear = t1, axis = (i), sign = s;                                \
        if (t2 < tfar) tfar = t2;                                                        \
        if (tnear > tfar) return false;                                                  \
    }
        FW_SLAB(ol.x, dl.x, hx, 0)
        FW_SLAB(ol.y, dl.y, hy, 1)
        FW_SLAB(ol.z, dl.z, hz, 2)
#undef FW_SLAB
        if (!(tnear >= 0.0f && tnear <= max_distance)) return false;
        t_hit = tnear;
        if (axis == 0) nl.x = sign;
        else if (axis == 1) nl.y = sign;
        else nl.z = sign;
    }
    *hit = FwRayHit{t_hit, aligned ? nl : fw_quat_mul_vec3(q, nl)};
    return true;
}

// SpatialQuery::cast_ray(origin, dir, max_distance, solid = true, filter): nearest hit, lowest index on ties
FW_HD bool fw_cast_ray(const FwCollider *colliders, uint32_t n, uint32_t mask, fw_v3 origin, fw_v3 dir, float max_distance,
                       FwRayHit *best) {
    bool any = false;
    // (the best hit so far in scalars, written to *best once at the end: a struct updated through a pointer inside the loop
    // lived in scratch memory on the device)
    float bd = 0.0f, bnx = 0.0f, bny = 0.0f, bnz = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        if (!(colliders[i].layers & mask)) continue;
#ifdef __HIP_DEVICE_COMPILE__
        {
            // A collider NO lane of the wave can reach this sub-step is skipped by the whole wave (a uniform branch): its
            // bounding sphere lies further from every lane's origin than the ray is long.  A ray that meets the collider, or
            // starts inside it, is closer than bound + max_distance; the factor covers fp32 rounding of the comparison, and a
            // NaN / infinite operand compares false: no skip.  Skipped colliders report no hit either way, so the result is the
            // host oracle's, which tests every collider.
            const fw_v3 dc = fw_sub3(origin, fw_v3{colliders[i].position[0], colliders[i].position[1], colliders[i].position[2]});
            const float reach = colliders[i].bound + max_distance;
            const bool far = fw_dot3(dc, dc) > reach * reach * 1.0001f + 1e-12f;
            if (__ballot(!far) == 0ull) continue;
        }
#endif
        FwRayHit h;
        if (fw_ray_collider(colliders[i], origin, dir, max_distance, &h) && (!any || h.distance < bd)) {
            bd = h.distance, bnx = h.normal.x, bny = h.normal.y, bnz = h.normal.z;
            any = true;
        }
    }
    *best = FwRayHit{bd, fw_v3{bnx, bny, bnz}};
    return any;
}

// particle_collision (src/core.rs:744-800).  Returns should_destroy; *pos / *vel are updated in place.
FW_HD bool fw_particle_collision(fw_v3 *pos_io, fw_v3 *vel_io, float delta, float restitution, float friction,
                                 bool destroy_on_collision, uint32_t mask, const FwCollider *colliders, uint32_t n) {
    fw_v3 pos = *pos_io, vel = *vel_io;
    const float orig_delta = delta;
    int n_steps = 0;
    bool should_destroy = false;
    while (delta > 0.0f && n_steps < 4) {
        // Dir3::try_from(vel): value / length when the length is finite and > 0, else Dir3::Y (core.rs:758-761)
        const float len = fw_len3(vel);
        const fw_v3 dir = (len < INFINITY && len > 0.0f) ? fw_v3{vel.x / len, vel.y / len, vel.z / len} : fw_v3{0.0f, 1.0f, 0.0f};
        FwRayHit hit;
        if (fw_cast_ray(colliders, n, mask, pos, dir, fw_len3(vel) * delta, &hit)) {
            if (hit.distance == 0.0f) {  // core.rs:766-776
                fw_v3 normal = hit.normal;
                if (normal.x == 0.0f && normal.y == 0.0f && normal.z == 0.0f) {
                    if (vel.x != 0.0f || vel.y != 0.0f || vel.z != 0.0f)
                        normal = fw_normalize3(vel);
                    else
                        normal = fw_v3{0.0f, 1.0f, 0.0f};
                }
                const float k = fmaxf(fw_len3(vel), 1.0f);
                pos = fw_add3(pos, fw_scale3(fw_scale3(normal, k), delta));  // vel.length().max(1.) * normal * delta
            } else {  // core.rs:777-787
                pos = fw_add3(pos, fw_scale3(fw_normalize_or_zero(vel), hit.distance));
                const fw_v3 vel_project0 = fw_project_onto(vel, hit.normal);
                const fw_v3 vel_reject = fw_sub3(vel, vel_project0);          // reject_from = self - project_onto
                const fw_v3 vel_project = fw_project_onto(vel, hit.normal);
                const float friction_dv = fminf(fw_len3(vel_project), fw_len3(vel_reject)) * friction;
                vel = fw_sub3(fw_sub3(vel_reject, fw_scale3(fw_normalize_or_zero(vel_reject), friction_dv)),
                              fw_scale3(vel_project, restitution));
                pos = fw_add3(pos, fw_scale3(hit.normal, 0.0001f));
                float nd = delta - hit.distance;  // (delta - hit.distance).clamp(0., orig_delta): the reference's units
                if (nd < 0.0f) nd = 0.0f;
                if (nd > orig_delta) nd = orig_delta;
                delta = nd;
            }
            should_destroy = destroy_on_collision;
            if (should_destroy) break;
        } else {
            pos = fw_add3(pos, fw_scale3(vel, delta));
            delta = 0.0f;
        }
        n_steps++;
    }
    *pos_io = pos, *vel_io = vel;
    return should_destroy;
}
