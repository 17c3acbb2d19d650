// cgrt_math.h -- float3 arithmetic and the reference's primitive intersectors, written once for
// host and gfx950 device code.
//
// Bit-exactness contract (DESIGN.md "Numerics"): every expression below keeps the operand order of
// the reference (src/ray_tracing.cpp + glm 0.9.9.8 scalar formulas) and is
// compiled with -ffp-contract=off, IEEE division and square root, denormals preserved.  With those
// flags x86-64 SSE and gfx950 VALU produce identical bit patterns for + - * / sqrt and for ordered
// comparisons, which is what makes `t` reproducible.  Never replace a/b by a*rcp(b), never use
// fminf/fmaxf where the reference uses a ternary (NaN order differs), never reassociate.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CGRT_HD __host__ __device__ __forceinline__
#else
#define CGRT_HD inline
#endif

namespace cgrt {

struct F3 {
    float x, y, z;
};

CGRT_HD F3 f3(float x, float y, float z) {
    F3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
CGRT_HD F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
CGRT_HD F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
CGRT_HD F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }
CGRT_HD F3 scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
// glm::dot(vec3): products first, then (x + y) + z
CGRT_HD float dot(F3 a, F3 b) {
    float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
    return (px + py) + pz;
}
// glm::cross
CGRT_HD F3 cross(F3 a, F3 b) { return f3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }

// Correctly rounded square roots.  __builtin_sqrtf lowers to llvm.sqrt.f32, which hipcc expands to the
// IEEE-correct refinement sequence under -fhip-fp32-correctly-rounded-divide-sqrt (its default).
// NOT __fsqrt_rn / __builtin_amdgcn_sqrtf: those emit a bare v_sqrt_f32 (1 ulp), which broke parity.
CGRT_HD float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
CGRT_HD double sqrt_ieee_d(double x) { return __builtin_sqrt(x); }
// glm::normalize(v) = v * inversesqrt(dot(v, v)), inversesqrt(x) = 1.0f / sqrt(x)
CGRT_HD F3 normalize(F3 v) { return scale(v, 1.0f / sqrt_ieee(dot(v, v))); }
CGRT_HD float length(F3 v) { return sqrt_ieee(dot(v, v)); }

// ---- ray_tracing.cpp:74-82 trianglePlane ----
CGRT_HD void triangle_plane(F3 v0, F3 v1, F3 v2, F3& n, float& D) {
    n = normalize(cross(sub(v1, v0), sub(v2, v0)));
    D = dot(v0, n);
}

// ---- ray_tracing.cpp:23-38 pointInTriangle ----
CGRT_HD bool point_in_triangle(F3 v0, F3 v1, F3 v2, F3 n, F3 p) {
    F3 e01 = sub(v1, v0), e12 = sub(v2, v1), e20 = sub(v0, v2);
    F3 q0 = sub(p, v0), q1 = sub(p, v1), q2 = sub(p, v2);
    return dot(n, cross(e01, q0)) >= 0 && dot(n, cross(e12, q1)) >= 0 && dot(n, cross(e20, q2)) >= 0;
}

// ---- ray_tracing.cpp:40-72 intersectRayWithPlane ----
// Returns true and writes t when the plane test passes (including the origin-on-plane case, which
// yields t = 0 with no `t < ray.t` guard, :43-47).
CGRT_HD bool ray_plane(float D, F3 n, F3 o, F3 d, float& t) {
    float on = dot(o, n);
    if (on == D) {
        t = 0;
        return true;
    }
    float den = dot(d, n);
    if (den == 0) return false;
    float tt = (D - on) / den;
    if (tt < 0) return false;
    if (tt >= t) return false;
    t = tt;
    return true;
}

// ---- ray_tracing.cpp:86-114 intersectRayWithTriangle, geometric part, with the plane
//      (ray-independent, :74-82) supplied precomputed.  On accept t is updated. ----
CGRT_HD bool ray_triangle_geom(F3 v0, F3 v1, F3 v2, F3 n, float D, F3 o, F3 d, float& t) {
    float tt = t;
    if (!ray_plane(D, n, o, d, tt)) return false;
    F3 p = add(o, scale(d, tt));
    if (!point_in_triangle(v0, v1, v2, n, p)) return false;  // :110 rollback == not committing tt
    t = tt;
    return true;
}

// ray_tracing.cpp:13-21 magnitude/area: squares and sum in double, sqrt in double, narrowed to float.
CGRT_HD float magnitude_ref(F3 a) {
    double s = ((double)a.x * (double)a.x + (double)a.y * (double)a.y) + (double)a.z * (double)a.z;
    return (float)sqrt_ieee_d(s);
}
CGRT_HD float area_ref(F3 v0, F3 v1, F3 v2) { return magnitude_ref(cross(sub(v1, v0), sub(v2, v0))) / 2.0f; }

// ---- ray_tracing.cpp:94-107: hitInfo.normal of an accepted triangle hit at parameter t ----
CGRT_HD F3 hit_normal(F3 v0, F3 v1, F3 v2, F3 pn, F3 n1, F3 n2, F3 n3, F3 o, F3 d, float t) {
    F3 p = add(o, scale(d, t));
    float a012 = area_ref(v0, v1, v2);
    float alpha = area_ref(p, v1, v2) / a012;
    float beta = area_ref(p, v0, v2) / a012;
    float gamma = area_ref(p, v0, v1) / a012;
    // glm: float * vec3 per component, sums left to right
    F3 s = add(add(f3(alpha * n1.x, alpha * n1.y, alpha * n1.z), f3(beta * n2.x, beta * n2.y, beta * n2.z)),
               f3(gamma * n3.x, gamma * n3.y, gamma * n3.z));
    F3 ni = normalize(s);
    return (dot(pn, neg(d)) > 0) ? ni : neg(ni);
}

// ---- ray_tracing.cpp:162-200 intersectRayWithShape(AxisAlignedBox) ----
// Returns true and writes the box parameter (entry, or exit when the origin is inside the slabs)
// into tbox when it is < t.  Does NOT modify the ray (the reference writes ray.t and every caller
// restores it: bvh.cpp:717-733, :835-838).
CGRT_HD bool ray_box(F3 lo, F3 hi, F3 o, F3 d, float t, float& tbox) {
    float tMinX = (lo.x - o.x) / d.x, tMinY = (lo.y - o.y) / d.y, tMinZ = (lo.z - o.z) / d.z;
    float tMaxX = (hi.x - o.x) / d.x, tMaxY = (hi.y - o.y) / d.y, tMaxZ = (hi.z - o.z) / d.z;
    float tInX = tMinX < tMaxX ? tMinX : tMaxX;
    float tOutX = tMinX > tMaxX ? tMinX : tMaxX;
    float tInY = tMinY < tMaxY ? tMinY : tMaxY;
    float tOutY = tMinY > tMaxY ? tMinY : tMaxY;
    float tInZ = tMinZ < tMaxZ ? tMinZ : tMaxZ;
    float tOutZ = tMinZ > tMaxZ ? tMinZ : tMaxZ;
    float tIn = tInX > tInY ? (tInX > tInZ ? tInX : tInZ) : (tInY > tInZ ? tInY : tInZ);
    float tOut = tOutX < tOutY ? (tOutX < tOutZ ? tOutX : tOutZ) : (tOutY < tOutZ ? tOutY : tOutZ);
    if (tIn > tOut || tOut < 0) return false;
    float cur = (tIn < 0) ? tOut : tIn;
    if (cur >= t) return false;
    tbox = cur;
    return true;
}

// ---- bvh.cpp:647-661 startsInBox (strict) ----
CGRT_HD bool starts_in_box(F3 o, F3 lo, F3 hi) {
    bool inX = lo.x < o.x && o.x < hi.x;
    bool inY = lo.y < o.y && o.y < hi.y;
    bool inZ = lo.z < o.z && o.z < hi.z;
    return inX && inY && inZ;
}

// ---- ray_tracing.cpp:118-158 intersectRayWithShape(Sphere) ----
// The unqualified sqrt(D) resolves to ::sqrt(double) under libstdc++'s <cmath>, so the roots are
// evaluated in double and narrowed (documented platform dependence, DESIGN.md "Numerics").
CGRT_HD bool ray_sphere(F3 c, float radius, F3 o, F3 d, float& t, F3& nrm) {
    F3 co = sub(o, c);
    float a = dot(d, d);
    float b = 2 * dot(d, co);
    float cc = dot(co, co) - radius * radius;
    float disc = b * b - 4 * a * cc;
    if (disc < 0) return false;
    double sq = sqrt_ieee_d((double)disc);
    float smallerT = (float)(((double)(-b) - sq) / (double)(2 * a));
    float biggerT = (float)(((double)(-b) + sq) / (double)(2 * a));
    float cur;
    if (smallerT >= 0)
        cur = smallerT;
    else if (biggerT >= 0)
        cur = biggerT;
    else
        return false;
    if (cur >= t) return false;
    t = cur;
    nrm = normalize(sub(add(o, scale(d, t)), c));
    return true;
}

// ---- soft shadows (main.cpp:168-218): which caller-supplied randomUnitVector() draw sample `smp` of (pixel, recursion
//      level, spherical light) uses.  Upstream draws from std::random_device (:46-59), which nothing can reproduce; the
//      draws are a table here and this integer hash (murmur3's 32-bit finaliser, chained) picks from it -- include/cgrt.h.
CGRT_HD uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
CGRT_HD uint32_t soft_sample_index(uint32_t seed, uint32_t pixel, uint32_t level, uint32_t light, uint32_t smp, uint32_t nunits) {
    uint32_t h = mix32(seed ^ 0x9e3779b9u);
    h = mix32(h ^ pixel);
    h = mix32(h ^ (level * 0x01000193u + light));
    h = mix32(h ^ smp);
    return h % nunits;
}
// One soft-shadow sample ray (main.cpp:177-181): towards position + radius * u, started 0.001 along; t = distance from the
// shifted origin to the sample point (`lightT`, :180).
CGRT_HD void soft_shadow_ray(F3 pointOn, F3 lpos, float radius, F3 u, F3& o, F3& d, float& t) {
    const F3 rp = add(lpos, f3(radius * u.x, radius * u.y, radius * u.z));
    d = normalize(sub(rp, pointOn));
    const float eps = (float)(0.001);
    o = add(pointOn, f3(eps * d.x, eps * d.y, eps * d.z));
    t = length(sub(o, rp));
}

// ---- camera: glm::qua(euler), qua * vec3 (glm 0.9.9.8), Trackball::generateRay ----
struct Q4 {
    float w, x, y, z;
};
CGRT_HD F3 quat_rotate(Q4 q, F3 v) {
    F3 qv = f3(q.x, q.y, q.z);
    F3 uv = cross(qv, v);
    F3 uuv = cross(qv, uv);
    return add(v, scale(add(scale(uv, q.w), uuv), 2.0f));
}

}  // namespace cgrt
