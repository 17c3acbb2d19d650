// walk_exact.h -- device code of the EXACT walk: BoundingVolumeHierarchy::intersect
// (src/bounding_volume_hierarchy.cpp:850-881) for one ray per lane, step for step.
//
// The reference's recursion (intersectRecursive -> intersectNonLeaf -> intersectDeeper ->
// intersectRayThatStartsOutsideBoxes -> intersectChildrenHierarchically, bvh.cpp:572-758) is restated as an ordered
// stack walk: the child the reference would enter first is followed immediately, the other one is pushed together
// with its box parameter tSecond and is skipped on pop iff ray.t < tSecond -- the reference's
// `hitFirst && ray.t < tSecond` (bvh.cpp:581-585), since ray.t only changes when a triangle is accepted.  A child whose
// box test failed (t = -1) is dropped, and an origin strictly inside both child boxes visits both unconditionally
// (:685-688).  Inside a reference leaf the linear scan of intersectLeaf (bvh.cpp:535-553) is replaced by an order-free
// but outcome-identical evaluation over a per-leaf 4-wide BVH (scan_leaf below, DESIGN.md "In-leaf accelerator").
// The per-lane stack (<= 11 deferred children, bvh.cpp:48, plus <= 15 in-leaf entries) lives in LDS, lane-interleaved
// so that every access is bank-conflict free; runtime-indexed register arrays would go to scratch.
//
// Included by trace_kernels.hip (the shipped kernels) and variant_kernels.hip (persistent / stamped variants).
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no FMA contraction), default IEEE div/sqrt, denormals on --
// see cgrt_math.h for why.
#pragma once
#include <hip/hip_runtime.h>

#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "trace_kernels.h"

namespace cgrt {

// ---- experiment knobs (build variants with -D...; defaults are the shipped configuration) ----
#ifndef CGRT_MIN_WAVES
#define CGRT_MIN_WAVES 0  // __launch_bounds__ second argument (waves per SIMD) for the trace kernels, 0 = unset
#endif
#ifndef CGRT_MAX_WAVES
#define CGRT_MAX_WAVES 4  // register budget: 4 waves per SIMD = up to 128 VGPRs, no scratch spills (LDS alone would let the
                          // compiler aim at 5 and spill; measured slower)
#endif
#if CGRT_MIN_WAVES > 0
#define CGRT_LB __launch_bounds__(CGRT_BLOCK, CGRT_MIN_WAVES)
#else
#define CGRT_LB __launch_bounds__(CGRT_BLOCK) __attribute__((amdgpu_waves_per_eu(3, CGRT_MAX_WAVES)))
#endif
// Per-lane LDS stack, in 4-byte slots: a deferred reference child takes 2 slots (ref, tSecond), at most
// MAX_LEVELS-1 of them; an in-leaf accelerator entry takes 1 slot (ref), at most SUB_STACK_ENTRIES.
#define CGRT_STACK_SLOTS (2 * (MAX_LEVELS - 1) + SUB_STACK_ENTRIES)
// Dynamic LDS of a traversal kernel launched with `block` threads: the waves' stacks, then 16 words per wave for the quad
// tail's owner map (walk_fast.h), then block / 64 + 1 words of workgroup scratch.
#define CGRT_LDS_WORDS(block) ((block) * CGRT_STACK_SLOTS + ((block) / 64) * 16 + (block) / 64 + 1 + CGRT_HINT_WORDS)
#define CGRT_HINT_WORDS 6  // a hinted frame's wave keeps what hint_finish needs here (not in registers through the walk)
#define CGRT_HINT_SCRATCH(lds) (CGRT_BLOCK_SCRATCH(lds) + (blockDim.x >> 6) + 1)
// this wave's stack region / owner map inside the dynamic LDS array `lds`
#define CGRT_WAVE_STACK(lds) ((lds) + (threadIdx.x >> 6) * (CGRT_STACK_SLOTS * 64))
#define CGRT_WAVE_MAP(lds) ((lds) + blockDim.x * CGRT_STACK_SLOTS + (threadIdx.x >> 6) * 16)
#define CGRT_BLOCK_SCRATCH(lds) ((lds) + blockDim.x * CGRT_STACK_SLOTS + (blockDim.x >> 6) * 16)

struct LaneCounters {
    uint32_t inner = 0, leaf = 0, tri = 0, sub = 0;
    uint32_t cert = 0, fallback = 0, entered = 0;  // path boxes tested by certificates; rays sent to the exact walk by the certified walk; rays past the root gate
    // wave-level iteration counts (diagnostic): in every executed loop body exactly one active lane adds 1,
    // so the sum over lanes is the number of times the WAVE ran that body
    uint32_t w_inner = 0, w_sub = 0, w_tri = 0;
};

__device__ __forceinline__ bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63);
}

__device__ __forceinline__ F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }

// Per-ray constants of the CONSERVATIVE slab test used inside leaves (never for the reference's own
// box tests).  The test must never reject a box that contains a point the reference's float
// arithmetic could accept as a hit, so every box is widened by eps = 2^-16 * (|origin|max + |scene|max)
// -- 256x the rounding unit at the magnitudes involved -- and zero/tiny direction components are
// clamped away from zero (2^-40 relative), which moves the ray by far less than eps over any distance
// at which something can be hit.  Rays outside the range where that argument holds (non-finite
// components, |d|max or |o|max beyond 2^+-40, NaN t, a slack that would be denormal) are flagged irregular and test every triangle of a
// leaf instead.  DESIGN.md "In-leaf accelerator" has the full argument.
typedef float f2v __attribute__((ext_vector_type(2)));  // one operand of the packed FP32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
struct RayPre {
    F3 inv;
    // per axis the addends of the {lower, upper} plane pair: -(o * inv + slack) for the plane the ray meets first,
    // -(o * inv - slack) for the other one (which is which depends on the direction's sign)
    f2v cx, cy, cz;
    bool sx, sy, sz;
    bool regular;
};

__device__ __forceinline__ RayPre make_raypre(const SceneDev& S, const F3 o, const F3 d, const float t) {
    RayPre P;
    const float dmax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float big = 1.099511627776e12f, small = 9.094947017729282e-13f;  // 2^40, 2^-40
    const bool finite = (fabsf(o.x) <= big) && (fabsf(o.y) <= big) && (fabsf(o.z) <= big) && (fabsf(d.x) <= big) &&
                        (fabsf(d.y) <= big) && (fabsf(d.z) <= big);  // false for NaN and +-inf too
    const float eps = S.scene_eps + 1.52587890625e-05f * omax;  // 2^-16
    // eps >= 2^-60 keeps the slack eps * |inv| (|inv| >= 2^-40) far above the denormals' absolute rounding error
    P.regular = finite && (dmax >= small) && !(t != t) && (S.scene_eps <= 16777216.0f) && (eps >= 8.673617379884035e-19f);
    const float fl = dmax * small;
    const float dx = fabsf(d.x) >= fl ? d.x : copysignf(fl, d.x);
    const float dy = fabsf(d.y) >= fl ? d.y : copysignf(fl, d.y);
    const float dz = fabsf(d.z) >= fl ? d.z : copysignf(fl, d.z);
    P.inv = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    P.sx = dx < 0;
    P.sy = dy < 0;
    P.sz = dz < 0;
    const F3 oi = f3(o.x * P.inv.x, o.y * P.inv.y, o.z * P.inv.z);
    const F3 sl = f3(eps * fabsf(P.inv.x), eps * fabsf(P.inv.y), eps * fabsf(P.inv.z));
    const F3 oin = f3(oi.x + sl.x, oi.y + sl.y, oi.z + sl.z), oif = f3(oi.x - sl.x, oi.y - sl.y, oi.z - sl.z);
    P.cx = P.sx ? (f2v){-oif.x, -oin.x} : (f2v){-oin.x, -oif.x};
    P.cy = P.sy ? (f2v){-oif.y, -oin.y} : (f2v){-oin.y, -oif.y};
    P.cz = P.sz ? (f2v){-oif.z, -oin.z} : (f2v){-oin.z, -oif.z};
    return P;
}

// Conservative [tn, tf] of the widened box whose planes come as per-axis {lower, upper} pairs (SubNode's layout): one
// packed fma per axis, then the entry / exit plane picked by the direction's sign.  Explicit fma: this is NOT reference
// arithmetic.
__device__ __forceinline__ void slab_cons(const RayPre& P, const f2v bx, const f2v by, const f2v bz, float& tn, float& tf) {
    const f2v tx = __builtin_elementwise_fma(bx, (f2v){P.inv.x, P.inv.x}, P.cx);
    const f2v ty = __builtin_elementwise_fma(by, (f2v){P.inv.y, P.inv.y}, P.cy);
    const f2v tz = __builtin_elementwise_fma(bz, (f2v){P.inv.z, P.inv.z}, P.cz);
    const float nx = P.sx ? tx.y : tx.x, fx = P.sx ? tx.x : tx.y;
    const float ny = P.sy ? ty.y : ty.x, fy = P.sy ? ty.x : ty.y;
    const float nz = P.sz ? tz.y : tz.x, fz = P.sz ? tz.x : tz.y;
    tn = fmaxf(fmaxf(nx, ny), nz);
    tf = fminf(fminf(fx, fy), fz);
}
// the four children of a 128-byte node held in six 16-byte quarters (cgrt_layout.h SubNode: per child x, y, z pairs)
__device__ __forceinline__ void slab_cons4(const RayPre& P, const float4 a0, const float4 b0, const float4 c0, const float4 a1, const float4 b1,
                                           const float4 c1, float& tn0, float& tf0, float& tn1, float& tf1, float& tn2, float& tf2, float& tn3,
                                           float& tf3) {
    slab_cons(P, (f2v){a0.x, a0.y}, (f2v){a0.z, a0.w}, (f2v){b0.x, b0.y}, tn0, tf0);
    slab_cons(P, (f2v){b0.z, b0.w}, (f2v){c0.x, c0.y}, (f2v){c0.z, c0.w}, tn1, tf1);
    slab_cons(P, (f2v){a1.x, a1.y}, (f2v){a1.z, a1.w}, (f2v){b1.x, b1.y}, tn2, tf2);
    slab_cons(P, (f2v){b1.z, b1.w}, (f2v){c1.x, c1.y}, (f2v){c1.z, c1.w}, tn3, tf3);
}

// State of one reference leaf's scan, order-free form (see bvh_builder.cpp "In-leaf accelerator"):
// State of one reference leaf's scan, order-free form (see bvh_builder.cpp "In-leaf accelerator"):
//   regular acceptances keep the lexicographic minimum of (t, scan position), starting from the entry
//   ray.t with strict <;  origin-on-plane acceptances (t = 0 without a guard) keep the LAST scan position.
struct LeafScan {
    float best_t;
    int best_k;
    uint32_t best_rec;
    int onp_k;
    uint32_t onp_rec;
};

// intersectRayWithTriangle (ray_tracing.cpp:86-114) for the record whose four 16-byte quarters are
// a, b, c, e (already loaded), reference arithmetic.
__device__ __forceinline__ void test_record(const float4 a, const float4 b, const float4 c, const float4 e, const uint32_t rec,
                                            const F3 o, const F3 d, LeafScan& L) {
    const F3 v0 = f3(a.x, a.y, a.z), v1 = f3(a.w, b.x, b.y), v2 = f3(b.z, b.w, c.x);
    const F3 n = f3(c.y, c.z, c.w);
    const float D = e.x;
    const int k = (int)__float_as_uint(e.w);
    const float on = dot(o, n);
    if (on == D) {  // ray_tracing.cpp:43-47: t = 0, no t < ray.t guard; the last one scanned wins
        const F3 p = add(o, scale(d, 0.0f));
        if (point_in_triangle(v0, v1, v2, n, p) && k > L.onp_k) {
            L.onp_k = k;
            L.onp_rec = rec;
        }
        return;
    }
    const float den = dot(d, n);
    if (den == 0) return;
    const float tt = (D - on) / den;
    if (tt < 0) return;
    // ray_tracing.cpp:65 `t >= ray.t` against the scan's running minimum; equal t is taken only from
    // an EARLIER scan position (the reference would have met that triangle first)
    if (tt >= L.best_t && !(tt == L.best_t && k < L.best_k)) return;
    const F3 p = add(o, scale(d, tt));
    if (!point_in_triangle(v0, v1, v2, n, p)) return;
    L.best_t = tt;
    L.best_k = k;
    L.best_rec = rec;
}

// The same test in two halves, for runs of one or two records: eval_record does all the arithmetic that does not depend
// on the scan state (no early exits, so the two records of a run interleave and their loads overlap), apply_eval then
// applies intersectRayWithPlane's rejections and the scan rule in order.  Same expressions, same comparisons.
struct TriEval {
    float tt;
    int k;
    bool onp, den_ok, inside;
};
__device__ __forceinline__ TriEval eval_record(const float4 a, const float4 b, const float4 c, const float4 e, const F3 o, const F3 d) {
    const F3 v0 = f3(a.x, a.y, a.z), v1 = f3(a.w, b.x, b.y), v2 = f3(b.z, b.w, c.x);
    const F3 n = f3(c.y, c.z, c.w);
    const float D = e.x;
    TriEval E;
    E.k = (int)__float_as_uint(e.w);
    const float on = dot(o, n);
    E.onp = (on == D);
    const float den = dot(d, n);
    E.den_ok = !(den == 0);
    const float q = (D - on) / den;
    E.tt = E.onp ? 0.0f : q;  // ray_tracing.cpp:43-47: origin on the plane -> t = 0
    const F3 p = add(o, scale(d, E.tt));
    E.inside = point_in_triangle(v0, v1, v2, n, p);
    return E;
}
// Two records at once on the packed FP32 pipe (v_pk_mul_f32 / v_pk_add_f32: two IEEE single operations per issue, each
// component rounded exactly like the scalar instruction).  The expression trees are eval_record's, component for
// component -- products and sums stay separate instructions (-ffp-contract=off covers vector types), the operand order is
// glm's -- so E0/E1 carry the same bits as two eval_record calls; the division has no packed form and stays scalar.
#ifndef CGRT_PACKED_PAIR
#define CGRT_PACKED_PAIR 1
#endif
struct P3 {
    f2v x, y, z;
};
__device__ __forceinline__ f2v splat2(const float s) { return (f2v){s, s}; }
__device__ __forceinline__ P3 p3(const f2v x, const f2v y, const f2v z) {
    P3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
__device__ __forceinline__ P3 p3_pair(const F3 a, const F3 b) { return p3((f2v){a.x, b.x}, (f2v){a.y, b.y}, (f2v){a.z, b.z}); }
__device__ __forceinline__ P3 p3_splat(const F3 a) { return p3(splat2(a.x), splat2(a.y), splat2(a.z)); }
__device__ __forceinline__ P3 sub(const P3 a, const P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ P3 add(const P3 a, const P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ P3 scale(const P3 a, const f2v s) { return p3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f2v dot(const P3 a, const P3 b) {
    const f2v px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
    return (px + py) + pz;
}
__device__ __forceinline__ P3 cross(const P3 a, const P3 b) {
    return p3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ void eval_pair(const float4 a0, const float4 b0, const float4 c0, const float4 e0, const float4 a1, const float4 b1,
                                          const float4 c1, const float4 e1, const F3 o, const F3 d, TriEval& E0, TriEval& E1) {
#if CGRT_PACKED_PAIR
    const P3 v0 = p3_pair(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z));
    const P3 v1 = p3_pair(f3(a0.w, b0.x, b0.y), f3(a1.w, b1.x, b1.y));
    const P3 v2 = p3_pair(f3(b0.z, b0.w, c0.x), f3(b1.z, b1.w, c1.x));
    const P3 n = p3_pair(f3(c0.y, c0.z, c0.w), f3(c1.y, c1.z, c1.w));
    const P3 po = p3_splat(o), pd = p3_splat(d);
    const float D0 = e0.x, D1 = e1.x;
    E0.k = (int)__float_as_uint(e0.w);
    E1.k = (int)__float_as_uint(e1.w);
    const f2v on = dot(po, n);
    const f2v den = dot(pd, n);
    E0.onp = (on.x == D0);
    E1.onp = (on.y == D1);
    E0.den_ok = !(den.x == 0);
    E1.den_ok = !(den.y == 0);
    const f2v num = (f2v){D0, D1} - on;
    const float q0 = num.x / den.x, q1 = num.y / den.y;
    E0.tt = E0.onp ? 0.0f : q0;
    E1.tt = E1.onp ? 0.0f : q1;
    const P3 p = add(po, scale(pd, (f2v){E0.tt, E1.tt}));
    // pointInTriangle (ray_tracing.cpp:23-38), both records
    const P3 e01 = sub(v1, v0), e12 = sub(v2, v1), e20 = sub(v0, v2);
    const P3 q0v = sub(p, v0), q1v = sub(p, v1), q2v = sub(p, v2);
    const f2v s0 = dot(n, cross(e01, q0v)), s1 = dot(n, cross(e12, q1v)), s2 = dot(n, cross(e20, q2v));
    E0.inside = s0.x >= 0 && s1.x >= 0 && s2.x >= 0;
    E1.inside = s0.y >= 0 && s1.y >= 0 && s2.y >= 0;
#else
    E0 = eval_record(a0, b0, c0, e0, o, d);
    E1 = eval_record(a1, b1, c1, e1, o, d);
#endif
}
__device__ __forceinline__ void apply_eval(const TriEval& E, const uint32_t rec, LeafScan& L) {
    const bool onp_take = E.onp && E.inside && (E.k > L.onp_k);
    L.onp_k = onp_take ? E.k : L.onp_k;
    L.onp_rec = onp_take ? rec : L.onp_rec;
    const bool behind = (E.tt >= L.best_t) && !((E.tt == L.best_t) && (E.k < L.best_k));
    const bool take = !E.onp && E.den_ok && !(E.tt < 0) && !behind && E.inside;
    L.best_t = take ? E.tt : L.best_t;
    L.best_k = take ? E.k : L.best_k;
    L.best_rec = take ? rec : L.best_rec;
}
template <bool COUNT>
__device__ __forceinline__ void test_pair(const SceneDev& S, const uint32_t first, const uint32_t n, const F3 o, const F3 d, LeafScan& L,
                                          LaneCounters& cnt) {
    if (COUNT) {
        cnt.tri += n;
        if (first_active_lane()) cnt.w_tri++;
    }
    const float4* q = reinterpret_cast<const float4*>(S.tris + first);
    const uint32_t j = (n > 1) ? 4u : 0u;  // a run of one: the second slot re-reads the first record and is not applied
    const float4 a0 = q[0], b0 = q[1], c0 = q[2], e0 = q[3];
    const float4 a1 = q[j], b1 = q[j + 1], c1 = q[j + 2], e1 = q[j + 3];
    TriEval E0, E1;
    eval_pair(a0, b0, c0, e0, a1, b1, c1, e1, o, d, E0, E1);
    apply_eval(E0, first, L);
    if (n > 1) apply_eval(E1, first + 1, L);
}

// Tests records [first, first + n), two loads in flight (the second record's 64 bytes are requested before
// the first one is evaluated; a run is 1..32 contiguous records).
template <bool COUNT>
__device__ __forceinline__ void test_run(const SceneDev& S, const uint32_t first, const uint32_t n, const F3 o, const F3 d, LeafScan& L,
                                         LaneCounters& cnt) {
    if (COUNT) cnt.tri += n;
    const float4* q = reinterpret_cast<const float4*>(S.tris + first);
    float4 a0 = q[0], b0 = q[1], c0 = q[2], e0 = q[3];
    for (uint32_t i = 0; i < n; i++) {
        float4 a1 = a0, b1 = b0, c1 = c0, e1 = e0;
        if (i + 1 < n) {  // the next record is requested before this one is evaluated
            a1 = q[4 * i + 4];
            b1 = q[4 * i + 5];
            c1 = q[4 * i + 6];
            e1 = q[4 * i + 7];
        }
        if (COUNT && first_active_lane()) cnt.w_tri++;
        test_record(a0, b0, c0, e0, first + i, o, d, L);
        a0 = a1;
        b0 = b1;
        c0 = c1;
        e0 = e1;
    }
}

// Run reference inside the in-leaf accelerator: REF_LEAF | (count - 1) << 26 | first record.
__device__ __forceinline__ uint32_t run_first(const uint32_t r) { return r & 0x03ffffffu; }
__device__ __forceinline__ uint32_t run_count(const uint32_t r) { return ((r >> 26) & 31u) + 1u; }

// One step through a 128-byte node of the in-leaf accelerator (four child boxes): the nearest hit child becomes `cur`,
// the other hit children are deferred far-to-near so that they pop near-to-far.  best_t = the scan's running minimum.
template <bool COUNT>
__device__ __forceinline__ void sub_node_step(const SceneDev& S, const RayPre& P, const float best_t, uint32_t& cur, int& sp,
                                              uint32_t* __restrict__ stk, LaneCounters& cnt) {
        if (COUNT) {
            cnt.sub++;
            if (first_active_lane()) cnt.w_sub++;
        }
        const float4* q = reinterpret_cast<const float4*>(S.subnodes + cur);
        const float4 a0 = q[0], b0 = q[1], c0 = q[2];
        const uint4 m = *reinterpret_cast<const uint4*>(q + 3);  // the four child references (cgrt_layout.h SubNode)
        const float4 a1 = q[4], b1 = q[5], c1 = q[6];
        float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
        slab_cons4(P, a0, b0, c0, a1, b1, c1, tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3);
        // Bound for culling: the running minimum, but never below 0 -- an origin-on-plane acceptance
        // ignores ray.t altogether (ray_tracing.cpp:43-47) and its box contains the origin (tn < 0).
        const float tc = fmaxf(best_t, 0.0f);
        const float inf = __builtin_inff();
        // sort key: entry parameter of a hit child, +inf for a missed (or absent: empty box) one
        float k0 = ((tn0 <= tf0) && (tf0 >= 0.0f) && (tn0 <= tc)) ? tn0 : inf;
        float k1 = ((tn1 <= tf1) && (tf1 >= 0.0f) && (tn1 <= tc)) ? tn1 : inf;
        float k2 = ((tn2 <= tf2) && (tf2 >= 0.0f) && (tn2 <= tc)) ? tn2 : inf;
        float k3 = ((tn3 <= tf3) && (tf3 >= 0.0f) && (tn3 <= tc)) ? tn3 : inf;
        uint32_t r0 = m.x, r1 = m.y, r2 = m.z, r3 = m.w;
#define CGRT_CSWAP(ka, ra, kb, rb)          \
{                                       \
    const bool sw = kb < ka;            \
    const float kt = sw ? kb : ka;      \
    const uint32_t rt = sw ? rb : ra;   \
    kb = sw ? ka : kb;                  \
    rb = sw ? ra : rb;                  \
    ka = kt;                            \
    ra = rt;                            \
}
        CGRT_CSWAP(k0, r0, k1, r1)
        CGRT_CSWAP(k2, r2, k3, r3)
        CGRT_CSWAP(k0, r0, k2, r2)
#ifndef CGRT_SORT_FULL
#define CGRT_SORT_FULL 1  // 0: only the nearest child is identified, the deferred ones keep their stored order
#endif
#if CGRT_SORT_FULL
        CGRT_CSWAP(k1, r1, k3, r3)
        CGRT_CSWAP(k1, r1, k2, r2)
#endif
#undef CGRT_CSWAP
        // Branch-free pushes: the slot is always written and only kept (sp advanced) when the child was hit; a
        // level defers at most three children, so the writes stay inside the lane's SUB_STACK_ENTRIES slots.
        stk[sp * CGRT_STRIDE] = r3;
        sp += (k3 < inf) ? 1 : 0;
        stk[sp * CGRT_STRIDE] = r2;
        sp += (k2 < inf) ? 1 : 0;
        stk[sp * CGRT_STRIDE] = r1;
        sp += (k1 < inf) ? 1 : 0;
        cur = (k0 < inf) ? r0 : REF_NONE;
    }
// intersectLeaf (bvh.cpp:535-553) for one ray.  "while-while": lanes first step through accelerator nodes
// until each stands on a run of triangles (or has nothing left), then the runs are tested together.
// Stack entries are bare references (carrying the entry parameter for culling on pop was measured: no gain).
template <bool COUNT>
__device__ __forceinline__ void scan_leaf(const SceneDev& S, const LeafRec LR, const F3 o, const F3 d, const RayPre& P, float& t,
                                          uint32_t& hit_rec, uint32_t* __restrict__ stk, const int sp0, LaneCounters& cnt) {
    LeafScan L;
    L.best_t = t;
    L.best_k = -1;
    L.best_rec = REF_NONE;
    L.onp_k = -1;
    L.onp_rec = REF_NONE;
    if (LR.sub_root == REF_NONE || !P.regular) {
        for (uint32_t i = 0; i < LR.count; i += 32) test_run<COUNT>(S, LR.first + i, min(32u, LR.count - i), o, d, L, cnt);
    } else {
        int sp = sp0;
        uint32_t cur = LR.sub_root;
        for (;;) {
            // ---- node phase ----
            while (cur != REF_NONE && !(cur & REF_LEAF)) sub_node_step<COUNT>(S, P, L.best_t, cur, sp, stk, cnt);
            // ---- triangle phase ----
            if (cur != REF_NONE) test_run<COUNT>(S, run_first(cur), run_count(cur), o, d, L, cnt);
            // ---- pop ----
            if (sp <= sp0) break;
            sp -= 1;
            cur = stk[sp * CGRT_STRIDE];
        }
    }
    if (L.onp_k >= 0) {
        t = 0.0f;
        hit_rec = L.onp_rec;
    } else if (L.best_k >= 0) {
        t = L.best_t;
        hit_rec = L.best_rec;
    }
}
// ---------------------------------------------------------------------------------------------
// Exact fast path for the reference's own box test (ray_tracing.cpp:162-200).
//
// That test needs twelve IEEE divisions per inner node, (box.lower - o) / d and (box.upper - o) / d,
// and their ROUNDED values decide culling and visit order, so they must be the correctly rounded
// quotients -- an approximate reciprocal is not an option.  hipcc's generic a / b costs ~11 VALU
// (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup).  For a fixed denominator the quotient can
// be had in 4: with yh = RN(1/d) (one true division per ray and axis) and yl ~ 1/d - yh,
//     q0 = RN(a * yh);  q1 = RN(a * yl + q0)          -> q1 is a faithful rounding of a/d (< 1 ulp)
//     r  = RN(a - d * q1)  (exact, one fma);  q = RN(q1 + r * yh)
// and Markstein's theorem (IBM J. R&D 34(1), 1990; Muller et al., Handbook of FP Arithmetic, ch. 4:
// "y correctly rounded reciprocal, q1 faithful  =>  RN(q1 + r*y) = RN(a/d)") makes q the correctly
// rounded quotient, provided nothing overflows, underflows or is subnormal on the way.  RayFast::fd
// says that this holds for every box coordinate of the scene and this ray: direction components in
// [2^-36, 2^36], origin and box coordinates zero or in [2^-40, 2^40] (then every numerator is 0 or in
// [2^-64, 2^41] and every quotient 0 or in [2^-100, 2^77]).  Under fd no quotient is NaN either, so the
// reference's `a < b ? a : b` ladders equal min/max up to the sign of a zero, which no comparison sees.
// Rays outside that envelope (a zero direction component is enough) take the generic path, same bits.
// tests/test_parity_gpu.py::test_fast_division_is_ieee hammers the 4-op form against a / d on the device.
struct RayFast {
    F3 yh, yl;
    bool fd;
    bool in_root;  // the origin is strictly inside the root box.  Every node box is the bound of a subset of the root's
                   // vertices (exact min/max), hence inside the root box: an origin that is not strictly inside the root
                   // box is strictly inside no node box, and startsInBox (bvh.cpp:647-661) is false without looking.
};

__device__ __forceinline__ float fdiv4(const float a, const float d, const float yh, const float yl) {
    const float q0 = a * yh;
    const float q1 = __builtin_fmaf(a, yl, q0);
    const float r = __builtin_fmaf(-d, q1, a);
    return __builtin_fmaf(r, yh, q1);
}

__device__ __forceinline__ bool in_fast_range(const float x, const float lo, const float hi) {
    const float a = fabsf(x);
    return (a >= lo) && (a <= hi);  // false for NaN
}

__device__ __forceinline__ RayFast make_rayfast(const SceneDev& S, const F3 o, const F3 d) {
    RayFast R;
    const float dlo = 1.4551915228366852e-11f, dhi = 68719476736.0f;          // 2^-36, 2^36
    const float clo = 9.094947017729282e-13f, chi = 1099511627776.0f;          // 2^-40, 2^40
    const bool dok = in_fast_range(d.x, dlo, dhi) && in_fast_range(d.y, dlo, dhi) && in_fast_range(d.z, dlo, dhi);
    const bool ook = (o.x == 0.0f || in_fast_range(o.x, clo, chi)) && (o.y == 0.0f || in_fast_range(o.y, clo, chi)) &&
                     (o.z == 0.0f || in_fast_range(o.z, clo, chi));
    R.fd = dok && ook && (S.fast_boxes != 0);
    R.in_root = starts_in_box(o, f3(S.root_box.lo[0], S.root_box.lo[1], S.root_box.lo[2]), f3(S.root_box.hi[0], S.root_box.hi[1], S.root_box.hi[2]));
    const float dx = R.fd ? d.x : 1.0f, dy = R.fd ? d.y : 1.0f, dz = R.fd ? d.z : 1.0f;
    R.yh = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);  // true divisions: RN(1/d)
    R.yl = f3(__builtin_fmaf(-dx, R.yh.x, 1.0f) * R.yh.x, __builtin_fmaf(-dy, R.yh.y, 1.0f) * R.yh.y,
              __builtin_fmaf(-dz, R.yh.z, 1.0f) * R.yh.z);
    return R;
}

// ray_box + starts_in_box for one child box under RayFast::fd.  `inside` is bvh.cpp:647-661
// (lo < o  <=>  lo - o < 0 exactly, denormals being preserved).
template <bool INSIDE>
__device__ __forceinline__ bool ray_box_fast(const F3 lo, const F3 hi, const F3 o, const F3 d, const RayFast& R, const float t,
                                             float& tbox, bool& inside) {
    const float ax0 = lo.x - o.x, ay0 = lo.y - o.y, az0 = lo.z - o.z;
    const float ax1 = hi.x - o.x, ay1 = hi.y - o.y, az1 = hi.z - o.z;
    const float x0 = fdiv4(ax0, d.x, R.yh.x, R.yl.x), x1 = fdiv4(ax1, d.x, R.yh.x, R.yl.x);
    const float y0 = fdiv4(ay0, d.y, R.yh.y, R.yl.y), y1 = fdiv4(ay1, d.y, R.yh.y, R.yl.y);
    const float z0 = fdiv4(az0, d.z, R.yh.z, R.yl.z), z1 = fdiv4(az1, d.z, R.yh.z, R.yl.z);
    const float tIn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float tOut = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    inside = INSIDE && (fmaxf(fmaxf(ax0, ay0), az0) < 0.0f) && (fminf(fminf(ax1, ay1), az1) > 0.0f);
    const float cur = (tIn < 0.0f) ? tOut : tIn;
    tbox = cur;
    return !((tIn > tOut) || (tOut < 0.0f)) && !(cur >= t);
}

// Ordered closest-hit walk of the reference tree for one ray, resumable: walk_begin() runs the root gate,
// walk_round() advances the ray by one "round" -- inner nodes of the reference tree until the lane stands on a
// leaf (or has nothing left: returns true), then that leaf's scan.  "while-while": inside a round a lane never
// waits on another lane's leaf scan to take an inner step and vice versa.  The plain kernels loop over rounds
// until done; the persistent kernel refills finished lanes with new rays between rounds.
struct Walk {
    F3 o, d;
    float t;           // ray.t
    uint32_t hit_rec;  // record of the last accepted triangle (REF_NONE if none)
    uint32_t cur;      // reference-tree node the lane stands on (REF_NONE: pop)
    int sp;
    RayPre P;
    RayFast R;
};

// intersectDataStructure, bvh.cpp:831-844.  Returns false when the ray does not enter the tree at all.
__device__ __forceinline__ bool walk_begin(const SceneDev& S, Walk& W) {
    W.hit_rec = REF_NONE;
    W.cur = REF_NONE;
    W.sp = 0;
    if (S.root_ref == REF_NONE) return false;
    const F3 lo = f3(S.root_box.lo[0], S.root_box.lo[1], S.root_box.lo[2]);
    const F3 hi = f3(S.root_box.hi[0], S.root_box.hi[1], S.root_box.hi[2]);
    float tb;
    // (the box test's write to ray.t is undone by the reference, :838)
    if (!(starts_in_box(W.o, lo, hi) || ray_box(lo, hi, W.o, W.d, W.t, tb))) return false;
    W.cur = S.root_ref;
    W.P = make_raypre(S, W.o, W.d, W.t);
    W.R = make_rayfast(S, W.o, W.d);
    return true;
}

// intersectNonLeaf (bvh.cpp:715-736) for the inner node `cur`: both child boxes are tested, the child the reference
// enters first becomes `cur` (REF_NONE if neither is entered), the other one is deferred with its box parameter.
template <bool COUNT>
__device__ __forceinline__ void topo_step(const SceneDev& S, const Walk& W, const F3 o, const F3 d, uint32_t& cur, int& sp,
                                          uint32_t* __restrict__ stk, LaneCounters& cnt) {
    // intersectNonLeaf, bvh.cpp:715-736
    if (COUNT) {
        cnt.inner++;
        if (first_active_lane()) cnt.w_inner++;
    }
    const float4* q = reinterpret_cast<const float4*>(S.packets + cur);
    const float4 a = q[0], b = q[1], c = q[2];
    const uint4 m = *reinterpret_cast<const uint4*>(q + 3);
    const F3 llo = f3(a.x, a.y, a.z), lhi = f3(a.w, b.x, b.y);
    const F3 rlo = f3(b.z, b.w, c.x), rhi = f3(c.y, c.z, c.w);
    // Which child is entered first, which one is deferred (bvh.cpp:679-701 intersectDeeper, :611-635
    // intersectRayThatStartsOutsideBoxes), as selects:
    //   a child is visited iff the origin is strictly inside its box or its box test succeeded (not the -1 sentinel);
    //   left goes first when the origin is inside it (:685-692), or -- origin inside neither -- when only left was hit
    //   or both were and tL < tR (:626-633); the deferred child carries its own box parameter as tSecond, except when
    //   the origin is inside both boxes: then right is visited unconditionally (:685-688), i.e. tSecond = -inf.
    auto order_and_push = [&](const float tL, const float tR, const bool inL, const bool inR) __attribute__((always_inline)) {
        const bool hitL = !(tL < 0), hitR = !(tR < 0);
        const bool wantL = inL || hitL, wantR = inR || hitR;
        const bool lfirst = wantL && (inL || !wantR || (!inR && tL < tR));
        const uint32_t first = lfirst ? m.x : (wantR ? m.y : REF_NONE);
        const uint32_t second = (wantL && wantR) ? (lfirst ? m.y : m.x) : REF_NONE;
        const float tsec = (inL && inR) ? -__builtin_inff() : (lfirst ? tR : tL);
        // branch-free push: the two slots above sp are always written and only kept when a child was deferred
        // (at most MAX_LEVELS - 1 deferred children exist at any time, so the slots are inside the lane's slice)
        stk[sp * CGRT_STRIDE] = second;
        stk[(sp + 1) * CGRT_STRIDE] = __float_as_uint(tsec);
        sp += (second != REF_NONE) ? 2 : 0;
        cur = first;
    };
    float tL = -1.0f, tR = -1.0f, tb;
    bool inL, inR;
    if (W.R.fd) {
        if (__any(W.R.in_root)) {
            if (ray_box_fast<true>(llo, lhi, o, d, W.R, W.t, tb, inL)) tL = tb;
            if (ray_box_fast<true>(rlo, rhi, o, d, W.R, W.t, tb, inR)) tR = tb;
            order_and_push(tL, tR, inL, inR);
        } else {  // no origin of this wave is inside the root box (RayFast::in_root): no inside tests, simpler ordering
            if (ray_box_fast<false>(llo, lhi, o, d, W.R, W.t, tb, inL)) tL = tb;
            if (ray_box_fast<false>(rlo, rhi, o, d, W.R, W.t, tb, inR)) tR = tb;
            order_and_push(tL, tR, false, false);
        }
    } else {
        if (ray_box(llo, lhi, o, d, W.t, tb)) tL = tb;
        if (ray_box(rlo, rhi, o, d, W.t, tb)) tR = tb;
        inL = starts_in_box(o, llo, lhi);
        inR = starts_in_box(o, rlo, rhi);
        order_and_push(tL, tR, inL, inR);
    }
}

// Pops deferred children until one is still wanted (bvh.cpp:582: a deferred child is skipped iff ray.t < tSecond).
__device__ __forceinline__ bool topo_pop(const float t, uint32_t& cur, int& sp, const uint32_t* __restrict__ stk) {
    while (sp > 0) {
        sp -= 2;
        const uint32_t r = stk[sp * CGRT_STRIDE];  // both words in one LDS access (ds_read2st64_b32)
        const float ts = __uint_as_float(stk[(sp + 1) * CGRT_STRIDE]);
        asm volatile("" : : "v"(r), "v"(ts));  // keeps the pair together: the reference is wanted whenever ts passes
        if (!(t < ts)) {
            cur = r;
            return true;
        }
    }
    return false;
}

// LeafRec of a leaf reference in either encoding (cgrt_layout.h REF_LEAF_ACCEL).
__device__ __forceinline__ LeafRec leaf_rec_of(const SceneDev& S, const uint32_t ref) {
    const uint32_t li = (ref & REF_LEAF_ACCEL) ? S.subnodes[(ref & REF_INDEX26) + 1u].pad[0] : (ref & ~REF_LEAF);
    return S.leaves[li];
}
// "One loop" form of the same walk: an iteration offers every lane, in this order, two topology steps (pops included),
// the entry into a leaf, two accelerator node steps, one run test and the pop inside the leaf; a lane takes the pieces
// its state asks for and never waits for the other lanes to finish their topology phase or their leaf.  The while-while
// form (walk_round) runs every phase to completion for all lanes of the wave, which serialises waves whose rays are out
// of phase: its hardest waves execute 4x the node steps of their hardest ray (profiles/r1_step4_wave_anatomy.txt).
// Same steps, same arithmetic, same order per ray; only the interleaving between lanes differs.
template <bool COUNT, bool ANYHIT>
__device__ __forceinline__ void walk_tree_unified(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    const F3 o = W.o, d = W.d;
    uint32_t cur = W.cur;      // reference-tree node (topology mode)
    uint32_t scur = REF_NONE;  // accelerator node or run (leaf mode)
    int sp = W.sp, sp0 = -1;   // sp0 >= 0: the lane is scanning a leaf whose accelerator stack starts at sp0
    LeafScan L;
    L.best_t = W.t;
    L.best_k = -1;
    L.best_rec = REF_NONE;
    L.onp_k = -1;
    L.onp_rec = REF_NONE;
    bool done = false;
    // The pieces of an iteration; each acts only on lanes whose state asks for it.
    auto T = [&]() __attribute__((always_inline)) {  // topology: pop if needed, one intersectNonLeaf step
        if (sp0 < 0 && !done) {
            if (cur == REF_NONE && !topo_pop(W.t, cur, sp, stk)) done = true;
            if (!done && !(cur & REF_LEAF)) topo_step<COUNT>(S, W, o, d, cur, sp, stk, cnt);
        }
    };
    auto E = [&]() __attribute__((always_inline)) {  // enter the leaf the lane stands on
        if (sp0 < 0 && !done && cur != REF_NONE && (cur & REF_LEAF)) {
            if (COUNT) cnt.leaf++;
            L.best_t = W.t;
            L.best_k = -1;
            L.best_rec = REF_NONE;
            L.onp_k = -1;
            L.onp_rec = REF_NONE;
            if (!(cur & REF_LEAF_ACCEL) || !W.P.regular) {
                // leaves without accelerator and rays outside its envelope: the whole leaf at once (scan_leaf)
                scan_leaf<COUNT>(S, leaf_rec_of(S, cur), o, d, W.P, W.t, W.hit_rec, stk, sp, cnt);
                if (ANYHIT && W.hit_rec != REF_NONE) done = true;
            } else {  // the reference IS the accelerator's root: nothing to load
                scur = cur & REF_INDEX26;
                sp0 = sp;
            }
            cur = REF_NONE;
        }
    };
    auto N = [&]() __attribute__((always_inline)) {  // one accelerator node step
        if (sp0 >= 0 && scur != REF_NONE && !(scur & REF_LEAF)) sub_node_step<COUNT>(S, W.P, L.best_t, scur, sp, stk, cnt);
    };
    auto R = [&]() __attribute__((always_inline)) {  // one run of triangles
        if (sp0 >= 0 && scur != REF_NONE && (scur & REF_LEAF)) {
            if (SUB_LEAF_TRIS <= 2 && run_count(scur) <= 2)
                test_pair<COUNT>(S, run_first(scur), run_count(scur), o, d, L, cnt);
            else
                test_run<COUNT>(S, run_first(scur), run_count(scur), o, d, L, cnt);
            scur = REF_NONE;
        }
    };
    auto P = [&]() __attribute__((always_inline)) {  // next deferred child of the leaf, or the leaf is finished
        if (sp0 >= 0 && scur == REF_NONE) {
            if (sp > sp0) {
                sp -= 1;
                scur = stk[sp * CGRT_STRIDE];
            } else {  // commit the scan (intersectLeaf's outcome) and return to the topology
                if (L.onp_k >= 0) {
                    W.t = 0.0f;
                    W.hit_rec = L.onp_rec;
                } else if (L.best_k >= 0) {
                    W.t = L.best_t;
                    W.hit_rec = L.best_rec;
                }
                sp0 = -1;
                if (ANYHIT && W.hit_rec != REF_NONE) done = true;
            }
        }
    };
#ifndef CGRT_EXACT_WAVE_LOOP
#define CGRT_EXACT_WAVE_LOOP 0  // 1: wave-level loop condition, measured 4 % slower for this walk (profiles/r2_exp_exact_wave_loop.txt)
#endif
#if CGRT_EXACT_WAVE_LOOP
    while (__any(!done)) {  // the loop condition is the wave's; the pieces predicate themselves on the lane's state
        T(); T(); E(); N(); N(); R(); P();
    }
#else
    while (!done) {
        // (sequences with more or fewer pieces per iteration measured slower: profiles/r1_exp_one_loop.txt)
        T(); T(); E(); N(); N(); R(); P();
    }
#endif
}

// Spheres (bvh.cpp:878-879), result assembly and the accepted hit's interpolated normal
// (ray_tracing.cpp:94-107), which depends only on the final (triangle, t).
__device__ __forceinline__ void resolve_hit(const SceneDev& S, const F3 o, const F3 d, float t, uint32_t hit_rec, const bool want_normal,
                                            CgrtHitDev& h, F3& nn) {
    uint32_t prim = 0xffffffffu;
    int32_t mat = -1;
    bool hit = false;
    if (hit_rec != REF_NONE) {
        const TriRecord* T = S.tris + hit_rec;
        prim = T->prim_id;
        mat = (int32_t)T->mesh_id;
        hit = true;
    }
    bool sphere_last = false;
    F3 sn = f3(0, 0, 0);
    for (uint32_t s = 0; s < S.nspheres; s++) {
        const SphereRecord sp = S.spheres[s];
        if (ray_sphere(f3(sp.c[0], sp.c[1], sp.c[2]), sp.radius, o, d, t, sn)) {
            prim = S.ntris + s;
            hit = true;
            sphere_last = true;
        }
    }
    h.t = t;
    h.prim_id = prim;
    h.material_id = mat;
    h.hit = hit ? 1u : 0u;
    nn = sn;
    if (want_normal && hit && !sphere_last) {
        const TriRecord* T = S.tris + hit_rec;
        const TriNormals* N = S.tri_normals + (hit_rec - S.tri_base);
        nn = hit_normal(ld3(T->v0), ld3(T->v1), ld3(T->v2), ld3(T->n), ld3(N->n1), ld3(N->n2), ld3(N->n3), o, d, t);
    }
}
__device__ __forceinline__ void finish_ray(const SceneDev& S, const F3 o, const F3 d, float t, uint32_t hit_rec, CgrtHitDev* out,
                                           float* out_normal) {
    CgrtHitDev h;
    F3 nn;
    resolve_hit(S, o, d, t, hit_rec, out_normal != nullptr, h, nn);
    *out = h;
    if (out_normal && h.hit) {
        out_normal[0] = nn.x;
        out_normal[1] = nn.y;
        out_normal[2] = nn.z;
    }
}

__device__ __forceinline__ void flush_counters(const LaneCounters& c, bool active, unsigned long long* g) {
    // wave reduction, then one atomic per wave and counter
    unsigned long long v[8] = {active ? 1ull : 0ull, c.inner, c.leaf, c.tri, c.sub, c.cert, c.fallback, c.entered};
    for (int k = 0; k < 8; k++) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(g + k, x);
    }
}

// Trackball::generateRay (trackball.cpp:92-103) for pixel (x, y): ndc as main.cpp:691-693.
__device__ __forceinline__ void primary_ray(const CameraDev& C, int W, int H, int x, int y, F3& o, F3& d) {
    const float px = float(x) / float(W) * 2.0f - 1.0f;
    const float py = float(y) / float(H) * 2.0f - 1.0f;
    const F3 cam = normalize(f3(-px * C.half_w, py * C.half_h, 1.0f));
    Q4 q;
    q.w = C.q[0];
    q.x = C.q[1];
    q.y = C.q[2];
    q.z = C.q[3];
    d = quat_rotate(q, cam);
    o = f3(C.pos[0], C.pos[1], C.pos[2]);
}

// Workgroup -> super-tile -> tile -> pixel (FrameDev in cgrt_layout.h): blockIdx % 8 selects the XCD lane of
// the rank's super-tile list, 16 consecutive workgroups of that lane cover one 64x64 super-tile, the 4 waves
// of a workgroup take 4 horizontally adjacent 8x8 tiles.
__device__ __forceinline__ bool tile_pixel_of(const FrameDev& F, const uint32_t b, const uint32_t tid, int& x, int& y) {
    const uint32_t lane8 = b & 7u, j = b >> 3;
    const int lane = (int)(tid & 63u);
    const uint32_t wpb = blockDim.x >> 6, bps = 64u / wpb;  // waves per workgroup, workgroups per super-tile
    const uint32_t s = (j / bps) * 8u + lane8;  // rank-local super-tile
    if (s >= F.nst_rank) return false;
    const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * s;
    const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
    const int idx = (int)((j % bps) * wpb + (tid >> 6));
    const int tx = stx * ST_TILES + (idx & 7), ty = sty * ST_TILES + (idx >> 3);
    x = F.x0 + tx * 8 + (lane & 7);
    y = F.y0 + ty * 8 + (lane >> 3);
    return x < F.x1 && y < F.y1;
}
// ---- frame hints (cgrt_layout.h HintDev) ----
// What hint_finish needs is parked in the workgroup's LDS (CGRT_HINT_SCRATCH; hinted launches have single-wave workgroups) by lane
// 0: carried in scalar registers through the walk it cost the frame kernel 12 B of scratch.
//   [0] s_memrealtime at the wave's start (low word)   [1] tile | hard << 31, 0xffffffff: the wave traces nothing (or no hints)
//   [2] generation to stamp   [3], [4] the HintDev's address   [5] the time below which no wave is hard
__device__ __forceinline__ void hint_park(uint32_t* s_hint, uint32_t t0, uint32_t tile_hard, uint32_t wgen, const HintDev* Hn, uint32_t thr_min) {
    if ((threadIdx.x & 63u) == 0u) {
        s_hint[0] = t0;
        s_hint[1] = tile_hard;
        s_hint[2] = wgen;
        s_hint[3] = (uint32_t)(uintptr_t)Hn;
        s_hint[4] = (uint32_t)((uintptr_t)Hn >> 32);
        s_hint[5] = thr_min;
    }
}
// The pixel of this lane in a hinted launch of single-wave workgroups.
__device__ __forceinline__ bool hinted_tile_pixel(const FrameDev& F, int& x, int& y, uint32_t* s_hint) {
    const HintDev* __restrict__ Hn = F.hint;
    const uint32_t* __restrict__ flag_r = Hn->flag_r;
    const uint32_t* __restrict__ list_r = Hn->list_r;
    uint32_t b = blockIdx.x;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    const uint32_t per = Hn->per_tile;
    const uint32_t cap = per == 4u ? (F.hint_blocks >> 2) : F.hint_blocks;  // entries this launch can take
    uint32_t nh = 0, listed = 0;
    if (F.hint_rgen) {
        listed = *Hn->count_r;
        nh = listed < cap ? listed : cap;
    }
    if (b == 0 && threadIdx.x == 0) {
        *Hn->count_z = 0u;
        // the threshold follows the share of the tiles that was listed (HintDev), and the host hears about it
        uint32_t thr = Hn->ctl[0];
        if (F.hint_rgen) {
            uint32_t up = 0, down = 0;
            if (listed > 2u * Hn->hi)
                up = thr >> 2;
            else if (listed > Hn->hi)
                up = thr >> 3;
            else if (2u * listed < Hn->lo)
                down = thr >> 3;
            else if (listed < Hn->lo)
                down = thr >> 4;
            thr = thr + up < Hn->thr_ceil ? thr + up : Hn->thr_ceil;
            thr = thr - down > Hn->thr_floor ? thr - down : Hn->thr_floor;
            Hn->ctl[0] = thr;
        }
        volatile uint32_t* mb = Hn->mailbox;
        mb[1] = listed;
        mb[2] = thr;
        __threadfence_system();
        mb[0] = F.hint_wgen;
    }
    const uint32_t ntiles = (uint32_t)F.tiles_x * (uint32_t)F.tiles_y;
    uint32_t tile = 0xffffffffu;
    bool in = false;
    if (b < F.hint_blocks) {  // a workgroup of the hard list
        const uint32_t i = per == 4u ? (b >> 2) : b, q = per == 4u ? (b & 3u) : 0u;
        if (i < nh) {
            const uint32_t cand = list_r[i];
            if (cand < ntiles && flag_r[cand] == ((F.hint_rgen << 16) | (i + 1u))) {  // (else not a consistent pair: the regular workgroup traces it)
                tile = cand | 0x80000000u;
                const int tx = (int)(cand % (uint32_t)F.tiles_x), ty = (int)(cand / (uint32_t)F.tiles_x);
                x = F.x0 + tx * 8 + (lane & 7);
                y = F.y0 + ty * 8 + (per == 4u ? 2 * (int)q : 0) + (lane >> 3);  // 16 rays per wave: tile rows 2q and 2q + 1 on lanes 0..15
                in = (per != 4u || lane < 16) && x < F.x1 && y < F.y1;
            }
        }
    } else {  // a regular workgroup: tile_pixel_of's mapping for single-wave workgroups
        b -= F.hint_blocks;
        const uint32_t lane8 = b & 7u, j = b >> 3;
        const uint32_t s = (j >> 6) * 8u + lane8;  // rank-local super-tile
        const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * s;
        const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
        const int idx = (int)(j & 63u);
        const int tx = stx * ST_TILES + (idx & 7), ty = sty * ST_TILES + (idx >> 3);
        if (s < F.nst_rank && tx < F.tiles_x && ty < F.tiles_y) {
            const uint32_t cand = (uint32_t)ty * (uint32_t)F.tiles_x + (uint32_t)tx;
            bool taken = false;
            if (F.hint_rgen) {
                const uint32_t v = flag_r[cand], slot = v & 0xffffu;
                taken = (v >> 16) == F.hint_rgen && slot != 0u && slot - 1u < nh && list_r[slot - 1u] == cand;  // a hard workgroup has it
            }
            if (!taken) {
                tile = cand;
                x = F.x0 + tx * 8 + (lane & 7);
                y = F.y0 + ty * 8 + (lane >> 3);
                in = x < F.x1 && y < F.y1;
            }
        }
    }
    hint_park(s_hint, t0, tile, F.hint_wgen, Hn, Hn->thr_min);
    return in;
}
// The wave's wall time decides whether its tile is on the next frame's hard list (a tile that is hard stays hard down to 3/4 of
// the threshold: the time of a wave depends on what runs beside it, and a tile at the threshold would change sides every frame).
__device__ __forceinline__ void hint_finish(const uint32_t* s_hint) {
    if ((threadIdx.x & 63u) != 0u) return;
    const uint32_t tile_hard = s_hint[1];
    if (tile_hard == 0xffffffffu) return;
    const uint32_t cost = (uint32_t)__builtin_amdgcn_s_memrealtime() - s_hint[0];
    if (cost < s_hint[5]) return;  // (nearly every wave)
    const HintDev* __restrict__ Hn = reinterpret_cast<const HintDev*>((uintptr_t)s_hint[3] | ((uintptr_t)s_hint[4] << 32));
    const bool hard = (tile_hard >> 31) != 0u;
    const uint32_t tile = tile_hard & 0x7fffffffu, wgen = s_hint[2];
    const bool wave16 = hard && Hn->per_tile == 4u;
    uint32_t thr = Hn->ctl[0];
    if (wave16) thr = (thr * 5u) / 9u;  // a 16-ray wave of a listed tile (64 hard rays: 97 us as one wave, 51 us as four)
    if (hard) thr -= thr >> 2;
    if (cost < thr) return;
    const uint32_t wg = wgen << 16;
    uint32_t* flag = Hn->flag_w + tile;
    if (wave16) {  // four waves per tile: the first one that is long claims it
        const uint32_t v = *flag;
        if ((v >> 16) == wgen || atomicCAS(flag, v, wg | 0xffffu) != v) return;
    }
    const uint32_t i = atomicAdd(Hn->count_w, 1u);
    if (i < Hn->cap) {
        Hn->list_w[i] = tile;
        *flag = wg | (i + 1u);
    } else {
        *flag = wg;  // the list is full: the tile stays a regular one
    }
}

__device__ __forceinline__ bool tile_pixel(const FrameDev& F, int lane, int& x, int& y) {
    (void)lane;
    return tile_pixel_of(F, blockIdx.x, threadIdx.x, x, y);
}

}  // namespace cgrt
