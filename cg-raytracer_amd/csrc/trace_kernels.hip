// trace_kernels.hip -- gfx950 kernels of the hot path: BoundingVolumeHierarchy::intersect
// (src/bounding_volume_hierarchy.cpp:850-881) for batches of rays, with optional fused primary-ray
// generation (src/main.cpp:691-694 + framework/src/trackball.cpp:92-103).
//
// One ray per lane.  The reference's recursion (intersectRecursive -> intersectNonLeaf ->
// intersectDeeper -> intersectRayThatStartsOutsideBoxes -> intersectChildrenHierarchically,
// bvh.cpp:572-758) is restated as an ordered stack walk: the child the reference would enter first is
// followed immediately, the other one is pushed together with its box parameter tSecond and is skipped
// on pop iff ray.t < tSecond -- the reference's `hitFirst && ray.t < tSecond` (bvh.cpp:581-585), since
// ray.t only changes when a triangle is accepted.  A child whose box test failed (t = -1) is dropped,
// and an origin strictly inside both child boxes visits both unconditionally (:685-688).
// Inside a reference leaf the linear scan of intersectLeaf (bvh.cpp:535-553) is replaced by an order-free
// but outcome-identical evaluation over a per-leaf 4-wide BVH (scan_leaf below, DESIGN.md "In-leaf accelerator").
// The per-lane stack (<= 11 deferred children, bvh.cpp:48, plus <= 15 in-leaf entries) lives in LDS,
// lane-interleaved so that every access is bank-conflict free; runtime-indexed register arrays would go
// to scratch.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no FMA contraction), default IEEE div/sqrt,
// denormals on -- see cgrt_math.h for why.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "trace_kernels.h"

namespace cgrt {

#define CGRT_BLOCK 256
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
#define CGRT_LB __launch_bounds__(CGRT_BLOCK) __attribute__((amdgpu_waves_per_eu(1, CGRT_MAX_WAVES)))
#endif
// Per-lane LDS stack, in 4-byte slots: a deferred reference child takes 2 slots (ref, tSecond), at most
// MAX_LEVELS-1 of them; an in-leaf accelerator entry takes 1 slot (ref), at most SUB_STACK_ENTRIES.
#define CGRT_STACK_SLOTS (2 * (MAX_LEVELS - 1) + SUB_STACK_ENTRIES)

#ifndef CGRT_QUAD
#define CGRT_QUAD 0  // 1: lane-cooperative leaf scans for waves with few live rays (scan_leaves_quad).  Bit-identical results;
                     // measured: slowest wave -10 %, but 154 VGPRs drop the kernel to 3 waves per SIMD and the frame gets 4 % slower
#endif
#ifndef CGRT_QUAD_MAX_RAYS
#define CGRT_QUAD_MAX_RAYS 16
#endif
#ifndef CGRT_SORT_FULL
#define CGRT_SORT_FULL 1  // 0: only the nearest hit child of a 4-wide node is identified, the deferred ones are pushed unsorted
#endif
#ifndef CGRT_STAMP_SUB
#define CGRT_STAMP_SUB 0  // diagnostic build: in-loop s_memtime stamps of the accelerator node step (load wait vs compute)
#endif
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
struct LaneCounters {
    unsigned long long c_wait = 0, c_comp = 0, c_tri = 0;  // CGRT_STAMP_SUB only
    unsigned long long c_piece[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // CGRT_STAMP_SUB, one-loop walk: cycles in T T E N N R P, loop trips
    uint32_t inner = 0, leaf = 0, tri = 0, sub = 0;
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
// components, |d|max or |o|max beyond 2^+-40, NaN t) are flagged irregular and test every triangle of a
// leaf instead.  DESIGN.md "In-leaf accelerator" has the full argument.
struct RayPre {
    F3 inv, oin, oif;
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
    P.regular = finite && (dmax >= small) && !(t != t) && (S.scene_eps <= 16777216.0f);
    const float fl = dmax * small;
    const float dx = fabsf(d.x) >= fl ? d.x : copysignf(fl, d.x);
    const float dy = fabsf(d.y) >= fl ? d.y : copysignf(fl, d.y);
    const float dz = fabsf(d.z) >= fl ? d.z : copysignf(fl, d.z);
    P.inv = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    P.sx = dx < 0;
    P.sy = dy < 0;
    P.sz = dz < 0;
    const float eps = S.scene_eps + 1.52587890625e-05f * omax;  // 2^-16
    const F3 oi = f3(o.x * P.inv.x, o.y * P.inv.y, o.z * P.inv.z);
    const F3 sl = f3(eps * fabsf(P.inv.x), eps * fabsf(P.inv.y), eps * fabsf(P.inv.z));
    P.oin = f3(oi.x + sl.x, oi.y + sl.y, oi.z + sl.z);
    P.oif = f3(oi.x - sl.x, oi.y - sl.y, oi.z - sl.z);
    (void)omax;
    return P;
}

// Conservative [tn, tf] of the widened box; explicit fma: this is NOT reference arithmetic.
__device__ __forceinline__ void slab_cons(const RayPre& P, const F3 lo, const F3 hi, float& tn, float& tf) {
    const float nx = P.sx ? hi.x : lo.x, fx = P.sx ? lo.x : hi.x;
    const float ny = P.sy ? hi.y : lo.y, fy = P.sy ? lo.y : hi.y;
    const float nz = P.sz ? hi.z : lo.z, fz = P.sz ? lo.z : hi.z;
    tn = fmaxf(fmaxf(__builtin_fmaf(nx, P.inv.x, -P.oin.x), __builtin_fmaf(ny, P.inv.y, -P.oin.y)), __builtin_fmaf(nz, P.inv.z, -P.oin.z));
    tf = fminf(fminf(__builtin_fmaf(fx, P.inv.x, -P.oif.x), __builtin_fmaf(fy, P.inv.y, -P.oif.y)), __builtin_fmaf(fz, P.inv.z, -P.oif.z));
}

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
    const TriEval E0 = eval_record(a0, b0, c0, e0, o, d);
    const TriEval E1 = eval_record(a1, b1, c1, e1, o, d);
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
        const uint2 m0 = *reinterpret_cast<const uint2*>(q + 3);
        const float4 a1 = q[4], b1 = q[5], c1 = q[6];
        const uint2 m1 = *reinterpret_cast<const uint2*>(q + 7);
        float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
        slab_cons(P, f3(a0.x, a0.y, a0.z), f3(a0.w, b0.x, b0.y), tn0, tf0);
        slab_cons(P, f3(b0.z, b0.w, c0.x), f3(c0.y, c0.z, c0.w), tn1, tf1);
        slab_cons(P, f3(a1.x, a1.y, a1.z), f3(a1.w, b1.x, b1.y), tn2, tf2);
        slab_cons(P, f3(b1.z, b1.w, c1.x), f3(c1.y, c1.z, c1.w), tn3, tf3);
        // Bound for culling: the running minimum, but never below 0 -- an origin-on-plane acceptance
        // ignores ray.t altogether (ray_tracing.cpp:43-47) and its box contains the origin (tn < 0).
        const float tc = fmaxf(best_t, 0.0f);
        const float inf = __builtin_inff();
        // sort key: entry parameter of a hit child, +inf for a missed (or absent: empty box) one
        float k0 = ((tn0 <= tf0) && (tf0 >= 0.0f) && (tn0 <= tc)) ? tn0 : inf;
        float k1 = ((tn1 <= tf1) && (tf1 >= 0.0f) && (tn1 <= tc)) ? tn1 : inf;
        float k2 = ((tn2 <= tf2) && (tf2 >= 0.0f) && (tn2 <= tc)) ? tn2 : inf;
        float k3 = ((tn3 <= tf3) && (tf3 >= 0.0f) && (tn3 <= tc)) ? tn3 : inf;
        uint32_t r0 = m0.x, r1 = m0.y, r2 = m1.x, r3 = m1.y;
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
#if CGRT_SORT_FULL
        CGRT_CSWAP(k1, r1, k3, r3)
        CGRT_CSWAP(k1, r1, k2, r2)
#endif
#undef CGRT_CSWAP
        // Branch-free pushes: the slot is always written and only kept (sp advanced) when the child was hit; a
        // level defers at most three children, so the writes stay inside the lane's SUB_STACK_ENTRIES slots.
        stk[sp * CGRT_BLOCK] = r3;
        sp += (k3 < inf) ? 1 : 0;
        stk[sp * CGRT_BLOCK] = r2;
        sp += (k2 < inf) ? 1 : 0;
        stk[sp * CGRT_BLOCK] = r1;
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
    } else if (SUB_WIDTH == 4) {
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
            cur = stk[sp * CGRT_BLOCK];
        }
    } else {
        int sp = sp0;
        uint32_t cur = LR.sub_root;
        for (;;) {
            // ---- node phase ----
            while (cur != REF_NONE && !(cur & REF_LEAF)) {
                if (COUNT) {
                    cnt.sub++;
                    if (first_active_lane()) cnt.w_sub++;
                }
#if CGRT_STAMP_SUB
                const unsigned long long st_a = stamp_now();
#endif
                const float4* q = reinterpret_cast<const float4*>(S.subnodes + cur);
                const float4 a = q[0], b = q[1], c = q[2];
                const uint2 m = *reinterpret_cast<const uint2*>(q + 3);
#if CGRT_STAMP_SUB
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long st_b = stamp_now();
                if (COUNT && first_active_lane()) cnt.c_wait += st_b - st_a;
#endif
                float tn0, tf0, tn1, tf1;
                slab_cons(P, f3(a.x, a.y, a.z), f3(a.w, b.x, b.y), tn0, tf0);
                slab_cons(P, f3(b.z, b.w, c.x), f3(c.y, c.z, c.w), tn1, tf1);
                // Bound for culling: the running minimum, but never below 0 -- an origin-on-plane acceptance
                // ignores ray.t altogether (ray_tracing.cpp:43-47) and its box contains the origin (tn < 0).
                const float tc = fmaxf(L.best_t, 0.0f);
                const bool h0 = (tn0 <= tf0) && (tf0 >= 0.0f) && (tn0 <= tc);
                const bool h1 = (tn1 <= tf1) && (tf1 >= 0.0f) && (tn1 <= tc);
                const bool swap = h1 && (!h0 || tn1 < tn0);  // child 1 goes first
                const uint32_t rn = swap ? m.y : m.x, rf = swap ? m.x : m.y;
                if (h0 && h1) {
                    stk[sp * CGRT_BLOCK] = rf;
                    sp += 1;
                }
                cur = (h0 || h1) ? rn : REF_NONE;
#if CGRT_STAMP_SUB
                {
                    const unsigned long long st_c = stamp_now();
                    if (COUNT && first_active_lane()) cnt.c_comp += st_c - st_b;
                }
#endif
            }
            // ---- triangle phase ----
#if CGRT_STAMP_SUB
            const unsigned long long st_t = stamp_now();
#endif
            if (cur != REF_NONE) test_run<COUNT>(S, run_first(cur), run_count(cur), o, d, L, cnt);
#if CGRT_STAMP_SUB
            {
                const unsigned long long st_u = stamp_now();
                if (COUNT && first_active_lane()) cnt.c_tri += st_u - st_t;
            }
#endif
            // ---- pop ----
            cur = REF_NONE;
            if (sp > sp0) {
                sp -= 1;
                cur = stk[sp * CGRT_BLOCK];
            }
            if (cur == REF_NONE) break;
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
        stk[sp * CGRT_BLOCK] = second;
        stk[(sp + 1) * CGRT_BLOCK] = __float_as_uint(tsec);
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
        const uint32_t r = stk[sp * CGRT_BLOCK];  // both words in one LDS access (ds_read2st64_b32)
        const float ts = __uint_as_float(stk[(sp + 1) * CGRT_BLOCK]);
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
    const uint32_t li = (ref & REF_LEAF_ACCEL) ? S.subnodes[ref & REF_INDEX26].pad[0] : (ref & ~REF_LEAF);
    return S.leaves[li];
}

// Topology phase of a round: intersectNonLeaf steps until the lane stands on a leaf (returns false, W.cur = the leaf)
// or has nothing left (returns true).
template <bool COUNT>
__device__ __forceinline__ bool walk_topology(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    const F3 o = W.o, d = W.d;
    uint32_t cur = W.cur;
    int sp = W.sp;
    // ---- topology phase: intersectNonLeaf steps until a leaf is reached ----
    for (;;) {
        if (cur == REF_NONE) {
            bool found = false;
            while (sp > 0) {
                sp -= 2;
                const float ts = __uint_as_float(stk[(sp + 1) * CGRT_BLOCK]);
                const uint32_t r = stk[sp * CGRT_BLOCK];
                if (!(W.t < ts)) {  // bvh.cpp:582: skip the deferred child iff ray.t < tSecond
                    cur = r;
                    found = true;
                    break;
                }
            }
            if (!found) {
                W.cur = REF_NONE;
                W.sp = 0;
                return true;
            }
        }
        if (cur & REF_LEAF) break;
        topo_step<COUNT>(S, W, o, d, cur, sp, stk, cnt);
    }
    // the lane stands on a leaf
    W.cur = cur;
    W.sp = sp;
    return false;
}

template <bool COUNT>
__device__ __forceinline__ bool walk_round(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    if (walk_topology<COUNT>(S, W, stk, cnt)) return true;
    // ---- leaf phase ----
    if (COUNT) cnt.leaf++;
    scan_leaf<COUNT>(S, leaf_rec_of(S, W.cur), W.o, W.d, W.P, W.t, W.hit_rec, stk, W.sp, cnt);
    W.cur = REF_NONE;
    return false;
}

#ifndef CGRT_PAIR
#define CGRT_PAIR 1  // runs of one or two records through test_pair (+1 %)
#endif
#ifndef CGRT_UNIFIED
#define CGRT_UNIFIED 1  // 0: the while-while rounds of walk_round (measured 7 % slower on the bench frame)
#endif
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
            if (!(cur & REF_LEAF_ACCEL) || !W.P.regular || SUB_WIDTH != 4) {
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
#if CGRT_PAIR
            if (SUB_LEAF_TRIS <= 2 && run_count(scur) <= 2)
                test_pair<COUNT>(S, run_first(scur), run_count(scur), o, d, L, cnt);
            else
#endif
                test_run<COUNT>(S, run_first(scur), run_count(scur), o, d, L, cnt);
            scur = REF_NONE;
        }
    };
    auto P = [&]() __attribute__((always_inline)) {  // next deferred child of the leaf, or the leaf is finished
        if (sp0 >= 0 && scur == REF_NONE) {
            if (sp > sp0) {
                sp -= 1;
                scur = stk[sp * CGRT_BLOCK];
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
    while (!done) {
        // (sequences with more or fewer pieces per iteration measured slower: profiles/r1_exp_one_loop.txt)
#if CGRT_STAMP_SUB
        unsigned long long st[8];
        st[0] = stamp_now();
        T();
        st[1] = stamp_now();
        T();
        st[2] = stamp_now();
        E();
        st[3] = stamp_now();
        N();
        st[4] = stamp_now();
        N();
        st[5] = stamp_now();
        R();
        st[6] = stamp_now();
        P();
        st[7] = stamp_now();
        if (COUNT && first_active_lane()) {
            for (int k = 0; k < 7; k++) cnt.c_piece[k] += st[k + 1] - st[k];
            cnt.c_piece[7] += 1;
            cnt.c_wait += st[3] - st[0];  // topology pieces
            cnt.c_comp += st[5] - st[3];  // node steps
            cnt.c_tri += st[7] - st[5];   // run test + pop
        }
#else
        T(); T(); E(); N(); N(); R(); P();
#endif
    }
}

// ANYHIT: stop after the first leaf that accepted a triangle.  The walk up to there is the reference's, so the hit FLAG
// is the reference's (some acceptance happens upstream iff one happens in the first leaf that has one); t and the
// record are those of that leaf, not the final ones.
template <bool COUNT, bool ANYHIT = false>
__device__ __forceinline__ void walk_tree(const SceneDev& S, const F3 o, const F3 d, float& t, uint32_t& hit_rec,
                                          uint32_t* __restrict__ stk, LaneCounters& cnt) {
    Walk W;
    W.o = o;
    W.d = d;
    W.t = t;
#if CGRT_UNIFIED
    if (walk_begin(S, W)) walk_tree_unified<COUNT, ANYHIT>(S, W, stk, cnt);
#else
    if (walk_begin(S, W))
        while (!walk_round<COUNT>(S, W, stk, cnt)) {
            if (ANYHIT && W.hit_rec != REF_NONE) break;
        }
#endif
    t = W.t;
    hit_rec = W.hit_rec;
}

#if CGRT_QUAD
// ---------------------------------------------------------------------------------------------
// Lane-cooperative leaf scan ("quad mode").  The frame ends when its hardest tiles do, and their waves spend most of
// their life with a handful of live rays on an otherwise idle SIMD, paying ~170 dependent VALU per 4-wide node step.
// When at most 16 lanes of a wave stand on an accelerated leaf, each such ray is spread over the 4 lanes of a quad for
// the duration of that leaf scan: lane k of the quad tests child box k of a node (one slab test instead of four, no
// sorting network: the children are ranked with three quad-local DPP compares) and triangle k of a run, and the quad
// shares the owner lane's LDS stack slice.  The scan rule (LeafScan) is order-free, so the outcome is the one of the
// scalar scan; the triangle arithmetic is the same test_record arithmetic, one triangle per lane.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(const uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(const float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
#define CGRT_QP_XOR1 0xB1  // quad_perm [1,0,3,2]
#define CGRT_QP_XOR2 0x4E  // quad_perm [2,3,0,1]
#define CGRT_QP_XOR3 0x1B  // quad_perm [3,2,1,0]

template <bool COUNT>
__device__ __forceinline__ void scan_leaves_quad(const SceneDev& S, const unsigned long long owners, const bool owner, const LeafRec LR,
                                                 Walk& W, uint32_t* __restrict__ s_stk_block, uint32_t* __restrict__ s_map,
                                                 LaneCounters& cnt) {
    const int lane = threadIdx.x & 63;
    const int nown = __popcll(owners);
    const int myrank = __popcll(owners & ((1ull << lane) - 1ull));
    if (owner) s_map[myrank] = (uint32_t)lane;
    __builtin_amdgcn_wave_barrier();
    const int Q = lane >> 2, k = lane & 3;
    const bool valid = Q < nown;
    const int src = valid ? (int)s_map[Q] : lane;
    // the owner's per-ray constants, broadcast to its quad (all 64 lanes take part in the shuffles)
    RayPre P;
    P.inv = f3(__shfl(W.P.inv.x, src, 64), __shfl(W.P.inv.y, src, 64), __shfl(W.P.inv.z, src, 64));
    P.oin = f3(__shfl(W.P.oin.x, src, 64), __shfl(W.P.oin.y, src, 64), __shfl(W.P.oin.z, src, 64));
    P.oif = f3(__shfl(W.P.oif.x, src, 64), __shfl(W.P.oif.y, src, 64), __shfl(W.P.oif.z, src, 64));
    const int sg = __shfl((int)((W.P.sx ? 1 : 0) | (W.P.sy ? 2 : 0) | (W.P.sz ? 4 : 0)), src, 64);
    P.sx = (sg & 1) != 0;
    P.sy = (sg & 2) != 0;
    P.sz = (sg & 4) != 0;
    P.regular = true;
    const F3 o = f3(__shfl(W.o.x, src, 64), __shfl(W.o.y, src, 64), __shfl(W.o.z, src, 64));
    const F3 d = f3(__shfl(W.d.x, src, 64), __shfl(W.d.y, src, 64), __shfl(W.d.z, src, 64));
    const float t_in = __shfl(W.t, src, 64);
    const uint32_t root = __shfl(LR.sub_root, src, 64);
    const int sp0 = __shfl(W.sp, src, 64);
    uint32_t* __restrict__ stk = s_stk_block + ((threadIdx.x & ~63u) + (uint32_t)src);  // the owner's LDS slice
    LeafScan L;  // quad-uniform
    L.best_t = t_in;
    L.best_k = -1;
    L.best_rec = REF_NONE;
    L.onp_k = -1;
    L.onp_rec = REF_NONE;
    if (valid) {
        const float inf = __builtin_inff();
        uint32_t cur = root;
        int sp = sp0;
        for (;;) {
            // ---- node phase: lane k tests child k of the 4-wide node ----
            while (cur != REF_NONE && !(cur & REF_LEAF)) {
                if (COUNT) {
                    if (k == 0) cnt.sub++;
                    if (first_active_lane()) cnt.w_sub++;
                }
                const char* rec = reinterpret_cast<const char*>(S.subnodes + cur + (uint32_t)(k >> 1));
                const float2* bp = reinterpret_cast<const float2*>(rec + (k & 1) * 24);
                const float2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
                const uint32_t ref = reinterpret_cast<const uint32_t*>(rec)[12 + (k & 1)];
                float tn, tf;
                slab_cons(P, f3(b0.x, b0.y, b1.x), f3(b1.y, b2.x, b2.y), tn, tf);
                const float tc = fmaxf(L.best_t, 0.0f);  // see scan_leaf
                const bool h = (tn <= tf) && (tf >= 0.0f) && (tn <= tc);
                const float key = h ? tn : inf;
                // rank of this child among the quad's four (ties by lane)
                const float k1 = dpp_f32<CGRT_QP_XOR1>(key), k2 = dpp_f32<CGRT_QP_XOR2>(key), k3 = dpp_f32<CGRT_QP_XOR3>(key);
                const int rank = ((k1 < key) || (k1 == key && (k ^ 1) < k) ? 1 : 0) + ((k2 < key) || (k2 == key && (k ^ 2) < k) ? 1 : 0) +
                                 ((k3 < key) || (k3 == key && (k ^ 3) < k) ? 1 : 0);
                const uint32_t hm = (uint32_t)((__ballot(h) >> (lane & ~3)) & 0xFull);
                const int nhit = __popc(hm);
                // the nearest hit child is next; the others are deferred far-to-near so that they pop near-to-far
                if (h && rank >= 1) stk[(sp + nhit - 1 - rank) * CGRT_BLOCK] = ref;
                sp += nhit > 0 ? nhit - 1 : 0;
                uint32_t r0 = (h && rank == 0) ? ref : 0u;
                r0 |= dpp_u32<CGRT_QP_XOR1>(r0);
                r0 |= dpp_u32<CGRT_QP_XOR2>(r0);
                cur = nhit > 0 ? r0 : REF_NONE;
            }
            // ---- triangle phase: lane k tests triangle k of the run ----
            if (cur != REF_NONE) {
                const uint32_t first = run_first(cur), n = run_count(cur);
                for (uint32_t base = 0; base < n; base += 4) {
                    bool has = false;
                    float c_tt = inf;
                    int c_k = 0x7fffffff, c_onk = -1;
                    uint32_t c_rec = REF_NONE, c_onrec = REF_NONE;
                    if (base + (uint32_t)k < n) {
                        if (COUNT) cnt.tri++;
                        const uint32_t r = first + base + (uint32_t)k;
                        LeafScan T = L;  // evaluate this triangle against the quad-uniform state
                        const float4* q = reinterpret_cast<const float4*>(S.tris + r);
                        test_record(q[0], q[1], q[2], q[3], r, o, d, T);
                        if (T.best_rec == r && (T.best_k != L.best_k || T.best_t != L.best_t || L.best_rec != r)) {
                            has = true;
                            c_tt = T.best_t;
                            c_k = T.best_k;
                            c_rec = r;
                        }
                        if (T.onp_k > L.onp_k) {
                            c_onk = T.onp_k;
                            c_onrec = r;
                        }
                    }
                    if (COUNT && first_active_lane()) cnt.w_tri++;
                    // quad reduction: lexicographic minimum of (t, scan position); last on-plane scan position
#define CGRT_QRED(CTRL)                                                                          \
    {                                                                                            \
        const bool o_has = dpp_u32<CTRL>(has ? 1u : 0u) != 0u;                                   \
        const float o_tt = dpp_f32<CTRL>(c_tt);                                                  \
        const int o_k = (int)dpp_u32<CTRL>((uint32_t)c_k);                                       \
        const uint32_t o_rec = dpp_u32<CTRL>(c_rec);                                             \
        const bool take = o_has && (!has || o_tt < c_tt || (o_tt == c_tt && o_k < c_k));         \
        has = has || o_has;                                                                      \
        c_tt = take ? o_tt : c_tt;                                                               \
        c_k = take ? o_k : c_k;                                                                  \
        c_rec = take ? o_rec : c_rec;                                                            \
        const int o_onk = (int)dpp_u32<CTRL>((uint32_t)c_onk);                                   \
        const uint32_t o_onrec = dpp_u32<CTRL>(c_onrec);                                         \
        const bool take2 = o_onk > c_onk;                                                        \
        c_onk = take2 ? o_onk : c_onk;                                                           \
        c_onrec = take2 ? o_onrec : c_onrec;                                                     \
    }
                    CGRT_QRED(CGRT_QP_XOR1)
                    CGRT_QRED(CGRT_QP_XOR2)
#undef CGRT_QRED
                    if (has) {
                        L.best_t = c_tt;
                        L.best_k = c_k;
                        L.best_rec = c_rec;
                    }
                    if (c_onk > L.onp_k) {
                        L.onp_k = c_onk;
                        L.onp_rec = c_onrec;
                    }
                }
            }
            // ---- pop (all four lanes read the same slot) ----
            if (sp <= sp0) break;
            sp -= 1;
            cur = stk[sp * CGRT_BLOCK];
        }
    }
    // results back to the owner lanes (quad lane 0 of quad `myrank`)
    const int ql = 4 * myrank;
    const float r_t = __shfl(L.best_t, ql, 64);
    const int r_k = __shfl(L.best_k, ql, 64);
    const uint32_t r_rec = __shfl(L.best_rec, ql, 64);
    const int r_onk = __shfl(L.onp_k, ql, 64);
    const uint32_t r_onrec = __shfl(L.onp_rec, ql, 64);
    if (owner) {
        if (r_onk >= 0) {
            W.t = 0.0f;
            W.hit_rec = r_onrec;
        } else if (r_k >= 0) {
            W.t = r_t;
            W.hit_rec = r_rec;
        }
    }
}

#endif  // CGRT_QUAD

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
    unsigned long long v[5] = {active ? 1ull : 0ull, c.inner, c.leaf, c.tri, c.sub};
    for (int k = 0; k < 5; k++) {
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
    const uint32_t s = (j >> 4) * 8u + lane8;  // rank-local super-tile
    if (s >= F.nst_rank) return false;
    const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * s;
    const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
    const int idx = (int)(j & 15u) * 4 + (int)(tid >> 6);
    const int tx = stx * ST_TILES + (idx & 7), ty = sty * ST_TILES + (idx >> 3);
    x = F.x0 + tx * 8 + (lane & 7);
    y = F.y0 + ty * 8 + (lane >> 3);
    return x < F.x1 && y < F.y1;
}
__device__ __forceinline__ bool tile_pixel(const FrameDev& F, int lane, int& x, int& y) {
    (void)lane;
    return tile_pixel_of(F, blockIdx.x, threadIdx.x, x, y);
}

// STAMP (diagnostic build only, never timed): lane 0 of every wave writes {start, end} of s_memtime and
// s_memrealtime plus its work counters
// into `counters`, which then is a 16 x nwaves u64 buffer that nothing else reads.
template <bool COUNT, bool STAMP = false>
__global__ CGRT_LB void k_trace_primary(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits,
                                                              float* __restrict__ normals, unsigned long long* counters) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = blockIdx.x * (CGRT_BLOCK / 64) + (threadIdx.x >> 6);
    unsigned long long st0 = 0, rt0 = 0;
    if (STAMP) {
        st0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    int x = 0, y = 0;
    const bool active = tile_pixel(F, lane, x, y);
    LaneCounters cnt;
#if CGRT_QUAD
    {
        __shared__ uint32_t s_map[(CGRT_BLOCK / 64) * 16];
        uint32_t* map = s_map + (threadIdx.x >> 6) * 16;
        uint32_t* stk = s_stk + threadIdx.x;
        const size_t pix = (size_t)y * F.W + x;
        Walk W;
        W.t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
        bool alive = false;
        if (active) {
            primary_ray(C, F.W, F.H, x, y, W.o, W.d);
            alive = walk_begin(S, W);
            if (!alive) finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
        }
        // wave-uniform round loop: every lane of the wave reaches the leaf phase together, so that idle lanes can be
        // lent to the rays that are still alive (scan_leaves_quad)
        while (__any(alive)) {
            bool at_leaf = false;
            if (alive) {
                if (walk_topology<COUNT || STAMP>(S, W, stk, cnt)) {
                    finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
                    alive = false;
                } else {
                    at_leaf = true;
                }
            }
            LeafRec LR;
            LR.first = 0;
            LR.count = 0;
            LR.sub_root = REF_NONE;
            LR.pad = 0;
            if (at_leaf) {
                LR = leaf_rec_of(S, W.cur);
                if (COUNT || STAMP) cnt.leaf++;
            }
            const bool qok = at_leaf && (LR.sub_root != REF_NONE) && W.P.regular && (SUB_WIDTH == 4);
            const unsigned long long qm = __ballot(qok);
            const int nq = __popcll(qm);
            const bool quad = nq > 0 && nq <= CGRT_QUAD_MAX_RAYS;
            if (quad) scan_leaves_quad<COUNT || STAMP>(S, qm, qok, LR, W, s_stk, map, cnt);
            if (at_leaf && !(quad && qok)) scan_leaf<COUNT || STAMP>(S, LR, W.o, W.d, W.P, W.t, W.hit_rec, stk, W.sp, cnt);
            if (at_leaf) W.cur = REF_NONE;
        }
    }
#else
    if (active) {
        F3 o, d;
        primary_ray(C, F.W, F.H, x, y, o, d);
        float t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
        uint32_t hit_rec = REF_NONE;
        walk_tree<COUNT || STAMP>(S, o, d, t, hit_rec, s_stk + threadIdx.x, cnt);
        const size_t pix = (size_t)y * F.W + x;
        finish_ray(S, o, d, t, hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
    }
#endif
    if (STAMP) {
        const unsigned long long st1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long nactive = __popcll(__ballot(active));
        uint32_t v[11] = {cnt.inner, cnt.leaf, cnt.tri, cnt.sub, cnt.w_inner, cnt.w_sub, cnt.w_tri, cnt.inner, cnt.leaf, cnt.tri, cnt.sub};
        for (int k = 0; k < 11; k++)
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t other = __shfl_down(v[k], off, 64);
                v[k] = k < 7 ? v[k] + other : (v[k] > other ? v[k] : other);  // sums, then per-lane maxima
            }
#if CGRT_STAMP_SUB
        unsigned long long cw[3] = {cnt.c_wait, cnt.c_comp, cnt.c_tri};
        unsigned long long cp[8];
        for (int k = 0; k < 8; k++) {
            cp[k] = cnt.c_piece[k];
            for (int off = 32; off > 0; off >>= 1) cp[k] += __shfl_down(cp[k], off, 64);
        }
        for (int k = 0; k < 3; k++)
            for (int off = 32; off > 0; off >>= 1) cw[k] += __shfl_down(cw[k], off, 64);
#endif
        if (lane == 0) {
            unsigned long long* p = counters + 16ull * wave_global;
            p[0] = st0;
            p[1] = st1;
            p[2] = rt0;
            p[3] = rt1;
            for (int k = 0; k < 11; k++) p[4 + k] = v[k];
            p[15] = nactive;
#if CGRT_STAMP_SUB
            p[12] = cw[0];  // replaces max_leaf: wave-summed cycles waiting for the node record
            p[13] = cw[1];  // replaces max_tri: ... computing the node step
            p[14] = cw[2];  // replaces max_sub: ... in the triangle phase
            // one-loop walk: cycles per piece T T E N, then N R P and the loop trips in the low/high halves of p[4..7], p[11]
            for (int k = 0; k < 4; k++) p[4 + k] = cp[k];
            p[11] = cp[4];
            p[2] = cp[5];  // (replaces the s_memrealtime pair)
            p[3] = cp[6];
            p[15] = nactive | (cp[7] << 8);
#endif
        }
    } else if (COUNT) {
        flush_counters(cnt, active, counters);
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent primary-frame kernel: a fixed grid of waves (4 workgroups per CU) pulls 8x8-pixel tiles
// from eight queues (one per blockIdx % 8 residue, i.e. per XCD under the observed placement; a wave
// drains its home queue first, then steals from the others) and keeps its 64 lanes busy: between two
// rounds of walk_round(), when at least CGRT_REFILL_MIN_IDLE lanes have finished their ray, the idle
// lanes are handed the next pixels of the wave's current tile (ballot of the idle lanes, rank by
// popcount of the lower lanes).  Rays that fail the root gate are written out at once and their lane is
// refilled in the same pass, so after a refill the wave's active lanes all carry rays that actually
// traverse.  Tiles are handed out in the same super-tile order as the plain kernel.
// Every wave leaves when all queues are exhausted and its lanes are idle; no wave waits on another.
// The queue block (8 heads + 1 exit counter, one 128-byte line each) is reset by the last wave to leave.
#ifndef CGRT_REFILL_MIN_IDLE
#define CGRT_REFILL_MIN_IDLE 16
#endif
#ifndef CGRT_QUEUE_CHUNK
#define CGRT_QUEUE_CHUNK 1  // tiles taken per atomic; larger chunks were measured much slower (4: 2.4x, 16: 7x): hard tiles cluster
#endif
// Queue block layout: head q at word 32*q (each head on its own 128-byte line: atomics on one line serialise
// at the memory side), exit counter at word 32*8.
#define CGRT_QUEUE_WORDS (32 * 9)

__device__ __forceinline__ uint32_t queue_units(const FrameDev& F, uint32_t q) {  // tiles in queue q
    return q < F.nst_rank ? ((F.nst_rank - q + 7u) / 8u) * 64u : 0u;
}

template <bool COUNT>
__global__ __launch_bounds__(CGRT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_trace_primary_persistent(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits,
                                                   float* __restrict__ normals, unsigned long long* counters,
                                                   unsigned int* __restrict__ queue, unsigned int total_waves) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    uint32_t* stk = s_stk + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t home = blockIdx.x & 7u;
    LaneCounters cnt;
    Walk W;
    bool active = false;
    size_t pix = 0;
    // wave-uniform refill state
    uint32_t tile_q = 0, tile_k = 0, next_pix = 64, tried = 0, chunk_left = 0;
    bool exhausted = false;
    uint32_t nrays = 0;
    for (;;) {
        unsigned long long act = __ballot(active);
        if (!exhausted && (64 - __popcll(act)) >= CGRT_REFILL_MIN_IDLE) {
            unsigned long long idle = ~act;
            while (idle != 0ull) {
                if (next_pix >= 64u) {  // next tile: rest of the chunk in hand, else home queue first, then the others in order
                    bool got = false;
                    if (chunk_left > 0u) {
                        tile_k += 1u;
                        chunk_left -= 1u;
                        next_pix = 0;
                        got = true;
                    }
                    while (!got && tried < 8u) {
                        const uint32_t q = (home + tried) & 7u;
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(queue + 32u * q, (unsigned)CGRT_QUEUE_CHUNK);
                        k = __builtin_amdgcn_readfirstlane(k);
                        const uint32_t nq = queue_units(F, q);
                        if (k < nq) {
                            tile_q = q;
                            tile_k = k;
                            chunk_left = min((uint32_t)CGRT_QUEUE_CHUNK, nq - k) - 1u;
                            next_pix = 0;
                            got = true;
                            break;
                        }
                        tried++;
                    }
                    if (!got) {
                        exhausted = true;
                        break;
                    }
                }
                const uint32_t nidle = (uint32_t)__popcll(idle);
                const uint32_t take = min(64u - next_pix, nidle);
                const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                const bool mine = ((idle >> lane) & 1ull) && rank < take;
                if (mine) {
                    const uint32_t p = next_pix + rank;
                    const uint32_t s_loc = tile_q + 8u * (tile_k >> 6);  // rank-local super-tile
                    const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * s_loc;
                    const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
                    const int idx = (int)(tile_k & 63u);
                    const int x = F.x0 + (stx * ST_TILES + (idx & 7)) * 8 + (int)(p & 7u);
                    const int y = F.y0 + (sty * ST_TILES + (idx >> 3)) * 8 + (int)(p >> 3);
                    if (x < F.x1 && y < F.y1) {
                        if (COUNT) nrays++;
                        pix = (size_t)y * F.W + x;
                        primary_ray(C, F.W, F.H, x, y, W.o, W.d);
                        W.t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
                        if (walk_begin(S, W))
                            active = true;
                        else
                            finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
                    }
                }
                next_pix += take;
                idle = ~__ballot(active);  // lanes whose ray died at the root gate are idle again
                if (__popcll(idle) < CGRT_REFILL_MIN_IDLE && next_pix < 64u) break;
            }
            act = __ballot(active);
        }
        if (act == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (active) {
            if (walk_round<COUNT>(S, W, stk, cnt)) {
                finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
                active = false;
            }
        }
    }
    if (COUNT) {
        unsigned long long v[5] = {nrays, cnt.inner, cnt.leaf, cnt.tri, cnt.sub};
        for (int k = 0; k < 5; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(counters + k, x);
        }
    }
    if (lane == 0) {
        if (atomicAdd(queue + 32 * 8, 1u) == total_waves - 1u) {  // last wave out: leave the queue block clean
            for (int q = 0; q < 9; q++) queue[32 * q] = 0u;
        }
    }
}

// Primary frame for the shading wavefront (cgrt_render): the fused kernel's walk, but only the rays that HIT are written,
// appended to a compact list {ray, hit, normal, pixel} (one atomic per workgroup, lanes ranked by ballot; workgroups
// finish roughly in launch order, so the list keeps the frame's tile order).  Pixels that miss need no further work upstream
// either (main.cpp:293: black).  count = one zeroed device word.
__global__ CGRT_LB void k_trace_primary_compact(SceneDev S, CameraDev C, FrameDev F, float* __restrict__ rays, CgrtHitDev* __restrict__ hits,
                                                float* __restrict__ normals, int* __restrict__ pixels, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    const int lane = threadIdx.x & 63;
    int x = 0, y = 0;
    const bool active = tile_pixel(F, lane, x, y);
    LaneCounters cnt;
    CgrtHitDev h;
    h.hit = 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0), nn = f3(0, 0, 0);
    if (active) {
        primary_ray(C, F.W, F.H, x, y, o, d);
        float t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
        uint32_t hit_rec = REF_NONE;
        walk_tree<false>(S, o, d, t, hit_rec, s_stk + threadIdx.x, cnt);
        resolve_hit(S, o, d, t, hit_rec, true, h, nn);
    }
    // one atomic per workgroup (same-address atomics serialise at the L2); the workgroup's LDS is only released when its
    // last wave ends anyway, so waiting for it here costs no occupancy
    const bool keep = active && h.hit != 0;
    const unsigned long long m = __ballot(keep);
    __shared__ uint32_t s_cnt[CGRT_BLOCK / 64 + 1];
    const unsigned w = threadIdx.x >> 6;
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (unsigned k = 0; k < CGRT_BLOCK / 64; k++) {
            const uint32_t c = s_cnt[k];
            s_cnt[k] = tot;
            tot += c;
        }
        s_cnt[CGRT_BLOCK / 64] = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    if (keep) {
        const unsigned long long idx = s_cnt[CGRT_BLOCK / 64] + s_cnt[w] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        float* r = rays + 7 * idx;
        r[0] = o.x;
        r[1] = o.y;
        r[2] = o.z;
        r[3] = d.x;
        r[4] = d.y;
        r[5] = d.z;
        r[6] = 3.402823466e+38f;
        hits[idx] = h;
        normals[3 * idx] = nn.x;
        normals[3 * idx + 1] = nn.y;
        normals[3 * idx + 2] = nn.z;
        pixels[idx] = y * F.W + x;
    }
}
// rgb of every pixel this rank owns := 0 (main.cpp:293; the hits are written over it afterwards)
__global__ __launch_bounds__(CGRT_BLOCK) void k_clear_owned(FrameDev F, float* __restrict__ rgb) {
    int x = 0, y = 0;
    if (!tile_pixel_of(F, blockIdx.x, threadIdx.x, x, y)) return;
    float* p = rgb + 3ull * ((unsigned long long)y * F.W + x);
    p[0] = p[1] = p[2] = 0.0f;
}

// dcount (optional): device word holding the number of rays actually present (<= n); the grid covers n.
template <bool COUNT>
__global__ CGRT_LB void k_trace_batch(SceneDev S, const float* __restrict__ rays, unsigned long long n,
                                                            CgrtHitDev* __restrict__ hits, float* __restrict__ normals,
                                                            unsigned long long* counters, const uint32_t* __restrict__ dcount) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    const unsigned long long i = (unsigned long long)blockIdx.x * CGRT_BLOCK + threadIdx.x;
    if (dcount) {
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    const bool active = i < n;
    LaneCounters cnt;
    if (active) {
        const float* r = rays + 7 * i;
        const F3 o = f3(r[0], r[1], r[2]), d = f3(r[3], r[4], r[5]);
        float t = r[6];
        uint32_t hit_rec = REF_NONE;
        walk_tree<COUNT>(S, o, d, t, hit_rec, s_stk + threadIdx.x, cnt);
        finish_ray(S, o, d, t, hit_rec, hits + i, normals ? normals + 3 * i : nullptr);
    }
    if (COUNT) flush_counters(cnt, active, counters);
}

// Soft shadows of spherical lights (main.cpp:168-218): `samples` shadow rays per (hit item, light), generated in
// registers from the item's ray + hit and the unit-vector table (no ray buffer: 200 samples x 2 M hits would be 12 GB),
// traversed, and counted: lit[item * nlights + l] = number of samples with !intersect || ray.t > lightT (:183-199).
// Thread g = (item * nlights + l) * samples + smp: a wave's rays leave one or two surface points towards one small
// sphere, the most coherent batch this library sees.  ANYHIT stops a ray at its first accepting leaf: the count needs
// the hit flag only (an accepted t is below lightT by construction, or 0 from the on-plane rule, never above).
template <bool ANYHIT>
__global__ CGRT_LB void k_soft_shadow(SceneDev S, SoftDev Q, const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits,
                                      const int* __restrict__ item_pixels, unsigned long long nthreads, uint32_t* __restrict__ lit) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    const unsigned long long g = (unsigned long long)blockIdx.x * CGRT_BLOCK + threadIdx.x;
    const bool in = g < nthreads;
    const unsigned long long key = in ? g / Q.samples : 0ull;  // item * nlights + l
    const uint32_t smp = in ? (uint32_t)(g - key * Q.samples) : 0u;
    const unsigned long long item = key / Q.nlights;
    const uint32_t l = (uint32_t)(key - item * Q.nlights);
    bool is_lit = false;
    const bool live = in && hits[item].hit != 0;
    if (live) {
        const float* r = rays + 7 * item;
        const F3 pointOn = add(f3(r[0], r[1], r[2]), scale(f3(r[3], r[4], r[5]), hits[item].t));
        const float* L = Q.lights + 7 * l;
        const float* u = Q.units + 3ull * soft_sample_index(Q.seed, (uint32_t)item_pixels[item], Q.level, l, smp, Q.nunits);
        F3 o, d;
        float t;
        soft_shadow_ray(pointOn, f3(L[0], L[1], L[2]), L[3], f3(u[0], u[1], u[2]), o, d, t);
        const float lightT = t;
        uint32_t hit_rec = REF_NONE;
        LaneCounters cnt;
        walk_tree<false, ANYHIT>(S, o, d, t, hit_rec, s_stk + threadIdx.x, cnt);
        bool hit = hit_rec != REF_NONE;
        if (!ANYHIT || !hit) {  // spheres come after the meshes in BoundingVolumeHierarchy::intersect (bvh.cpp:875-880)
            for (uint32_t k = 0; k < S.nspheres; k++) {
                F3 nrm;
                hit |= ray_sphere(f3(S.spheres[k].c[0], S.spheres[k].c[1], S.spheres[k].c[2]), S.spheres[k].radius, o, d, t, nrm);
            }
        }
        is_lit = !hit || t > lightT;
    }
    // one atomic per (wave, key): a wave's keys are consecutive
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const unsigned long long k0 = __shfl(key, first);
        const unsigned long long same = __ballot(live && key == k0);
        const uint32_t c = (uint32_t)__popcll(__ballot(live && key == k0 && is_lit));
        if ((int)(threadIdx.x & 63) == first && c) atomicAdd(lit + k0, c);
        todo &= ~same;
    }
}

__global__ __launch_bounds__(CGRT_BLOCK) void k_generate_rays(CameraDev C, int W, int H, int x0, int y0, int x1, int y1,
                                                              float* __restrict__ rays) {
    const unsigned long long i = (unsigned long long)blockIdx.x * CGRT_BLOCK + threadIdx.x;
    const int rw = x1 - x0;
    const unsigned long long n = (unsigned long long)rw * (unsigned long long)(y1 - y0);
    if (i >= n) return;
    const int x = x0 + (int)(i % (unsigned long long)rw), y = y0 + (int)(i / (unsigned long long)rw);
    F3 o, d;
    primary_ray(C, W, H, x, y, o, d);
    float* r = rays + 7 * i;
    r[0] = o.x;
    r[1] = o.y;
    r[2] = o.z;
    r[3] = d.x;
    r[4] = d.y;
    r[5] = d.z;
    r[6] = 3.402823466e+38f;
}

// ---- element-wise primitives (src/ray_tracing.h:10-20) ----
__global__ void k_ray_triangle(const float* __restrict__ tri, const float* __restrict__ rays, unsigned long long n,
                               float* __restrict__ t_out, uint8_t* __restrict__ hit, float* __restrict__ normals) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = tri + 18 * i;
    const float* r = rays + 7 * i;
    const F3 v0 = ld3(q), v1 = ld3(q + 3), v2 = ld3(q + 6);
    const F3 o = ld3(r), d = ld3(r + 3);
    float t = r[6];
    F3 pn;
    float D;
    triangle_plane(v0, v1, v2, pn, D);  // the reference rebuilds the plane per call (ray_tracing.cpp:88)
    const bool h = ray_triangle_geom(v0, v1, v2, pn, D, o, d, t);
    t_out[i] = t;
    hit[i] = h;
    if (h && normals) {
        const F3 nn = hit_normal(v0, v1, v2, pn, ld3(q + 9), ld3(q + 12), ld3(q + 15), o, d, t);
        normals[3 * i] = nn.x;
        normals[3 * i + 1] = nn.y;
        normals[3 * i + 2] = nn.z;
    }
}
__global__ void k_ray_plane(const float* __restrict__ plane, const float* __restrict__ rays, unsigned long long n,
                            float* __restrict__ t_out, uint8_t* __restrict__ hit) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 7 * i;
    float t = r[6];
    hit[i] = ray_plane(plane[4 * i], ld3(plane + 4 * i + 1), ld3(r), ld3(r + 3), t);
    t_out[i] = t;
}
__global__ void k_ray_box(const float* __restrict__ box, const float* __restrict__ rays, unsigned long long n,
                          float* __restrict__ t_out, uint8_t* __restrict__ hit, uint8_t* __restrict__ inside) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 7 * i;
    const F3 lo = ld3(box + 6 * i), hi = ld3(box + 6 * i + 3), o = ld3(r), d = ld3(r + 3);
    float t = r[6], tb;
    const bool h = ray_box(lo, hi, o, d, t, tb);
    t_out[i] = h ? tb : t;  // the reference writes ray.t = box parameter on success (ray_tracing.cpp:198)
    hit[i] = h;
    if (inside) inside[i] = starts_in_box(o, lo, hi);
}
__global__ void k_ray_sphere(const float* __restrict__ sph, const float* __restrict__ rays, unsigned long long n,
                             float* __restrict__ t_out, uint8_t* __restrict__ hit, float* __restrict__ normals) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 7 * i;
    float t = r[6];
    F3 nn = f3(0, 0, 0);
    const bool h = ray_sphere(ld3(sph + 4 * i), sph[4 * i + 3], ld3(r), ld3(r + 3), t, nn);
    t_out[i] = t;
    hit[i] = h;
    if (h && normals) {
        normals[3 * i] = nn.x;
        normals[3 * i + 1] = nn.y;
        normals[3 * i + 2] = nn.z;
    }
}
__global__ void k_triangle_plane(const float* __restrict__ tri, unsigned long long n, float* __restrict__ plane) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 pn;
    float D;
    triangle_plane(ld3(tri + 9 * i), ld3(tri + 9 * i + 3), ld3(tri + 9 * i + 6), pn, D);
    plane[4 * i] = D;
    plane[4 * i + 1] = pn.x;
    plane[4 * i + 2] = pn.y;
    plane[4 * i + 3] = pn.z;
}
__global__ void k_point_in_triangle(const float* __restrict__ in, unsigned long long n, uint8_t* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in + 15 * i;
    out[i] = point_in_triangle(ld3(q), ld3(q + 3), ld3(q + 6), ld3(q + 9), ld3(q + 12));
}

// calibration kernel for the FETCH_SIZE counter: every lane reads ONE 64-byte record (4 x dwordx4, the access shape of
// the traversal kernels) at a pseudo-random, never repeated position of a table far larger than the Infinity Cache.
__global__ void k_gather_calib(const float4* __restrict__ table, unsigned long long nrecords, unsigned long long mult,
                               unsigned long long add, float* __restrict__ sink) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrecords) return;
    const unsigned long long j = (i * mult + add) % nrecords;  // mult coprime with nrecords: a permutation
    const float4* q = table + 4 * j;
    const float4 a = q[0], b = q[1], c = q[2], e = q[3];
    const float v = a.x + b.y + c.z + e.w;
    if (v == 123456.0f) sink[0] = v;  // keeps the loads alive
}

// diagnostic / test kernel: fdiv4 against IEEE a / d, bit for bit (zeros compare equal regardless of sign)
__global__ void k_fastdiv_check(const float* __restrict__ a, const float* __restrict__ d, unsigned long long n,
                                unsigned long long* __restrict__ mismatches, float* __restrict__ first_bad) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dd = d[i], aa = a[i];
    const float yh = 1.0f / dd;
    const float yl = __builtin_fmaf(-dd, yh, 1.0f) * yh;
    const float q = fdiv4(aa, dd, yh, yl), ref = aa / dd;
    const bool same = (__float_as_uint(q) == __float_as_uint(ref)) || (q == 0.0f && ref == 0.0f);
    if (!same) {
        if (atomicAdd(mismatches, 1ull) == 0ull) {
            first_bad[0] = aa;
            first_bad[1] = dd;
            first_bad[2] = q;
            first_bad[3] = ref;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(unsigned long long n, unsigned block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_trace_primary(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                unsigned long long* counters, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    const unsigned blocks = F.nblocks;
    static const unsigned lds_pad = [] {
        const char* e = getenv("CGRT_EXP_LDS_PAD");  // experiment knob: extra dynamic LDS to cap blocks per CU
        return e ? (unsigned)atoi(e) : 0u;
    }();
    if (counters)
        hipLaunchKernelGGL(k_trace_primary<true>, dim3(blocks), dim3(CGRT_BLOCK), lds_pad, stream, S, C, F, hits, normals, counters);
    else
        hipLaunchKernelGGL(k_trace_primary<false>, dim3(blocks), dim3(CGRT_BLOCK), lds_pad, stream, S, C, F, hits, normals, counters);
    return hipGetLastError();
}
hipError_t launch_trace_primary_persistent(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                           unsigned long long* counters, unsigned int* queue, unsigned blocks, hipStream_t stream) {
    if (F.nst_rank == 0) return hipSuccess;
    const unsigned long long tiles = (unsigned long long)((F.nst_rank + 7u) / 8u) * 8u * 64u;
    unsigned b = (unsigned)std::min<unsigned long long>(blocks, (tiles + 3) / 4);
    b = std::max(8u, (b + 7u) & ~7u);
    const unsigned waves = b * (CGRT_BLOCK / 64);
    if (counters)
        hipLaunchKernelGGL(k_trace_primary_persistent<true>, dim3(b), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters, queue, waves);
    else
        hipLaunchKernelGGL(k_trace_primary_persistent<false>, dim3(b), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters, queue, waves);
    return hipGetLastError();
}
hipError_t launch_trace_primary_stamped(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits,
                                        unsigned long long* stamps, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    const unsigned blocks = F.nblocks;
    hipLaunchKernelGGL((k_trace_primary<false, true>), dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, nullptr, stamps);
    return hipGetLastError();
}
hipError_t launch_trace_batch(const SceneDev& S, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                              unsigned long long* counters, hipStream_t stream, const uint32_t* dcount) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = grid_for(n, CGRT_BLOCK);
    if (counters)
        hipLaunchKernelGGL(k_trace_batch<true>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, rays, n, hits, normals, counters, dcount);
    else
        hipLaunchKernelGGL(k_trace_batch<false>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, rays, n, hits, normals, counters, dcount);
    return hipGetLastError();
}
hipError_t launch_trace_primary_compact(const SceneDev& S, const CameraDev& C, const FrameDev& F, float* rays, CgrtHitDev* hits, float* normals,
                                        int* pixels, uint32_t* count, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_trace_primary_compact, dim3(F.nblocks), dim3(CGRT_BLOCK), 0, stream, S, C, F, rays, hits, normals, pixels, count);
    return hipGetLastError();
}
hipError_t launch_clear_owned(const FrameDev& F, float* rgb, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_clear_owned, dim3(F.nblocks), dim3(CGRT_BLOCK), 0, stream, F, rgb);
    return hipGetLastError();
}
hipError_t launch_soft_shadow(const SceneDev& S, const SoftDev& Q, const float* rays, const CgrtHitDev* hits, const int* item_pixels,
                              unsigned long long nitems, uint32_t* lit, int anyhit, hipStream_t stream) {
    const unsigned long long nthreads = nitems * Q.nlights * Q.samples;
    if (nthreads == 0) return hipSuccess;
    const unsigned long long blocks = (nthreads + CGRT_BLOCK - 1) / CGRT_BLOCK;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (anyhit)
        hipLaunchKernelGGL(k_soft_shadow<true>, dim3((unsigned)blocks), dim3(CGRT_BLOCK), 0, stream, S, Q, rays, hits, item_pixels, nthreads, lit);
    else
        hipLaunchKernelGGL(k_soft_shadow<false>, dim3((unsigned)blocks), dim3(CGRT_BLOCK), 0, stream, S, Q, rays, hits, item_pixels, nthreads, lit);
    return hipGetLastError();
}
hipError_t launch_generate_rays(const CameraDev& C, int W, int H, int x0, int y0, int x1, int y1, float* rays, hipStream_t stream) {
    const unsigned long long n = (unsigned long long)(x1 - x0) * (unsigned long long)(y1 - y0);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_rays, dim3(grid_for(n, CGRT_BLOCK)), dim3(CGRT_BLOCK), 0, stream, C, W, H, x0, y0, x1, y1, rays);
    return hipGetLastError();
}
hipError_t launch_gather_calib(const void* table, unsigned long long nrecords, unsigned long long mult, unsigned long long add, float* sink,
                               hipStream_t s) {
    if (nrecords)
        hipLaunchKernelGGL(k_gather_calib, dim3(grid_for(nrecords, 256)), dim3(256), 0, s, static_cast<const float4*>(table), nrecords, mult,
                           add, sink);
    return hipGetLastError();
}
hipError_t launch_fastdiv_check(const float* a, const float* d, unsigned long long n, unsigned long long* mismatches, float* first_bad,
                                hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fastdiv_check, dim3(grid_for(n, 256)), dim3(256), 0, s, a, d, n, mismatches, first_bad);
    return hipGetLastError();
}
hipError_t launch_ray_triangle(const float* tri, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, float* normals,
                               hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_ray_triangle, dim3(grid_for(n, 256)), dim3(256), 0, s, tri, rays, n, t_out, hit, normals);
    return hipGetLastError();
}
hipError_t launch_ray_plane(const float* plane, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_ray_plane, dim3(grid_for(n, 256)), dim3(256), 0, s, plane, rays, n, t_out, hit);
    return hipGetLastError();
}
hipError_t launch_ray_box(const float* box, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, uint8_t* inside,
                          hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_ray_box, dim3(grid_for(n, 256)), dim3(256), 0, s, box, rays, n, t_out, hit, inside);
    return hipGetLastError();
}
hipError_t launch_ray_sphere(const float* sph, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, float* normals,
                             hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_ray_sphere, dim3(grid_for(n, 256)), dim3(256), 0, s, sph, rays, n, t_out, hit, normals);
    return hipGetLastError();
}
hipError_t launch_triangle_plane(const float* tri, unsigned long long n, float* plane, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_triangle_plane, dim3(grid_for(n, 256)), dim3(256), 0, s, tri, n, plane);
    return hipGetLastError();
}
hipError_t launch_point_in_triangle(const float* in, unsigned long long n, uint8_t* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_point_in_triangle, dim3(grid_for(n, 256)), dim3(256), 0, s, in, n, out);
    return hipGetLastError();
}

}  // namespace cgrt
