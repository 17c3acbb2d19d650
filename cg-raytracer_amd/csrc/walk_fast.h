// walk_fast.h -- the CERTIFIED walk: the same answer as the exact walk of walk_exact.h (bit for bit: flag, t, record),
// found on a better search structure and then proven to be the reference's answer.
//
// Why it is allowed.  For a ray, let G be the set of triangles intersectRayWithTriangle (ray_tracing.cpp:86-114) accepts
// on geometry alone: den != 0, t >= 0 (or -0), the hit point passes pointInTriangle.  None of this depends on ray.t.
// The reference's walk (bvh.cpp:572-758, :535-553) returns the smallest t over the triangles of G it actually tests with
// t < ray.t, and ray.t is monotone: at any moment it is >= t* = the minimum over ALL of G (below the initial ray.t).  So:
//   (1) if no triangle of G lies below the initial ray.t, the reference misses too (it tests a subset);
//   (2) otherwise let T* be the triangle with the unique minimum t*.  The reference reaches T*'s leaf iff on the way
//       down every node of the path is entered: origin strictly inside its box (startsInBox, bvh.cpp:647-661), or the
//       box test (ray_tracing.cpp:162-200) passes with its parameter `cur` below the ray.t of that moment -- and, when
//       the child was deferred, ray.t has not dropped below `cur` = tSecond by the time it is popped (:581-585).  ray.t >= t*
//       throughout, so "cur < t*" at every node of the path implies all of that, whatever the walk met before; in the leaf
//       T* is then accepted (t* < ray.t unless an equal-t triangle came first) and nothing later beats it.
// The certificate is therefore: a strict unique minimum (any equal-t acceptance anywhere -> not certified), no
// origin-on-plane acceptance anywhere (ray_tracing.cpp:43-47 has no t < ray.t guard: visit order decides), and
// inside-or-(exact box test at t*) for the <= 11 boxes of T*'s path (SceneDev::paths) -- the reference's own arithmetic
// (ray_box_fast, exact quotients).  A ray that is not certified is simply walked by walk_tree_unified: the certified walk
// can only be slower, never different.  F4's false misses fail a path box and take that route.
//
// The search itself runs on the fast tree (bvh_builder.cpp build_fast_tree): 4-wide nodes over the reference leaves, then
// the leaves' own accelerators, conservative slab tests only (slab_cons, the argument of walk_exact.h RayPre), any order.
#pragma once
#include "walk_exact.h"

namespace cgrt {

struct FastScan {
    float best_t;       // running minimum, starts at ray.t
    uint32_t best_rec;  // its record (REF_NONE: nothing accepted yet)
    bool tie;           // another accepted triangle has exactly best_t
    bool onp;           // an origin-on-plane acceptance was seen (sticky)
};

__device__ __forceinline__ void fast_apply(const TriEval& E, const uint32_t rec, FastScan& F) {
    const bool valid = !E.onp && E.den_ok && !(E.tt < 0) && E.inside;
    F.onp = F.onp || (E.onp && E.inside);
    const bool lt = valid && (E.tt < F.best_t);
    const bool eq = valid && (E.tt == F.best_t) && (F.best_rec != REF_NONE);  // equal to the INITIAL ray.t is a plain reject (:65)
    F.tie = lt ? false : (F.tie || eq);
    F.best_t = lt ? E.tt : F.best_t;
    F.best_rec = lt ? rec : F.best_rec;
}

// records [first, first + n): runs of the accelerators (1..2 by default) and small leaves without accelerator (<= 32)
template <bool COUNT>
__device__ __forceinline__ void fast_test_run(const SceneDev& S, const uint32_t first, const uint32_t n, const F3 o, const F3 d, FastScan& F,
                                              LaneCounters& cnt) {
    if (COUNT) {
        cnt.tri += n;
        if (first_active_lane()) cnt.w_tri++;
    }
    const float4* q = reinterpret_cast<const float4*>(S.tris + first);
    if (n <= 2) {  // both records evaluated without early exits, so that their loads and arithmetic interleave
        const uint32_t j = (n > 1) ? 4u : 0u;
        const float4 a0 = q[0], b0 = q[1], c0 = q[2], e0 = q[3];
        const float4 a1 = q[j], b1 = q[j + 1], c1 = q[j + 2], e1 = q[j + 3];
        const TriEval E0 = eval_record(a0, b0, c0, e0, o, d);
        const TriEval E1 = eval_record(a1, b1, c1, e1, o, d);
        fast_apply(E0, first, F);
        if (n > 1) fast_apply(E1, first + 1, F);
    } else {
        for (uint32_t i = 0; i < n; i++) fast_apply(eval_record(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3], o, d), first + i, F);
    }
}

// The certificate's box part for record `rec` accepted at parameter t: every box on the reference's way to the record's
// leaf is entered at t (see the header).  Reference arithmetic: ray_box_fast under RayFast::fd.
template <bool COUNT>
__device__ __forceinline__ bool path_certified(const SceneDev& S, const Walk& W, const uint32_t rec, const float t, LaneCounters& cnt) {
    const uint32_t leaf = S.tri_leaf[rec - S.tri_base];
    const uint32_t n = S.leaves[leaf].path_len;
    const float2* pb = reinterpret_cast<const float2*>(S.paths + (size_t)leaf * (PATH_BOXES * 6));
    if (COUNT) cnt.cert += n;
    bool ok = true;
    for (uint32_t i = 0; i < n; i++) {
        const float2 a = pb[3 * i], b = pb[3 * i + 1], c = pb[3 * i + 2];
        float tb;
        bool inside;
        const bool hit = ray_box_fast<true>(f3(a.x, a.y, b.x), f3(b.y, c.x, c.y), W.o, W.d, W.R, t, tb, inside);
        ok = ok && (hit || inside);
    }
    return ok;
}

// Precondition: walk_begin(S, W) returned true, S.fast_root != REF_NONE, W.P.regular and W.R.fd.
// Returns true when (W.t, W.hit_rec) now hold the reference's result; false: W is untouched, take the exact walk.
// ANYHIT (the caller only needs the hit FLAG): the search stops at the first accepted triangle; if that triangle's path is
// certified the reference's flag is set too -- had the reference accepted nothing, ray.t would still be the initial one and
// it would reach that leaf and accept the triangle.
template <bool COUNT, bool ANYHIT>
__device__ __forceinline__ bool walk_fast(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    const F3 o = W.o, d = W.d;
    FastScan F;
    F.best_t = W.t;
    F.best_rec = REF_NONE;
    F.tie = false;
    F.onp = false;
    uint32_t cur = S.fast_root;
    int sp = 0;
    bool done = false;
    while (!done) {
        if (cur != REF_NONE && !(cur & REF_LEAF)) sub_node_step<COUNT>(S, W.P, F.best_t, cur, sp, stk, cnt);
        if (cur != REF_NONE && !(cur & REF_LEAF)) sub_node_step<COUNT>(S, W.P, F.best_t, cur, sp, stk, cnt);
        if (cur != REF_NONE && (cur & REF_LEAF)) {
            fast_test_run<COUNT>(S, run_first(cur), run_count(cur), o, d, F, cnt);
            cur = REF_NONE;
            if (ANYHIT && (F.best_rec != REF_NONE || F.onp)) done = true;
        }
        if (cur == REF_NONE && !done) {
            if (sp > 0) {
                sp -= 1;
                cur = stk[sp * CGRT_BLOCK];
            } else {
                done = true;
            }
        }
    }
    if (F.onp || F.tie) return false;
    if (F.best_rec != REF_NONE) {
        if (!path_certified<COUNT>(S, W, F.best_rec, F.best_t, cnt)) return false;
        W.t = F.best_t;
        W.hit_rec = F.best_rec;
    }
    return true;
}

// BoundingVolumeHierarchy::intersect's mesh part (bvh.cpp:870-875) for one ray: root gate, then the certified walk when the
// scene has a fast tree and the ray lies inside both envelopes (RayPre::regular, RayFast::fd), the exact walk otherwise
// or when no certificate was obtained.
// ANYHIT (exact walk): stop after the first leaf that accepted a triangle.  The walk up to there is the reference's, so the
// hit FLAG is the reference's (some acceptance happens upstream iff one happens in the first leaf that has one); t and the
// record are those of that leaf, not the final ones.
template <bool COUNT, bool FAST, bool ANYHIT = false>
__device__ __forceinline__ void walk_tree(const SceneDev& S, const F3 o, const F3 d, float& t, uint32_t& hit_rec,
                                          uint32_t* __restrict__ stk, LaneCounters& cnt) {
    Walk W;
    W.o = o;
    W.d = d;
    W.t = t;
    if (walk_begin(S, W)) {
        if (COUNT) cnt.entered++;
        bool certified = false;
        if (FAST && W.P.regular && W.R.fd) {
            certified = walk_fast<COUNT, ANYHIT>(S, W, stk, cnt);
            if (COUNT && !certified) cnt.fallback++;
        }
        if (!certified) walk_tree_unified<COUNT, ANYHIT>(S, W, stk, cnt);
    }
    t = W.t;
    hit_rec = W.hit_rec;
}

}  // namespace cgrt
