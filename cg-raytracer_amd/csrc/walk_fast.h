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
//       the child was deferred, ray.t has not dropped below `cur` = tSecond by the time it is popped (:581-585).  Until T* is
//       accepted ray.t is the initial value or the t of some OTHER accepted triangle, all of them strictly above t* (the
//       minimum is unique), so "cur <= t*" at every node of the path implies all of that, whatever the walk met before --
//       including cur == t* exactly, the case of an axis-aligned face in its zero-thickness box; in the leaf T* is then
//       accepted (t* < ray.t) and nothing later beats it.
// The certificate is therefore: a strict unique minimum (any equal-t acceptance anywhere -> not certified), no
// origin-on-plane acceptance anywhere (ray_tracing.cpp:43-47 has no t < ray.t guard: visit order decides), and
// inside-or-(exact box test with cur <= t*) for the <= 11 boxes of T*'s path (SceneDev::paths) -- the reference's own arithmetic
// (ray_box_fast, exact quotients) -- of which only those that do not contain their successor on the path need the test
// (path_certified: a box containing an entered box is entered; in practice the leaf's own box alone).  A ray that is not
// certified is simply walked by walk_tree_unified: the certified walk can only be slower, never different.  F4's false misses
//  fail a path box and take that route.
//
// The search itself runs on the fast tree (bvh_builder.cpp build_fast_tree): 4-wide nodes over the reference leaves, then
// the leaves' own accelerators, conservative slab tests only (slab_cons, the argument of walk_exact.h RayPre), any order.
#pragma once
#include "walk_exact.h"

namespace cgrt {

// What the caller needs from the search:
//   WALK_CLOSEST  the closest accepted triangle (BoundingVolumeHierarchy::intersect proper);
//   WALK_ANYHIT   only the hit FLAG: the search stops at the first accepted triangle whose path is certified -- had the
//                 reference accepted nothing, ray.t would still be the initial one and it would reach that leaf and accept it;
//   WALK_OCCLUDED pointInShadow's question (main.cpp:104-135): is there a hit with `ray.t + epsilon < |fromPosToLight|`?
//                 fl(t + eps) is monotone in t, so the reference answers yes iff its closest t qualifies, and ANY accepted
//                 triangle T that qualifies AND whose path is certified settles it: the reference either accepts T or holds
//                 a ray.t <= t_T by then (a box culled with cur <= t_T means ray.t <= cur), so its final t qualifies too.  No qualifying triangle anywhere -> the reference's
//                 hit, if any, does not qualify either (it tests a subset): reported as a miss, which is what the caller's
//                 test `hit && !(t + eps >= dist)` makes of it.  The search is bounded by the light's distance.
enum { WALK_CLOSEST = 0, WALK_ANYHIT = 1, WALK_OCCLUDED = 2 };
#define CGRT_SHADOW_EPS 0.001f  // main.cpp:110

struct FastScan {
    float best_t;       // running minimum, starts at ray.t
    uint32_t best_rec;  // its record (REF_NONE: nothing accepted yet)
    bool tie;           // another accepted triangle has exactly best_t
    bool onp;           // an origin-on-plane acceptance was seen (sticky)
};

template <int MODE>
__device__ __forceinline__ void fast_apply(const TriEval& E, const uint32_t rec, const float qlen, FastScan& F) {
    const bool valid = !E.onp && E.den_ok && !(E.tt < 0) && E.inside && (MODE != WALK_OCCLUDED || !(E.tt + CGRT_SHADOW_EPS >= qlen));
    F.onp = F.onp || (E.onp && E.inside);
    const bool lt = valid && (E.tt < F.best_t);
    const bool eq = valid && (E.tt == F.best_t) && (F.best_rec != REF_NONE);  // equal to the INITIAL ray.t is a plain reject (:65)
    F.tie = lt ? false : (F.tie || eq);
    F.best_t = lt ? E.tt : F.best_t;
    F.best_rec = lt ? rec : F.best_rec;
}

// records [first, first + n): runs of the accelerators (1..2 by default) and small leaves without accelerator (<= 32)
template <bool COUNT, int MODE>
__device__ __forceinline__ void fast_test_run(const SceneDev& S, const uint32_t first, const uint32_t n, const F3 o, const F3 d, const float qlen,
                                              FastScan& F, LaneCounters& cnt) {
    if (COUNT) {
        cnt.tri += n;
        if (first_active_lane()) cnt.w_tri++;
    }
    const float4* q = reinterpret_cast<const float4*>(S.tris + first);
    if (n <= 2) {  // both records evaluated without early exits, so that their loads and arithmetic interleave
        const uint32_t j = (n > 1) ? 4u : 0u;
        const float4 a0 = q[0], b0 = q[1], c0 = q[2], e0 = q[3];
        const float4 a1 = q[j], b1 = q[j + 1], c1 = q[j + 2], e1 = q[j + 3];
        TriEval E0, E1;
        eval_pair(a0, b0, c0, e0, a1, b1, c1, e1, o, d, E0, E1);
        fast_apply<MODE>(E0, first, qlen, F);
        if (n > 1) fast_apply<MODE>(E1, first + 1, qlen, F);
    } else {
        for (uint32_t i = 0; i < n; i++) fast_apply<MODE>(eval_record(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3], o, d), first + i, qlen, F);
    }
}

// The certificate's box part for record `rec` accepted at parameter t: every box on the reference's way to the record's
// leaf is entered at t (see the header).  Reference arithmetic: ray_box_fast under RayFast::fd, ray_box + starts_in_box outside it.
template <bool COUNT>
__device__ __forceinline__ bool path_certified(const SceneDev& S, const Walk& W, const uint32_t rec, const float t, LaneCounters& cnt) {
    const uint32_t leaf = S.tri_leaf[rec - S.tri_base];
    const uint32_t pl = S.leaves[leaf].path_len;
    const float2* pb = reinterpret_cast<const float2*>(S.paths + (size_t)leaf * (PATH_BOXES * 6));
    const RayFast R = make_rayfast(S, W.o, W.d);  // (recomputed here rather than kept in registers through the search)
    // Inside RayFast::fd (finite operands, no zero direction component: no NaN, no infinity) a box that CONTAINS a box entered
    // at t is entered at t as well: per axis fl(lo - o) and fl(x / d) are monotone in lo, so tIn can only fall and tOut only
    // rise from the inner box to the outer one; then either tIn >= 0 and cur = tIn <= the inner box's, or tIn < 0 <= tOut, which
    // puts the origin strictly inside the outer box unless it lies on one of its exit planes, where cur = tOut = 0.  The
    // builder marks the boxes that do not contain their successor on the path (LeafRec::path_len bits 8..; the leaf's own
    // box always): only those are tested.  Outside fd the ternary ladders see NaNs and every box is tested.
    uint32_t need = R.fd ? (pl >> 8) : ((1u << (pl & 0xffu)) - 1u);
    if (COUNT) cnt.cert += __popc(need);
    bool ok = true;
    while (need) {
        const uint32_t i = (uint32_t)__ffs((int)need) - 1u;
        need &= need - 1u;
        const float2 a = pb[3 * i], b = pb[3 * i + 1], c = pb[3 * i + 2];
        const F3 lo = f3(a.x, a.y, b.x), hi = f3(b.y, c.x, c.y);
        float tb;
        bool inside, geom;
        // the geometric part of the box test with the reference's arithmetic (t = +inf: no `cur >= ray.t` rejection), then
        // cur <= t: see the header -- every ray.t the reference can hold before it accepts T is strictly above t
        // (a NaN parameter -- 0 / 0 on a zero direction component -- is never rejected by `cur >= ray.t`, nor skipped on pop by
        // `ray.t < tSecond`: !(tb > t) says the same)
        if (R.fd) {
            geom = ray_box_fast<true>(lo, hi, W.o, W.d, R, __builtin_inff(), tb, inside);
        } else {  // outside the exact fast division's envelope: the reference's test as written (IEEE divisions, ternary ladders)
            tb = 0.0f;
            geom = ray_box(lo, hi, W.o, W.d, __builtin_inff(), tb);
            inside = starts_in_box(W.o, lo, hi);
        }
        ok = ok && (inside || (geom && !(tb > t)));
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Quad tail.  A frame ends with its hardest waves, and those spend most of their life with a handful of live rays whose
// lanes take turns (a node body for these, a run body for those) on an otherwise idle SIMD.  The certified search is
// order-free, so once a wave is down to <= 16 live rays each of them is spread over the four lanes of a quad: per round
// the quad takes up to four entries off the ray's stack (it stays in the owner's LDS slice), each lane steps through
// one of them -- a node's four boxes, or a run's triangles -- and the survivors go back on the stack, nearest last.  The
// four entries cannot cull one another inside a round (a little extra work), everything else is the lane-per-ray search:
// same boxes, same triangle arithmetic, the minimum (with its tie and on-plane flags) reduced over the quad.
// A stack that would outgrow its slice gives the ray up (-> exact walk), like any other missing certificate.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(const uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(const float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
#define CGRT_QP_XOR1 0xB1   // quad_perm [1,0,3,2]
#define CGRT_QP_XOR2 0x4E   // quad_perm [2,3,0,1]
#define CGRT_QP_BCAST(i) ((i) * 0x55)  // quad_perm [i,i,i,i]
#ifndef CGRT_QUAD_TAIL_RAYS
#define CGRT_QUAD_TAIL_RAYS 16  // a wave with at most this many live rays switches to the quad tail (0: never)
#endif

struct QuadState {  // quad-uniform: every lane of a quad holds the same values
    float best_t;
    uint32_t best_rec;
    bool tie, onp, failed;
};

// merge of two scan states that started from the same state (associative, commutative)
__device__ __forceinline__ void quad_merge(float& bt, uint32_t& br, bool& tie, const float obt, const uint32_t obr, const bool otie) {
    const bool take = obt < bt;
    const bool same = (obt == bt);
    tie = take ? otie : (same ? (tie || otie || (obr != br)) : tie);
    br = take ? obr : br;
    bt = take ? obt : bt;
}

// live: lanes that carry an unfinished ray (<= 16 of them); their state is (F, cur, sp) with the stack in their own slice.
// On return the owners hold their ray's final FastScan and `failed`.
template <bool COUNT, int MODE>
__device__ __forceinline__ void quad_tail(const SceneDev& S, const unsigned long long live, const bool mine, const Walk& W, const float qlen_own, FastScan& F,
                                          uint32_t cur, int sp, bool& failed, uint32_t* __restrict__ wave_stk, uint32_t* __restrict__ s_map,
                                          LaneCounters& cnt) {
    const int lane = threadIdx.x & 63;
    const int myrank = __popcll(live & ((1ull << lane) - 1ull));
    uint32_t* __restrict__ my = wave_stk + lane;
    if (mine) {
        if (cur != REF_NONE) {  // the node or run the lane stands on goes back on its stack
            my[sp * CGRT_STRIDE] = cur;
            sp += 1;
        }
        s_map[myrank] = (uint32_t)lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int Q = lane >> 2, q = lane & 3;
    const bool valid = Q < (int)__popcll(live);
    const int src = valid ? (int)s_map[Q] : lane;
    // the owner's ray, broadcast to its quad (all 64 lanes take part in the shuffles)
    RayPre P;
    P.inv = f3(__shfl(W.P.inv.x, src, 64), __shfl(W.P.inv.y, src, 64), __shfl(W.P.inv.z, src, 64));
    P.cx = (f2v){__shfl(W.P.cx.x, src, 64), __shfl(W.P.cx.y, src, 64)};
    P.cy = (f2v){__shfl(W.P.cy.x, src, 64), __shfl(W.P.cy.y, src, 64)};
    P.cz = (f2v){__shfl(W.P.cz.x, src, 64), __shfl(W.P.cz.y, src, 64)};
    const int sg = __shfl((int)((W.P.sx ? 1 : 0) | (W.P.sy ? 2 : 0) | (W.P.sz ? 4 : 0)), src, 64);
    P.sx = (sg & 1) != 0;
    P.sy = (sg & 2) != 0;
    P.sz = (sg & 4) != 0;
    P.regular = true;
    const F3 o = f3(__shfl(W.o.x, src, 64), __shfl(W.o.y, src, 64), __shfl(W.o.z, src, 64));
    const F3 d = f3(__shfl(W.d.x, src, 64), __shfl(W.d.y, src, 64), __shfl(W.d.z, src, 64));
    const float qlen = (MODE == WALK_OCCLUDED) ? __shfl(qlen_own, src, 64) : 0.0f;
    QuadState G;
    G.best_t = __shfl(F.best_t, src, 64);
    G.best_rec = __shfl(F.best_rec, src, 64);
    const int fl = __shfl((int)((F.tie ? 1 : 0) | (F.onp ? 2 : 0)), src, 64);
    G.tie = (fl & 1) != 0;
    G.onp = (fl & 2) != 0;
    G.failed = false;
    const int osp = __shfl(sp, src, 64);  // (unconditional: a shuffle only sees lanes that execute it)
    int qsp = valid ? osp : 0;
    uint32_t* __restrict__ stk = wave_stk + src;  // the owner's LDS slice
    const float inf = __builtin_inff();
    while (__any(qsp > 0)) {
        // entries this round: up to four, fewer when the survivors (at most 4 per node) might not fit
        int n = qsp < 4 ? qsp : 4;
        while (n > 0 && qsp + 3 * n > CGRT_STACK_SLOTS) n -= 1;
        if (qsp > 0 && n == 0) {
            G.failed = true;
            qsp = 0;
        }
        const bool has = q < n;
        const uint32_t ref = has ? stk[(qsp - 1 - q) * CGRT_STRIDE] : REF_NONE;
        qsp -= n;
        uint32_t r0 = REF_NONE, r1 = REF_NONE, r2 = REF_NONE, r3 = REF_NONE;
        int c = 0;
        float lt = G.best_t;
        uint32_t lr = G.best_rec;
        bool ltie = G.tie, lonp = false;
        // ONE round trip to memory per round: a node entry and a run entry are both "the 128 bytes this reference names" (a
        // 4-wide node; the two records of a run, the first one twice for a run of one), so every lane requests its eight
        // quarters before either kind is evaluated -- a quad whose entries are of both kinds no longer pays the latency twice
        // (profiles/r3_exp_unified_tail.txt).
        const bool is_node = has && !(ref & REF_LEAF), is_run = has && (ref & REF_LEAF) != 0u;
        const uint32_t rfirst = run_first(ref), rcount = is_run ? run_count(ref) : 0u;
        if (has) {
        const float4* g = is_node ? reinterpret_cast<const float4*>(S.subnodes + ref) : reinterpret_cast<const float4*>(S.tris + rfirst);
        const uint32_t j = (is_node || rcount > 1u) ? 4u : 0u;
        const float4 u0 = g[0], u1 = g[1], u2 = g[2], u3 = g[3];
        const float4 u4 = g[j], u5 = g[j + 1], u6 = g[j + 2], u7 = g[j + 3];
        if (is_node) {  // a 4-wide node: its four boxes, hit children sorted near to far
            if (COUNT) {
                cnt.sub++;
                if (first_active_lane()) cnt.w_sub++;
            }
            const uint4 m = make_uint4(__float_as_uint(u3.x), __float_as_uint(u3.y), __float_as_uint(u3.z), __float_as_uint(u3.w));
            float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
            slab_cons4(P, u0, u1, u2, u4, u5, u6, tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3);
            const float tc = fmaxf(G.best_t, 0.0f);
            float k0 = ((tn0 <= tf0) && (tf0 >= 0.0f) && (tn0 <= tc)) ? tn0 : inf;
            float k1 = ((tn1 <= tf1) && (tf1 >= 0.0f) && (tn1 <= tc)) ? tn1 : inf;
            float k2 = ((tn2 <= tf2) && (tf2 >= 0.0f) && (tn2 <= tc)) ? tn2 : inf;
            float k3 = ((tn3 <= tf3) && (tf3 >= 0.0f) && (tn3 <= tc)) ? tn3 : inf;
            r0 = m.x, r1 = m.y, r2 = m.z, r3 = m.w;
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
            CGRT_CSWAP(k1, r1, k3, r3)
            CGRT_CSWAP(k1, r1, k2, r2)
#undef CGRT_CSWAP
            c = (k0 < inf ? 1 : 0) + (k1 < inf ? 1 : 0) + (k2 < inf ? 1 : 0) + (k3 < inf ? 1 : 0);  // hit children are r0 .. r(c-1)
        } else {  // a run of records: the first two are in the registers already
            FastScan L;
            L.best_t = G.best_t;
            L.best_rec = G.best_rec;
            L.tie = G.tie;
            L.onp = false;
            if (COUNT) {
                cnt.tri += rcount;
                if (first_active_lane()) cnt.w_tri++;
            }
            TriEval E0, E1;
            eval_pair(u0, u1, u2, u3, u4, u5, u6, u7, o, d, E0, E1);
            fast_apply<MODE>(E0, rfirst, qlen, L);
            if (rcount > 1u) fast_apply<MODE>(E1, rfirst + 1u, qlen, L);
            if (rcount > 2u) {  // small leaves without accelerator (<= 32 records): the rest one by one
                const float4* g = reinterpret_cast<const float4*>(S.tris + rfirst);
                for (uint32_t i = 2; i < rcount; i++)
                    fast_apply<MODE>(eval_record(g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], o, d), rfirst + i, qlen, L);
            }
            lt = L.best_t;
            lr = L.best_rec;
            ltie = L.tie;
            lonp = L.onp;
        }
        }
        // survivors back on the stack: lane 0 held the top entry, so its children go on top (pushed last), and inside a
        // lane the nearest child last
        const int c0q = (int)dpp_u32<CGRT_QP_BCAST(0)>((uint32_t)c), c1q = (int)dpp_u32<CGRT_QP_BCAST(1)>((uint32_t)c);
        const int c2q = (int)dpp_u32<CGRT_QP_BCAST(2)>((uint32_t)c), c3q = (int)dpp_u32<CGRT_QP_BCAST(3)>((uint32_t)c);
        const int off = (q < 3 ? c3q : 0) + (q < 2 ? c2q : 0) + (q < 1 ? c1q : 0);
        int pos = qsp + off;
        if (c > 3) stk[(pos++) * CGRT_STRIDE] = r3;
        if (c > 2) stk[(pos++) * CGRT_STRIDE] = r2;
        if (c > 1) stk[(pos++) * CGRT_STRIDE] = r1;
        if (c > 0) stk[(pos++) * CGRT_STRIDE] = r0;
        qsp += c0q + c1q + c2q + c3q;
        // the scan state, reduced over the quad
        {
            float ot = dpp_f32<CGRT_QP_XOR1>(lt);
            uint32_t orr = dpp_u32<CGRT_QP_XOR1>(lr);
            uint32_t ofl = dpp_u32<CGRT_QP_XOR1>((ltie ? 1u : 0u) | (lonp ? 2u : 0u));
            quad_merge(lt, lr, ltie, ot, orr, (ofl & 1u) != 0);
            lonp = lonp || ((ofl & 2u) != 0);
            ot = dpp_f32<CGRT_QP_XOR2>(lt);
            orr = dpp_u32<CGRT_QP_XOR2>(lr);
            ofl = dpp_u32<CGRT_QP_XOR2>((ltie ? 1u : 0u) | (lonp ? 2u : 0u));
            quad_merge(lt, lr, ltie, ot, orr, (ofl & 1u) != 0);
            lonp = lonp || ((ofl & 2u) != 0);
        }
        G.best_t = lt;
        G.best_rec = lr;
        G.tie = ltie;
        G.onp = G.onp || lonp;
        if (MODE != WALK_CLOSEST && (G.best_rec != REF_NONE || G.onp)) qsp = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round reads what the other lanes of the quad just wrote
    }
    // results back to the owners (quad `myrank`, any of its lanes)
    const int ql = 4 * myrank;
    const float r_t = __shfl(G.best_t, ql, 64);
    const uint32_t r_rec = __shfl(G.best_rec, ql, 64);
    const int r_fl = __shfl((int)((G.tie ? 1 : 0) | (G.onp ? 2 : 0) | (G.failed ? 4 : 0)), ql, 64);
    if (mine) {
        F.best_t = r_t;
        F.best_rec = r_rec;
        F.tie = (r_fl & 1) != 0;
        F.onp = (r_fl & 2) != 0;
        failed = (r_fl & 4) != 0;
    }
}

// The certified search for a whole wave.  `alive`: the lane carries a ray that passed the root gate, lies inside both
// envelope (W.P.regular) and wants the certified walk; the other lanes only help in the quad tail.  Must be called
// by all 64 lanes (wave-uniform control flow).  Returns, per alive lane, true when (W.t, W.hit_rec) now hold the answer the
// MODE asks for (see WALK_*); false: W is untouched, take the exact walk.  qlen: WALK_OCCLUDED's |fromPosToLight|.
template <bool COUNT, int MODE>
__device__ __forceinline__ bool walk_fast_wave(const SceneDev& S, const bool alive, Walk& W, const float qlen, uint32_t* __restrict__ wave_stk,
                                               uint32_t* __restrict__ s_map, LaneCounters& cnt) {
    uint32_t* __restrict__ stk = wave_stk + (threadIdx.x & 63u);
    const F3 o = W.o, d = W.d;
    FastScan F;
    // the search bound: ray.t, and for an occlusion query the light's distance (t + eps < qlen needs t < qlen)
    F.best_t = (MODE == WALK_OCCLUDED) ? fminf(W.t, qlen) : W.t;
    F.best_rec = REF_NONE;
    F.tie = false;
    F.onp = false;
    uint32_t cur = alive ? S.fast_root : REF_NONE;
    int sp = 0;
    bool done = !alive, failed = false;
    for (;;) {
        const unsigned long long live = __ballot(!done);
        if (live == 0ull) break;
        if (CGRT_QUAD_TAIL_RAYS > 0 && __popcll(live) <= CGRT_QUAD_TAIL_RAYS) {
            quad_tail<COUNT, MODE>(S, live, !done, W, qlen, F, cur, sp, failed, wave_stk, s_map, cnt);
            break;
        }
        if (!done) {
            // A trip = [node step][node step][run test][pop].  Trips built around a node-or-run step with ONE load phase -- [node][U],
            // [U][U], [node][node][U] -- were measured slower on every frame (profiles/r3_exp_loop_shapes.txt): in the bulk the loop is
            // VALU-bound and such a step executes both bodies.
            if (cur != REF_NONE && !(cur & REF_LEAF)) sub_node_step<COUNT>(S, W.P, F.best_t, cur, sp, stk, cnt);
            if (cur != REF_NONE && !(cur & REF_LEAF)) sub_node_step<COUNT>(S, W.P, F.best_t, cur, sp, stk, cnt);
            if (cur != REF_NONE && (cur & REF_LEAF)) {
                fast_test_run<COUNT, MODE>(S, run_first(cur), run_count(cur), o, d, qlen, F, cnt);
                cur = REF_NONE;
                if (MODE != WALK_CLOSEST && (F.best_rec != REF_NONE || F.onp)) done = true;
            }
            if (cur == REF_NONE && !done) {
                if (sp > 0) {
                    sp -= 1;
                    cur = stk[sp * CGRT_STRIDE];
                } else {
                    done = true;
                }
            }
        }
    }
    if (!alive || failed || F.onp || (MODE == WALK_CLOSEST && F.tie)) return false;  // (a tie changes the record, not the flag)
    if (F.best_rec != REF_NONE) {
        if (!path_certified<COUNT>(S, W, F.best_rec, F.best_t, cnt)) return false;
        W.t = F.best_t;
        W.hit_rec = F.best_rec;
    }
    return true;
}

// BoundingVolumeHierarchy::intersect's mesh part (bvh.cpp:870-875) for the rays of a wave (one per `active` lane; call
// with all 64 lanes; wave_stk / s_map: CGRT_WAVE_STACK / CGRT_WAVE_MAP of the kernel's dynamic LDS): root gate, then the
// certified walk when the scene has a fast tree and the ray lies inside the search's envelope (RayPre::regular), the exact
// walk otherwise or when no certificate was obtained.
// MODE (see WALK_*): the exact walk answers WALK_ANYHIT by stopping after the first leaf that accepted a triangle -- the walk
// up to there is the reference's, so the hit FLAG is the reference's (some acceptance happens upstream iff one happens in
// the first leaf that has one); t and the record are those of that leaf, not the final ones -- and WALK_OCCLUDED with the
// full closest hit.
template <bool COUNT, bool FAST, int MODE = WALK_CLOSEST>
__device__ __forceinline__ void walk_tree(const SceneDev& S, const bool active, const F3 o, const F3 d, float& t, uint32_t& hit_rec,
                                          uint32_t* __restrict__ wave_stk, uint32_t* __restrict__ s_map, LaneCounters& cnt,
                                          const float qlen = 0.0f) {
    Walk W;
    W.o = o;
    W.d = d;
    W.t = t;
    W.hit_rec = REF_NONE;
    const bool entered = active && walk_begin(S, W);
    if (COUNT && entered) cnt.entered++;
    bool certified = false;
    if (FAST) {
        const bool eligible = entered && W.P.regular;
        if (__any(eligible)) certified = walk_fast_wave<COUNT, MODE>(S, eligible, W, qlen, wave_stk, s_map, cnt);
        if (COUNT && eligible && !certified) cnt.fallback++;
    }
    if (entered && !certified) {
        if (FAST) {  // (the per-ray constants of the exact walk are rebuilt rather than kept alive through the search)
            W.P = make_raypre(S, W.o, W.d, W.t);
            W.R = make_rayfast(S, W.o, W.d);
        }
        walk_tree_unified<COUNT, MODE == WALK_ANYHIT>(S, W, wave_stk + (threadIdx.x & 63u), cnt);
    }
    t = W.t;
    hit_rec = W.hit_rec;
}

}  // namespace cgrt
