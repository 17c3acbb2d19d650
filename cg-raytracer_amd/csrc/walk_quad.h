// walk_quad.h -- the certified search of walk_fast.h with FOUR LANES PER RAY: the latency-regime shape of the traversal
// kernels, chosen per launch (trace_kernels.hip quad_shape_for) when a launch puts few rays on a SIMD.
//
// Why a second shape.  A small launch -- a 1/8 share of a 4K frame, a list of a few thousand shadow rays, the tail of any
// frame -- ends when its hardest RAYS end, and a hard ray is a dependent chain of ~100 steps (record load -> box or triangle
// arithmetic -> stack) that one lane walks alone: ~1100 cycles per step on an otherwise idle SIMD, of which ~500 are the
// issue slots of the ~130 VALU instructions a 4-wide node step costs ONE lane (a wave64 instruction occupies the SIMD for
// four cycles however few lanes are live).  Here a ray owns a quad: lane k of the quad tests child box k of the node (3
// packed fma + min/max instead of 12), the four children are ranked with three quad-permute compares, the survivors go on the
// ray's LDS stack with one write per lane, and a run's triangles are tested one per lane.  A node step is ~45 instructions
// per wave, 4 load instructions of 8 bytes per lane on ONE 128-byte line per quad (instead of seven 16-byte loads on 64
// lines per wave), and a wave carries 16 rays instead of 64, so the hard rays of a tile no longer wait for each other's
// bodies.  The price is throughput -- per-ray set-up (ray generation, root gate, per-ray constants, certificate, normal) is
// evaluated by all four lanes -- which is why large launches keep the lane-per-ray shape.
//
// The SEARCH is the one of walk_fast.h -- same tree, same conservative slab test (slab_cons), same triangle arithmetic
// (eval_record), same scan rule (fast_apply, quad_merge), same certificate (path_certified), exact walk as fallback -- and it
// is order-free, so the answer is bit-identical whatever the shape (tests/test_certified.py, tests/test_quad_shape_gpu.py).
//
// Wide tail: once a wave is down to <= 4 live rays each of them takes a whole row of 16 lanes: up to four entries of its stack
// per round, one per quad of the row, four boxes each (the 4 entries x 4 boxes of VERDICT r2 item 1(ii)).
#pragma once
#include "walk_fast.h"

namespace cgrt {

#define CGRT_QP_XOR3 0x1B  // quad_perm [3,2,1,0]
// Per-ray stack of the quad shape: FAST_STACK_ENTRIES deferred children at most (three per level), an odd stride so that the 16
// quads of a wave fall on different LDS banks.  The region aliases the lane-per-ray stacks (CGRT_WAVE_STACK): the exact
// fallback only starts when the search is over.
#define CGRT_QSLOTS 35
static_assert(CGRT_QSLOTS >= FAST_STACK_ENTRIES + 2, "quad stack too small");
static_assert(16 * CGRT_QSLOTS <= 64 * CGRT_STACK_SLOTS, "the quad stacks must fit the wave's LDS region");
#ifndef CGRT_QUAD_WIDE_RAYS
#define CGRT_QUAD_WIDE_RAYS 4  // a wave with at most this many live rays gives each a row of 16 lanes (0: never)
#endif

// One node step for the ray of this quad: lane q tests child box q.  On return (quad-uniform) cur = the nearest hit child or
// REF_NONE, the other hit children are on the stack, nearest on top.
template <bool COUNT>
__device__ __forceinline__ void quad_node_step(const SceneDev& S, const RayPre& P, const float best_t, const int q, uint32_t& cur, int& sp,
                                               uint32_t* __restrict__ stk, LaneCounters& cnt) {
    if (COUNT && q == 0) {
        cnt.sub++;
        if (first_active_lane()) cnt.w_sub++;
    }
    const uint32_t* base = reinterpret_cast<const uint32_t*>(S.subnodes + cur);
    // child q's box: words 0..5 / 6..11 of the node's first half, 16..21 / 22..27 of its second (cgrt_layout.h SubNode), 8-byte aligned
    const float2* b = reinterpret_cast<const float2*>(base + ((q & 1) * 6 + (q >> 1) * 16));
    const float2 bx = b[0], by = b[1], bz = b[2];
    const uint32_t ref = base[12 + q];  // the four child references sit in words 12..15 of the first half
    float tn, tf;
    slab_cons(P, (f2v){bx.x, bx.y}, (f2v){by.x, by.y}, (f2v){bz.x, bz.y}, tn, tf);
    const float inf = __builtin_inff();
    const float tc = fmaxf(best_t, 0.0f);  // never below 0: an origin-on-plane acceptance ignores ray.t (walk_exact.h sub_node_step)
    const bool hit = (tn <= tf) && (tf >= 0.0f) && (tn <= tc) && (tn < inf);
    const float key = hit ? tn : inf;
    // rank among the quad's hit children by (entry parameter, child index): three quad permutes
    const float k1 = dpp_f32<CGRT_QP_XOR1>(key), k2 = dpp_f32<CGRT_QP_XOR2>(key), k3 = dpp_f32<CGRT_QP_XOR3>(key);
    const bool hi1 = (q & 1) != 0, hi2 = (q & 2) != 0;  // the partner's index is below mine
    const int rank = ((k1 < key || (k1 == key && hi1)) ? 1 : 0) + ((k2 < key || (k2 == key && hi2)) ? 1 : 0) + ((k3 < key || (k3 == key && hi2)) ? 1 : 0);
    uint32_t c = hit ? 1u : 0u;
    c += dpp_u32<CGRT_QP_XOR1>(c);
    c += dpp_u32<CGRT_QP_XOR2>(c);
    // deferred children: rank 1 .. c-1, the nearest of them on top
    if (hit && rank > 0) stk[sp + (int)c - 1 - rank] = ref;
    uint32_t v = (hit && rank == 0) ? ref : 0u;
    v |= dpp_u32<CGRT_QP_XOR1>(v);
    v |= dpp_u32<CGRT_QP_XOR2>(v);
    cur = c > 0u ? v : REF_NONE;
    sp += c > 1u ? (int)c - 1 : 0;
}

// The scan state of a quad after each lane applied its own records to a copy of the shared state: reduced over the quad.
__device__ __forceinline__ void quad_reduce(FastScan& G, float lt, uint32_t lr, bool ltie, bool lonp) {
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
    G.best_t = lt;
    G.best_rec = lr;
    G.tie = ltie;
    G.onp = G.onp || lonp;
}

// A run of records for the ray of this quad: record i of every group of four goes to lane i.
template <bool COUNT, int MODE>
__device__ __forceinline__ void quad_run_step(const SceneDev& S, const uint32_t ref, const int q, const F3 o, const F3 d, const float qlen, FastScan& G,
                                              LaneCounters& cnt) {
    const uint32_t first = run_first(ref), n = run_count(ref);
    if (COUNT && q == 0) {
        cnt.tri += n;
        if (first_active_lane()) cnt.w_tri++;
    }
    for (uint32_t g0 = 0; g0 < n; g0 += 4) {
        FastScan L = G;
        L.onp = false;
        const uint32_t i = g0 + (uint32_t)q;
        if (i < n) {
            const float4* p = reinterpret_cast<const float4*>(S.tris + first + i);
            const float4 a = p[0], b = p[1], c = p[2], e = p[3];
            fast_apply<MODE>(eval_record(a, b, c, e, o, d), first + i, qlen, L);
        }
        quad_reduce(G, L.best_t, L.best_rec, L.tie, L.onp);
    }
}

// ---- wide tail: <= 4 live rays, a row of 16 lanes each (four quads: one stack entry per quad, one child box per lane) ----
__device__ __forceinline__ float shfl_f(const float v, const int src) { return __shfl(v, src, 64); }
template <bool COUNT, int MODE>
__device__ __forceinline__ void quad_wide_tail(const SceneDev& S, const unsigned long long live_q0, const bool mine, const Walk& W, const float qlen_own,
                                               FastScan& F, uint32_t cur, int sp, bool& failed, uint32_t* __restrict__ wave_stk, LaneCounters& cnt) {
    // live_q0: ballot of the q == 0 lanes of the live quads (<= 4 bits).  Row r serves the r-th of them.
    const int lane = threadIdx.x & 63, q = lane & 3, e = (lane >> 2) & 3, row = lane >> 4;
    uint32_t* __restrict__ own = wave_stk + (lane >> 2) * CGRT_QSLOTS;
    if (mine && cur != REF_NONE) {  // the node or run the quad stands on goes back on its stack (all four lanes write the same word)
        own[sp] = cur;
        sp += 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // the row's source lane: lane q0 of the row-th live quad
    unsigned long long m = live_q0;
    int src = lane;
    const int nlive = (int)__popcll(live_q0);
    for (int r = 0; r < row && m; r++) m &= m - 1ull;
    const bool valid = row < nlive;
    if (valid) src = __ffsll((long long)m) - 1;
    RayPre P;
    P.inv = f3(shfl_f(W.P.inv.x, src), shfl_f(W.P.inv.y, src), shfl_f(W.P.inv.z, src));
    P.cx = (f2v){shfl_f(W.P.cx.x, src), shfl_f(W.P.cx.y, src)};
    P.cy = (f2v){shfl_f(W.P.cy.x, src), shfl_f(W.P.cy.y, src)};
    P.cz = (f2v){shfl_f(W.P.cz.x, src), shfl_f(W.P.cz.y, src)};
    const int sg = __shfl((int)((W.P.sx ? 1 : 0) | (W.P.sy ? 2 : 0) | (W.P.sz ? 4 : 0)), src, 64);
    P.sx = (sg & 1) != 0;
    P.sy = (sg & 2) != 0;
    P.sz = (sg & 4) != 0;
    P.regular = true;
    const F3 o = f3(shfl_f(W.o.x, src), shfl_f(W.o.y, src), shfl_f(W.o.z, src));
    const F3 d = f3(shfl_f(W.d.x, src), shfl_f(W.d.y, src), shfl_f(W.d.z, src));
    const float qlen = (MODE == WALK_OCCLUDED) ? shfl_f(qlen_own, src) : 0.0f;
    FastScan G;
    G.best_t = shfl_f(F.best_t, src);
    G.best_rec = (uint32_t)__shfl((int)F.best_rec, src, 64);
    const int fl = __shfl((int)((F.tie ? 1 : 0) | (F.onp ? 2 : 0)), src, 64);
    G.tie = (fl & 1) != 0;
    G.onp = (fl & 2) != 0;
    bool gfailed = false;
    const int osp = __shfl(sp, src, 64);
    int rsp = valid ? osp : 0;  // row-uniform
    uint32_t* __restrict__ stk = wave_stk + (src >> 2) * CGRT_QSLOTS;  // the owner's stack
    const int rbase = lane & ~15;
    while (__any(rsp > 0)) {
        // entries this round: up to four, fewer when the survivors (at most 4 per node) might not fit
        int n = rsp < 4 ? rsp : 4;
        while (n > 0 && rsp + 3 * n > CGRT_QSLOTS) n -= 1;
        if (rsp > 0 && n == 0) {
            gfailed = true;
            rsp = 0;
        }
        const bool has = e < n;
        const uint32_t ref = has ? stk[rsp - 1 - e] : REF_NONE;
        rsp -= n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every entry is read before a survivor overwrites its slot
        // node entries: lane q tests child q, children ranked inside the quad
        uint32_t c = 0u, cref = 0u;
        int rank = 0;
        bool hit = false;
        if (has && !(ref & REF_LEAF)) {
            if (COUNT && q == 0) {
                cnt.sub++;
                if (first_active_lane()) cnt.w_sub++;
            }
            const uint32_t* base = reinterpret_cast<const uint32_t*>(S.subnodes + ref);
            const float2* b = reinterpret_cast<const float2*>(base + ((q & 1) * 6 + (q >> 1) * 16));
            const float2 bx = b[0], by = b[1], bz = b[2];
            cref = base[12 + q];
            float tn, tf;
            slab_cons(P, (f2v){bx.x, bx.y}, (f2v){by.x, by.y}, (f2v){bz.x, bz.y}, tn, tf);
            const float inf = __builtin_inff();
            const float tc = fmaxf(G.best_t, 0.0f);
            hit = (tn <= tf) && (tf >= 0.0f) && (tn <= tc) && (tn < inf);
            const float key = hit ? tn : inf;
            const float k1 = dpp_f32<CGRT_QP_XOR1>(key), k2 = dpp_f32<CGRT_QP_XOR2>(key), k3 = dpp_f32<CGRT_QP_XOR3>(key);
            const bool hi1 = (q & 1) != 0, hi2 = (q & 2) != 0;
            rank = ((k1 < key || (k1 == key && hi1)) ? 1 : 0) + ((k2 < key || (k2 == key && hi2)) ? 1 : 0) + ((k3 < key || (k3 == key && hi2)) ? 1 : 0);
            c = hit ? 1u : 0u;
            c += dpp_u32<CGRT_QP_XOR1>(c);
            c += dpp_u32<CGRT_QP_XOR2>(c);
        }
        // run entries: record i of every group of four goes to lane i; the state starts from the row's and is merged below
        FastScan L = G;
        L.onp = false;
        if (has && (ref & REF_LEAF)) {
            const uint32_t first = run_first(ref), nr = run_count(ref);
            if (COUNT && q == 0) {
                cnt.tri += nr;
                if (first_active_lane()) cnt.w_tri++;
            }
            for (uint32_t i = (uint32_t)q; i < nr; i += 4) {
                const float4* p = reinterpret_cast<const float4*>(S.tris + first + i);
                const float4 a = p[0], b = p[1], cc = p[2], ee = p[3];
                fast_apply<MODE>(eval_record(a, b, cc, ee, o, d), first + i, qlen, L);
            }
        }
        // survivors back on the stack: quad 0 held the top entry, so its children go on top, and inside a quad the nearest on top
        const int c0 = __shfl((int)c, rbase + 0, 64), c1 = __shfl((int)c, rbase + 4, 64), c2 = __shfl((int)c, rbase + 8, 64), c3 = __shfl((int)c, rbase + 12, 64);
        const int off = (e < 3 ? c3 : 0) + (e < 2 ? c2 : 0) + (e < 1 ? c1 : 0);
        if (hit) stk[rsp + off + (int)c - 1 - rank] = cref;
        rsp += c0 + c1 + c2 + c3;
        // the scan state, reduced over the row: inside the quad by DPP, then across the four quads
        {
            float lt = L.best_t;
            uint32_t lr = L.best_rec;
            bool ltie = L.tie, lonp = L.onp;
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
            for (int x = 4; x <= 8; x <<= 1) {
                ot = __shfl_xor(lt, x, 64);
                orr = (uint32_t)__shfl_xor((int)lr, x, 64);
                ofl = (uint32_t)__shfl_xor((int)((ltie ? 1u : 0u) | (lonp ? 2u : 0u)), x, 64);
                quad_merge(lt, lr, ltie, ot, orr, (ofl & 1u) != 0);
                lonp = lonp || ((ofl & 2u) != 0);
            }
            G.best_t = lt;
            G.best_rec = lr;
            G.tie = ltie;
            G.onp = G.onp || lonp;
        }
        if (MODE != WALK_CLOSEST && (G.best_rec != REF_NONE || G.onp)) rsp = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next round reads what the other lanes of the row just wrote
    }
    // results back to the owners: row `myrow` (the rank of my quad among the live ones), any of its lanes
    const int myrow = (int)__popcll(live_q0 & ((1ull << (lane & ~3)) - 1ull));
    const float r_t = shfl_f(G.best_t, 16 * myrow);
    const uint32_t r_rec = (uint32_t)__shfl((int)G.best_rec, 16 * myrow, 64);
    const int r_fl = __shfl((int)((G.tie ? 1 : 0) | (G.onp ? 2 : 0) | (gfailed ? 4 : 0)), 16 * myrow, 64);
    if (mine) {
        F.best_t = r_t;
        F.best_rec = r_rec;
        F.tie = (r_fl & 1) != 0;
        F.onp = (r_fl & 2) != 0;
        failed = (r_fl & 4) != 0;
    }
}

// The certified search for a wave of 16 rays, one per quad.  `alive` (quad-uniform): the quad carries a ray that passed the root
// gate and lies inside the search's envelope.  Must be called by all 64 lanes.  Returns (quad-uniform) true when (W.t, W.hit_rec)
// now hold the answer the MODE asks for; false: W is untouched, take the exact walk.  W, qlen: the same values in all four lanes.
template <bool COUNT, int MODE>
__device__ __forceinline__ bool walk_quad_wave(const SceneDev& S, const bool alive, Walk& W, const float qlen, uint32_t* __restrict__ wave_stk,
                                               LaneCounters& cnt) {
    const int lane = threadIdx.x & 63, q = lane & 3;
    uint32_t* __restrict__ stk = wave_stk + (lane >> 2) * CGRT_QSLOTS;
    const F3 o = W.o, d = W.d;
    FastScan F;
    F.best_t = (MODE == WALK_OCCLUDED) ? fminf(W.t, qlen) : W.t;
    F.best_rec = REF_NONE;
    F.tie = false;
    F.onp = false;
    uint32_t cur = alive ? S.fast_root : REF_NONE;
    int sp = 0;
    bool done = !alive, failed = false;
    auto POP = [&]() __attribute__((always_inline)) {
        if (!done && cur == REF_NONE) {
            if (sp > 0) {
                sp -= 1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the slot was written by another lane of the quad
                cur = stk[sp];
            } else {
                done = true;
            }
        }
    };
    for (;;) {
        const unsigned long long live = __ballot(!done);
        if (live == 0ull) break;
        if (CGRT_QUAD_WIDE_RAYS > 0 && __popcll(live) <= 4 * CGRT_QUAD_WIDE_RAYS) {
            quad_wide_tail<COUNT, MODE>(S, live & 0x1111111111111111ull, !done, W, qlen, F, cur, sp, failed, wave_stk, cnt);
            break;
        }
        if (!done && !(cur & REF_LEAF)) quad_node_step<COUNT>(S, W.P, F.best_t, q, cur, sp, stk, cnt);
        POP();
        if (!done && !(cur & REF_LEAF)) quad_node_step<COUNT>(S, W.P, F.best_t, q, cur, sp, stk, cnt);
        POP();
        if (!done && (cur & REF_LEAF)) {
            quad_run_step<COUNT, MODE>(S, cur, q, o, d, qlen, F, cnt);
            cur = REF_NONE;
            if (MODE != WALK_CLOSEST && (F.best_rec != REF_NONE || F.onp)) done = true;
        }
        POP();
    }
    if (!alive || failed || F.onp || (MODE == WALK_CLOSEST && F.tie)) return false;
    if (F.best_rec != REF_NONE) {
        const uint32_t cert0 = cnt.cert;
        const bool ok = path_certified<COUNT>(S, W, F.best_rec, F.best_t, cnt);
        if (COUNT && q != 0) cnt.cert = cert0;  // counted once per ray
        if (!ok) return false;
        W.t = F.best_t;
        W.hit_rec = F.best_rec;
    }
    return true;
}

// walk_tree (walk_fast.h) in the quad shape: all four lanes of a quad are called with the SAME ray; (t, hit_rec) are valid in
// every lane of the quad on return when the search was certified, in lane q == 0 always (the exact fallback runs there).
template <bool COUNT, int MODE = WALK_CLOSEST>
__device__ __forceinline__ void walk_tree_quad(const SceneDev& S, const bool active, const F3 o, const F3 d, float& t, uint32_t& hit_rec,
                                               uint32_t* __restrict__ wave_stk, LaneCounters& cnt, const float qlen = 0.0f) {
    const int q = threadIdx.x & 3;
    Walk W;
    W.o = o;
    W.d = d;
    W.t = t;
    W.hit_rec = REF_NONE;
    const bool entered = active && walk_begin(S, W);
    if (COUNT && entered && q == 0) cnt.entered++;
    const bool eligible = entered && W.P.regular;
    bool certified = false;
    if (__any(eligible)) certified = walk_quad_wave<COUNT, MODE>(S, eligible, W, qlen, wave_stk, cnt);
    if (COUNT && eligible && !certified && q == 0) cnt.fallback++;
    if (entered && !certified && q == 0) {  // one lane per ray walks exactly, on the lane-per-ray stack layout (the search is over)
        W.P = make_raypre(S, W.o, W.d, W.t);
        W.R = make_rayfast(S, W.o, W.d);
        walk_tree_unified<COUNT, MODE == WALK_ANYHIT>(S, W, wave_stk + (threadIdx.x & 63u), cnt);
    }
    t = W.t;
    hit_rec = W.hit_rec;
}

}  // namespace cgrt
