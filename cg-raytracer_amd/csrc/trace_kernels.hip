// trace_kernels.hip -- the shipped gfx950 kernels of the hot path: BoundingVolumeHierarchy::intersect
// (src/bounding_volume_hierarchy.cpp:850-881) for batches of rays, with optional fused primary-ray generation
// (src/main.cpp:691-694 + framework/src/trackball.cpp:92-103).  One ray per lane; the walk itself is in walk_exact.h
// (the reference's steps, one by one) and walk_fast.h (the certified walk: same answer from a better structure plus a
// proof, exact walk as fallback).  FAST selects the latter; the launchers take it when the scene carries a fast tree.
// Variants that are not shipped (persistent waves, per-wave stamps) live in variant_kernels.hip, the element-wise
// primitives of src/ray_tracing.h in prim_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "walk_quad.h"

#ifndef CGRT_LANE16_MAX_RAYS_DEFAULT
#define CGRT_LANE16_MAX_RAYS_DEFAULT 131072ull  // ray lists of at most this many rays: 16 rays per wave (see "kernel shape per launch")
#endif
#ifndef CGRT_QUAD4_MAX_RAYS_DEFAULT
#define CGRT_QUAD4_MAX_RAYS_DEFAULT 8192ull  // ... of at most this many: 4 rays per wave, 16 lanes per ray (walk_quad.h)
#endif

namespace cgrt {

// Quad shape (walk_quad.h): a launch of single-wave workgroups, FOUR per 8x8 tile.  Workgroup b serves tile-block
// tb = (b / 32) * 8 + b % 8 -- the workgroup index the lane-per-ray launch with 64 threads gives that tile, same blockIdx % 8
// residue, i.e. same XCD -- and of its 64 pixels the 16 with index (b / 8 % 4) * 16 + threadIdx / 4; the four lanes of a quad
// carry the same pixel.  Results land where the lane-per-ray launch puts them (packed or not).
template <bool QUAD>
__device__ __forceinline__ bool frame_pixel(const FrameDev& F, int& x, int& y, size_t& packed_index, bool& writer) {
    if (QUAD) {
        const uint32_t b = blockIdx.x, tb = ((b >> 5) << 3) | (b & 7u);
        const uint32_t p = ((b >> 3) & 3u) * 16u + (threadIdx.x >> 2);
        packed_index = (size_t)tb * 64u + p;
        writer = (threadIdx.x & 3u) == 0u;
        return tile_pixel_of(F, tb, p, x, y);
    }
    packed_index = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    writer = true;
    return tile_pixel_of(F, blockIdx.x, threadIdx.x, x, y);
}

// HINT: the launch carries frame hints (cgrt_layout.h HintDev) -- an instantiation of its own, so that the plain frame kernel is the
// kernel it was (the hint code costs every launch a few percent when it is merely compiled in: measured).
template <bool COUNT, bool FAST, bool QUAD = false, bool HINT = false>
__global__ CGRT_LB void k_trace_primary(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits, float* __restrict__ normals,
                                        unsigned long long* counters) {
    extern __shared__ uint32_t s_lds[];  // CGRT_LDS_WORDS(blockDim.x): stacks, quad-tail owner maps, workgroup scratch
    int x = 0, y = 0;
    size_t pidx;
    bool writer;
    bool active;
    if (HINT && F.hint) {  // (the run-time test is redundant; with it the compiler keeps the walk out of scratch)
        pidx = 0;          // (hinted frames are never packed)
        writer = true;
        active = hinted_tile_pixel(F, x, y, CGRT_HINT_SCRATCH(s_lds));
    } else {
        if (HINT && (threadIdx.x & 63u) == 0u) CGRT_HINT_SCRATCH(s_lds)[1] = 0xffffffffu;  // nothing for hint_finish
        active = frame_pixel<QUAD>(F, x, y, pidx, writer);
    }
    LaneCounters cnt;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    if (active) primary_ray(C, F.W, F.H, x, y, o, d);
    float t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
    uint32_t hit_rec = REF_NONE;
    if (QUAD)
        walk_tree_quad<COUNT>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), cnt);
    else
        walk_tree<COUNT, FAST>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
    if (active && writer) {
        const size_t pix = F.packed ? pidx : (size_t)y * F.W + x;
        finish_ray(S, o, d, t, hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
    }
    if (COUNT) flush_counters(cnt, active && writer, counters);
    if (HINT) hint_finish(CGRT_HINT_SCRATCH(s_lds));
}

// Primary frame for the shading wavefront (cgrt_render): the fused kernel's walk, but only the rays that HIT are written,
// appended to a compact list {ray, hit, normal, pixel} (one atomic per workgroup, lanes ranked by ballot; workgroups
// finish roughly in launch order, so the list keeps the frame's tile order).  Pixels that miss need no further work upstream
// either (main.cpp:293: black).  count = one zeroed device word.
template <bool COUNT, bool FAST, bool QUAD = false>
__global__ CGRT_LB void k_trace_primary_compact(SceneDev S, CameraDev C, FrameDev F, float* __restrict__ rays, CgrtHitDev* __restrict__ hits,
                                                float* __restrict__ normals, int* __restrict__ pixels, uint32_t* __restrict__ count,
                                                unsigned long long* counters, float* __restrict__ rgb, const SpawnDev* __restrict__ spawn_dev) {
    extern __shared__ uint32_t s_lds[];  // CGRT_LDS_WORDS(blockDim.x): stacks, quad-tail owner maps, workgroup scratch
    const int lane = threadIdx.x & 63;
    int x = 0, y = 0;
    size_t pidx;
    bool writer;
    const bool active = frame_pixel<QUAD>(F, x, y, pidx, writer);
    if (active && writer && rgb) {  // every pixel this rank owns starts black (main.cpp:293); the hits are written over it at the end of the frame
        float* p = rgb + 3ull * ((unsigned long long)y * F.W + x);
        p[0] = p[1] = p[2] = 0.0f;
    }
    LaneCounters cnt;
    CgrtHitDev h;
    h.hit = 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0), nn = f3(0, 0, 0);
    if (active) primary_ray(C, F.W, F.H, x, y, o, d);
    float t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
    uint32_t hit_rec = REF_NONE;
    if (QUAD)
        walk_tree_quad<COUNT>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), cnt);
    else
        walk_tree<COUNT, FAST>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
    if (active && writer) resolve_hit(S, o, d, t, hit_rec, true, h, nn);
    if (COUNT) flush_counters(cnt, active && writer, counters);
    // one atomic per workgroup (same-address atomics serialise at the L2); the workgroup's LDS is only released when its
    // last wave ends anyway, so waiting for it here costs no occupancy
    const bool keep = active && writer && h.hit != 0;
    // Fused level-0 spawn (a predicted frame, capi.cpp render_impl; k_spawn's work from this lane's registers, spawn_rays.h): does
    // the hit want a mirror ray?  (The parameters are read from device memory HERE: as kernel arguments they were loaded before
    // the walk and their 22 scalar registers pushed the walk into scratch.)
    F3 ks = f3(0.f, 0.f, 0.f);
    bool wants_mirror = false;
    if (spawn_dev && keep) {
        const int mid = h.material_id;
        const float* mats = spawn_dev->materials;
        if (mid >= 0) ks = f3(mats[8 * mid + 3], mats[8 * mid + 4], mats[8 * mid + 5]);
        wants_mirror = !(ks.z <= 0.01f) && spawn_dev->spawn;  // (:246 tests ks.z only; spawn = level + 1 < maxLevel, :267)
    }
    const unsigned long long m = __ballot(keep), mm = __ballot(wants_mirror);
    uint32_t* s_cnt = CGRT_BLOCK_SCRATCH(s_lds);
    const unsigned w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(m) | ((uint32_t)__popcll(mm) << 16);  // (a workgroup has at most 1024 lanes)
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, mtot = 0;
        for (unsigned k = 0; k < nw; k++) {
            const uint32_t c = s_cnt[k];
            s_cnt[k] = tot | (mtot << 16);
            tot += c & 0xffffu;
            mtot += c >> 16;
        }
        if (!tot) {
            s_cnt[nw] = 0u;
        } else if (spawn_dev) {
            // count[0] = hits, count[1] = mirror rays, ONE 64-bit atomic for both: three more counters of their own (the shadow rays and
            // hits of k_spawn's counter block, the mirror rays) quadrupled the same-line atomics and the Cornell frame's primary
            // kernel went from 70 to 174 us.  (The level's shadow rays are hits x lights: no counter.)
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(count), (unsigned long long)tot | ((unsigned long long)mtot << 32));
            s_cnt[nw] = (uint32_t)old;
            s_lds[0] = (uint32_t)(old >> 32);  // (wave 0's stack: its walk is over)
        } else {
            s_cnt[nw] = atomicAdd(count, tot);
        }
    }
    __syncthreads();
    if (keep) {
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long idx = s_cnt[nw] + (s_cnt[w] & 0xffffu) + (uint32_t)__popcll(m & below);
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
        int px = 0, py = 0;  // (the pixel is rebuilt rather than kept in registers through the walk)
        size_t pi;
        bool wr;
        (void)frame_pixel<QUAD>(F, px, py, pi, wr);
        pixels[idx] = py * F.W + px;
        if (spawn_dev) {
            // Entry idx's shadow ray towards light l is shadow ray idx * nlights + l (every entry of level 0 is a hit: no append).
            const SpawnDev SP = *spawn_dev;
            const F3 pointOn = add(o, scale(d, h.t));
            for (unsigned l = 0; l < SP.nlights; l++) {
                const unsigned long long q = idx * SP.nlights + l;
                SP.sslot[q] = (int)q;
                spawn_shadow_ray(SP.lights, l, pointOn, q, SP.srays, SP.sdist);
            }
            const uint32_t child = s_lds[0] + (s_cnt[w] >> 16) + (uint32_t)__popcll(mm & below);
            if (wants_mirror) {
                spawn_mirror_ray(pointOn, d, nn, child, SP.next_rays);
                SP.next_pixels[child] = py * F.W + px;
            }
            SP.lvl[2 * idx + 1] = make_float4(ks.x, ks.y, ks.z, __int_as_float(wants_mirror ? (int)child : -1));
        }
    }
}
// Two lists in ONE launch: a level's shadow list (occlusion queries, k_trace_shadow's body) and its mirror list (closest hits,
// k_trace_batch's body), workgroups dealt alternately so that both are on the chip at once.  A predicted frame (capi.cpp
// render_impl) used to run them on two streams; the event that started the second stream and the wait that joined it again cost
// ~17 us of an idle GPU per frame (profiles/r3_config3.txt) -- one launch on one stream needs neither.  Lane shapes only (64 or 16
// rays per single-wave workgroup, by the host's estimate or on the device: see "kernel shape per launch").
struct ListPairDev {
    const float* rays_a;  // the shadow list
    const float* dist_a;
    CgrtHitDev* hits_a;
    const uint32_t* dcount_a;
    unsigned long long n_a;
    unsigned dmul_a, rpw_a, adapt_a, blocks_a;
    const float* rays_b;  // the mirror list
    CgrtHitDev* hits_b;
    float* normals_b;
    const uint32_t* dcount_b;
    unsigned long long n_b;
    unsigned rpw_b, adapt_b, blocks_b;
};
template <bool FAST>
__global__ CGRT_LB void k_trace_pair(SceneDev S, ListPairDev P) {
    extern __shared__ uint32_t s_lds[];
    const unsigned b = blockIdx.x, both = P.blocks_a < P.blocks_b ? P.blocks_a : P.blocks_b;
    const bool is_a = b < 2u * both ? (b & 1u) == 0u : P.blocks_a > P.blocks_b;
    const unsigned bid = b < 2u * both ? (b >> 1) : b - both;  // the workgroup's index within its list's launch
    LaneCounters cnt;
    if (is_a) {
        unsigned long long n = P.n_a;
        if (P.dcount_a) {
            const unsigned long long present = (unsigned long long)*P.dcount_a * P.dmul_a;
            n = present < n ? present : n;
        }
        unsigned rpw = P.rpw_a;
        if (P.adapt_a) rpw = (n <= P.adapt_a) ? 16u : 64u;
        const unsigned long long i = (unsigned long long)bid * rpw + threadIdx.x;
        const bool active = i < n && threadIdx.x < rpw;
        F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
        float t = 0.0f, qlen = 0.0f;
        if (active) {
            const float* r = P.rays_a + 7 * i;
            o = f3(r[0], r[1], r[2]);
            d = f3(r[3], r[4], r[5]);
            t = r[6];
            qlen = P.dist_a[i];
        }
        uint32_t hit_rec = REF_NONE;
        walk_tree<false, FAST, WALK_OCCLUDED>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt, qlen);
        if (active) finish_ray(S, o, d, t, hit_rec, P.hits_a + i, nullptr);
    } else {
        unsigned long long n = P.n_b;
        if (P.dcount_b) {
            const unsigned long long present = *P.dcount_b;
            n = present < n ? present : n;
        }
        unsigned rpw = P.rpw_b;
        if (P.adapt_b) rpw = (n <= P.adapt_b) ? 16u : 64u;
        const unsigned long long i = (unsigned long long)bid * rpw + threadIdx.x;
        const bool active = i < n && threadIdx.x < rpw;
        F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
        float t = 0.0f;
        if (active) {
            const float* r = P.rays_b + 7 * i;
            o = f3(r[0], r[1], r[2]);
            d = f3(r[3], r[4], r[5]);
            t = r[6];
        }
        uint32_t hit_rec = REF_NONE;
        walk_tree<false, FAST>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
        if (active) finish_ray(S, o, d, t, hit_rec, P.hits_b + i, P.normals_b ? P.normals_b + 3 * i : nullptr);
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
template <bool COUNT, bool FAST, bool QUAD = false>
__global__ CGRT_LB void k_trace_batch(SceneDev S, const float* __restrict__ rays, unsigned long long n,
                                                            CgrtHitDev* __restrict__ hits, float* __restrict__ normals,
                                                            unsigned long long* counters, const uint32_t* __restrict__ dcount, unsigned qrpw, unsigned adapt_max) {
    extern __shared__ uint32_t s_lds[];  // CGRT_LDS_WORDS(blockDim.x): stacks, quad-tail owner maps, workgroup scratch
    if (dcount) {
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    // quad shape: the four lanes of a quad carry the same ray; a wave carries qrpw rays (16, or 4: the wide tail from the start)
    // lane shape: one ray per lane; a SPARSE launch (single-wave workgroups with qrpw < 64 rays each) leaves the upper lanes of every
    // wave empty, so that a wave of <= 16 rays runs the quad tail (four lanes per ray, four stack entries in flight) from its
    // first step.  adapt_max (lists whose length only the device knows): sparse iff that length is at most adapt_max -- the grid
    // covers both layouts.
    if (!QUAD && adapt_max) qrpw = (n <= adapt_max) ? 16u : 64u;
    const bool sparse = !QUAD && qrpw < 64u;
    const unsigned long long i = QUAD ? ((unsigned long long)blockIdx.x * qrpw + (threadIdx.x >> 2)) : (sparse ? (unsigned long long)blockIdx.x * qrpw + threadIdx.x : g);
    const bool writer = !QUAD || (threadIdx.x & 3u) == 0u;
    const bool active = i < n && (!QUAD || (threadIdx.x >> 2) < qrpw) && (!sparse || threadIdx.x < qrpw);
    LaneCounters cnt;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    float t = 0.0f;
    if (active) {
        const float* r = rays + 7 * i;
        o = f3(r[0], r[1], r[2]);
        d = f3(r[3], r[4], r[5]);
        t = r[6];
    }
    uint32_t hit_rec = REF_NONE;
    if (QUAD)
        walk_tree_quad<COUNT>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), cnt);
    else
        walk_tree<COUNT, FAST>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
    if (active && writer) {
        // (the ray's index is rebuilt rather than kept in two registers through the walk)
        const unsigned long long k = QUAD ? ((unsigned long long)blockIdx.x * qrpw + (threadIdx.x >> 2))
                                          : (sparse ? (unsigned long long)blockIdx.x * qrpw + threadIdx.x : (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x);
        finish_ray(S, o, d, t, hit_rec, hits + k, normals ? normals + 3 * k : nullptr);
    }
    if (COUNT) flush_counters(cnt, active && writer, counters);
}

// pointInShadow's rays (main.cpp:104-135) for the shading wavefront: dist[i] = |fromPosToLight| of ray i.  The caller only
// evaluates `hit && !(t + epsilon >= dist)`: the certified walk answers that question directly (WALK_OCCLUDED: bounded by
// the light's distance, stops at the first qualifying triangle); a ray without certificate gets the exact closest hit.
// hits[i] therefore holds a hit that decides the test like the reference's own, not necessarily the closest one.
template <bool COUNT, bool FAST, bool QUAD = false>
__global__ CGRT_LB void k_trace_shadow(SceneDev S, const float* __restrict__ rays, const float* __restrict__ dist, unsigned long long n,
                                       CgrtHitDev* __restrict__ hits, const uint32_t* __restrict__ dcount, unsigned long long* counters, unsigned qrpw, unsigned adapt_max,
                                       unsigned dmul) {
    extern __shared__ uint32_t s_lds[];
    if (dcount) {  // dmul: *dcount counts the entries whose shadow rays these are, dmul rays (lights) each
        const unsigned long long present = (unsigned long long)*dcount * dmul;
        n = present < n ? present : n;
    }
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!QUAD && adapt_max) qrpw = (n <= adapt_max) ? 16u : 64u;  // (see k_trace_batch)
    const bool sparse = !QUAD && qrpw < 64u;
    const unsigned long long i = QUAD ? ((unsigned long long)blockIdx.x * qrpw + (threadIdx.x >> 2)) : (sparse ? (unsigned long long)blockIdx.x * qrpw + threadIdx.x : g);
    const bool writer = !QUAD || (threadIdx.x & 3u) == 0u;
    const bool active = i < n && (!QUAD || (threadIdx.x >> 2) < qrpw) && (!sparse || threadIdx.x < qrpw);
    LaneCounters cnt;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    float t = 0.0f, qlen = 0.0f;
    if (active) {
        const float* r = rays + 7 * i;
        o = f3(r[0], r[1], r[2]);
        d = f3(r[3], r[4], r[5]);
        t = r[6];
        qlen = dist[i];
    }
    uint32_t hit_rec = REF_NONE;
    if (QUAD)
        walk_tree_quad<COUNT, WALK_OCCLUDED>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), cnt, qlen);
    else
        walk_tree<COUNT, FAST, WALK_OCCLUDED>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt, qlen);
    if (active && writer) finish_ray(S, o, d, t, hit_rec, hits + i, nullptr);
    if (COUNT) flush_counters(cnt, active && writer, counters);
}

// Soft shadows of spherical lights (main.cpp:168-218): `samples` shadow rays per (hit item, light), generated in
// registers from the item's ray + hit and the unit-vector table (no ray buffer: 200 samples x 2 M hits would be 12 GB),
// traversed, and counted: lit[item * nlights + l] = number of samples with !intersect || ray.t > lightT (:183-199).
// Thread g = (item * nlights + l) * samples + smp: a wave's rays leave one or two surface points towards one small
// sphere, the most coherent batch this library sees.  ANYHIT stops a ray at its first accepting leaf: the count needs
// the hit flag only (an accepted t is below lightT by construction, or 0 from the on-plane rule, never above).
template <bool ANYHIT, bool FAST>
__global__ CGRT_LB void k_soft_shadow(SceneDev S, SoftDev Q, const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits,
                                      const int* __restrict__ item_pixels, unsigned long long nthreads, uint32_t* __restrict__ lit) {
    extern __shared__ uint32_t s_lds[];  // CGRT_LDS_WORDS(blockDim.x): stacks, quad-tail owner maps, workgroup scratch
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = g < nthreads;
    const unsigned long long key = in ? g / Q.samples : 0ull;  // item * nlights + l
    const uint32_t smp = in ? (uint32_t)(g - key * Q.samples) : 0u;
    const unsigned long long item = key / Q.nlights;
    const uint32_t l = (uint32_t)(key - item * Q.nlights);
    bool is_lit = false;
    const bool live = in && hits[item].hit != 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    float t = 0.0f;
    if (live) {
        const float* r = rays + 7 * item;
        const F3 pointOn = add(f3(r[0], r[1], r[2]), scale(f3(r[3], r[4], r[5]), hits[item].t));
        const float* L = Q.lights + 7 * l;
        const float* u = Q.units + 3ull * soft_sample_index(Q.seed, (uint32_t)item_pixels[item], Q.level, l, smp, Q.nunits);
        soft_shadow_ray(pointOn, f3(L[0], L[1], L[2]), L[3], f3(u[0], u[1], u[2]), o, d, t);
    }
    const float lightT = t;
    uint32_t hit_rec = REF_NONE;
    LaneCounters cnt;
    walk_tree<false, FAST, ANYHIT ? WALK_ANYHIT : WALK_CLOSEST>(S, live, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
    if (live) {
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
// ---------------------------------------------------------------------------------------------
// launchers (host).  A scene with a fast tree (SceneDev::fast_root) takes the FAST instantiations; the C-ABI clears
// fast_root in its copy of SceneDev to force the exact walk (cgrt_scene_set_walk).  Workgroup size: FrameDev::block for
// frames, trace_block() for ray lists -- 64 threads (one wave, its LDS released the moment it ends) for the certified walk,
// whose waves differ widely in length; 256 for the exact walk (measured: profiles/r2_exp_block_size.txt).
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(unsigned long long n, unsigned block) { return (unsigned)((n + block - 1) / block); }
static inline size_t lds_bytes(unsigned block) { return sizeof(uint32_t) * (size_t)CGRT_LDS_WORDS(block); }
int trace_block(const SceneDev& S) {
    static const int forced = [] {
        const char* e = getenv("CGRT_BLOCK_THREADS");  // experiment knob: 64 / 128 / 256
        const int v = e ? atoi(e) : 0;
        return (v == 64 || v == 128 || v == 256) ? v : 0;
    }();
    if (forced) return forced;
    return S.fast_root != REF_NONE ? 64 : 256;
}
#define CGRT_LAUNCH2(KERNEL, A, fast, grid, block, stream, ...)                                                             \
    do {                                                                                                                      \
        if (fast)                                                                                                             \
            hipLaunchKernelGGL((KERNEL<A, true>), dim3(grid), dim3(block), lds_bytes(block), stream, __VA_ARGS__);           \
        else                                                                                                                  \
            hipLaunchKernelGGL((KERNEL<A, false>), dim3(grid), dim3(block), lds_bytes(block), stream, __VA_ARGS__);          \
    } while (0)
// the quad shape (walk_quad.h): single-wave workgroups, four lanes per ray
#define CGRT_LAUNCHQ(KERNEL, A, grid, stream, ...) \
    hipLaunchKernelGGL((KERNEL<A, true, true>), dim3(grid), dim3(64), lds_bytes(64), stream, __VA_ARGS__)

// ---- kernel shape per launch ----
// How a launch lays its rays out on the lanes (results never depend on it; tests/test_quad_shape_gpu.py):
//   LANE64  one ray per lane, 64 per wave: the throughput shape (walk_fast.h), every frame and every large list;
//   LANE16  one ray per lane, 16 rays per single-wave workgroup: the wave is in the quad tail -- four lanes per ray, four stack
//           entries in flight -- from its first step, and a hard ray shares its wave with 15 others instead of 63;
//   QUAD4   four rays per wave, a row of 16 lanes each: four stack entries x four child boxes per round (walk_quad.h);
//   QUAD16  quad per ray, 16 rays per wave (walk_quad.h's basic shape; never chosen by size, kept selectable).
// A launch ends when its hardest rays end, and a hard ray is a dependent chain that gets ~2x faster when it does not share its
// wave's bodies with 63 others (profiles/r3_exp_list_shapes.txt: 64 hard rays 97 us as one wave, 51 us as four, 45 us as
// sixteen) -- but sparse waves cost throughput, so the shape goes by the number of rays: <= g_quad4_max -> QUAD4,
// <= g_lane16_max -> LANE16, else LANE64 (crossovers measured: 4 K rays 75 / 49 / 46 us, 16 K 84 / 56 / 60, 64 K 94 / 79 / 136,
// 255 K 126 / 142 / 372 for LANE64 / LANE16 / QUAD4).  Lists whose length only the device knows (cgrt_render's wavefront) choose
// between LANE64 and LANE16 on the device (adapt_max).  Frames keep LANE64: 71 % of a frame's waves die at the root gate and a
// sparse shape pays its per-wave set-up four times for them (share of a 4K frame: 94 us -> 151 us, profiles/r3_exp_quad_shape.txt).
enum { SHAPE_AUTO = -1, SHAPE_LANE64 = 0, SHAPE_QUAD16 = 1, SHAPE_LANE16 = 2, SHAPE_QUAD4 = 3 };
static std::atomic<int> g_shape_mode{SHAPE_AUTO};
static std::atomic<unsigned long long> g_lane16_max{CGRT_LANE16_MAX_RAYS_DEFAULT};
static std::atomic<unsigned long long> g_quad4_max{CGRT_QUAD4_MAX_RAYS_DEFAULT};
void set_quad_shape(int mode, unsigned long long max_rays) {
    g_shape_mode.store((mode < 0 || mode > 3) ? SHAPE_AUTO : mode);
    if (max_rays) {
        g_lane16_max.store(max_rays);
        g_quad4_max.store(std::min<unsigned long long>(max_rays, CGRT_QUAD4_MAX_RAYS_DEFAULT));
    }
}
void get_quad_shape(int* mode, unsigned long long* max_rays) {
    *mode = g_shape_mode.load();
    *max_rays = g_lane16_max.load();
}
static int shape_mode() {
    static const int env_mode = [] {
        const char* e = getenv("CGRT_SHAPE");  // experiment knob: -1 auto / 0 lane64 / 1 quad16 / 2 lane16 / 3 quad4
        return e ? atoi(e) : -2;
    }();
    return env_mode != -2 ? env_mode : g_shape_mode.load();
}
static unsigned long long env_ull(const char* name) {
    const char* e = getenv(name);
    return e ? strtoull(e, nullptr, 10) : 0ull;
}
// shape of a ray LIST of n rays (n known to the host)
static int list_shape(const SceneDev& S, unsigned long long n) {
    if (S.fast_root == REF_NONE || trace_block(S) != 64) return SHAPE_LANE64;
    const int mode = shape_mode();
    if (mode != SHAPE_AUTO) return mode;
    static const unsigned long long e16 = env_ull("CGRT_LANE16_MAX"), e4 = env_ull("CGRT_QUAD4_MAX");
    if (n <= (e4 ? e4 : g_quad4_max.load())) return SHAPE_QUAD4;
    if (n <= (e16 ? e16 : g_lane16_max.load())) return SHAPE_LANE16;
    return SHAPE_LANE64;
}
// lists sized on the device: 0 = the host's choice stands, else "LANE16 iff the device count is at most this"
static unsigned list_adapt_max(const SceneDev& S, const uint32_t* dcount) {
    if (!dcount || S.fast_root == REF_NONE || trace_block(S) != 64 || shape_mode() != SHAPE_AUTO) return 0u;
    static const unsigned long long e16 = env_ull("CGRT_LANE16_MAX");
    return (unsigned)std::min<unsigned long long>(e16 ? e16 : g_lane16_max.load(), 0x7fffffffull);
}
bool quad_shape_for(const SceneDev& S, unsigned long long rays) {  // frames: the quad shape only when forced
    (void)rays;
    return S.fast_root != REF_NONE && trace_block(S) == 64 && shape_mode() == SHAPE_QUAD16;
}

hipError_t launch_trace_primary(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                unsigned long long* counters, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    const bool fast = S.fast_root != REF_NONE;
    if (F.block == 64 && quad_shape_for(S, (unsigned long long)F.nblocks * 64ull)) {
        if (counters)
            CGRT_LAUNCHQ(k_trace_primary, true, 4u * F.nblocks, stream, S, C, F, hits, normals, counters);
        else
            CGRT_LAUNCHQ(k_trace_primary, false, 4u * F.nblocks, stream, S, C, F, hits, normals, counters);
        return hipGetLastError();
    }
    if (F.hint && fast && F.block == 64 && !counters) {  // frame hints: the hard list's workgroups come first
        hipLaunchKernelGGL((k_trace_primary<false, true, false, true>), dim3(F.nblocks + F.hint_blocks), dim3(64), lds_bytes(64), stream, S, C, F, hits,
                           normals, counters);
        return hipGetLastError();
    }
    if (counters)
        CGRT_LAUNCH2(k_trace_primary, true, fast, F.nblocks, (unsigned)F.block, stream, S, C, F, hits, normals, counters);
    else
        CGRT_LAUNCH2(k_trace_primary, false, fast, F.nblocks, (unsigned)F.block, stream, S, C, F, hits, normals, counters);
    return hipGetLastError();
}
// grid of a list launch in the lane shapes: covers 64 rays per workgroup, and 16 per workgroup up to the adaptive bound
static unsigned lane_grid(unsigned long long n, unsigned block, unsigned rpw, unsigned adapt_max) {
    if (adapt_max) return std::max(grid_for(n, block), grid_for(std::min<unsigned long long>(n, adapt_max), 16));
    return grid_for(n, rpw < 64u ? rpw : block);
}
hipError_t launch_trace_batch(const SceneDev& S, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                              unsigned long long* counters, hipStream_t stream, const uint32_t* dcount, unsigned long long expected) {
    if (n == 0) return hipSuccess;
    const unsigned block = (unsigned)trace_block(S);
    const bool fast = S.fast_root != REF_NONE;
    // expected (with dcount): the host's estimate of the device count picks the shape, as an exact length would
    const unsigned adapt = expected ? 0u : list_adapt_max(S, dcount);
    const int shape = adapt ? SHAPE_LANE64 : list_shape(S, expected ? expected : n);
    if (shape == SHAPE_QUAD16 || shape == SHAPE_QUAD4) {
        const unsigned q = shape == SHAPE_QUAD4 ? 4u : 16u;
        if (counters)
            CGRT_LAUNCHQ(k_trace_batch, true, grid_for(n, q), stream, S, rays, n, hits, normals, counters, dcount, q, 0u);
        else
            CGRT_LAUNCHQ(k_trace_batch, false, grid_for(n, q), stream, S, rays, n, hits, normals, counters, dcount, q, 0u);
        return hipGetLastError();
    }
    const unsigned rpw = shape == SHAPE_LANE16 ? 16u : 64u;
    const unsigned grid = lane_grid(n, block, rpw, adapt);
    if (counters)
        CGRT_LAUNCH2(k_trace_batch, true, fast, grid, block, stream, S, rays, n, hits, normals, counters, dcount, rpw, adapt);
    else
        CGRT_LAUNCH2(k_trace_batch, false, fast, grid, block, stream, S, rays, n, hits, normals, counters, dcount, rpw, adapt);
    return hipGetLastError();
}
hipError_t launch_trace_shadow(const SceneDev& S, const float* rays, const float* dist, unsigned long long n, CgrtHitDev* hits, hipStream_t stream,
                               const uint32_t* dcount, unsigned long long* counters, unsigned long long expected, unsigned dmul) {
    if (n == 0) return hipSuccess;
    const unsigned block = (unsigned)trace_block(S);
    const bool fast = S.fast_root != REF_NONE;
    const unsigned adapt = expected ? 0u : list_adapt_max(S, dcount);
    const int shape = adapt ? SHAPE_LANE64 : list_shape(S, expected ? expected : n);
    if (shape == SHAPE_QUAD16 || shape == SHAPE_QUAD4) {
        const unsigned q = shape == SHAPE_QUAD4 ? 4u : 16u;
        if (counters)
            CGRT_LAUNCHQ(k_trace_shadow, true, grid_for(n, q), stream, S, rays, dist, n, hits, dcount, counters, q, 0u, dmul ? dmul : 1u);
        else
            CGRT_LAUNCHQ(k_trace_shadow, false, grid_for(n, q), stream, S, rays, dist, n, hits, dcount, counters, q, 0u, dmul ? dmul : 1u);
        return hipGetLastError();
    }
    const unsigned rpw = shape == SHAPE_LANE16 ? 16u : 64u;
    const unsigned grid = lane_grid(n, block, rpw, adapt);
    if (counters)
        CGRT_LAUNCH2(k_trace_shadow, true, fast, grid, block, stream, S, rays, dist, n, hits, dcount, counters, rpw, adapt, dmul ? dmul : 1u);
    else
        CGRT_LAUNCH2(k_trace_shadow, false, fast, grid, block, stream, S, rays, dist, n, hits, dcount, counters, rpw, adapt, dmul ? dmul : 1u);
    return hipGetLastError();
}
// The two lists of one level in one launch (k_trace_pair); false when the scene or the forced shape does not allow it -- the caller
// then launches them one by one.
bool can_trace_pair(const SceneDev& S) { return S.fast_root != REF_NONE && trace_block(S) == 64 && (shape_mode() == SHAPE_AUTO || shape_mode() == SHAPE_LANE64 || shape_mode() == SHAPE_LANE16); }
hipError_t launch_trace_pair(const SceneDev& S, const float* srays, const float* sdist, unsigned long long ns, CgrtHitDev* shits, const uint32_t* sdcount,
                             unsigned sdmul, unsigned long long sexpected, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                             const uint32_t* dcount, unsigned long long expected, hipStream_t stream) {
    if (ns == 0 && n == 0) return hipSuccess;
    auto lane_shape = [&](unsigned long long cap, unsigned long long exp_, const uint32_t* dc, unsigned& rpw, unsigned& adapt, unsigned& blocks) {
        adapt = exp_ ? 0u : list_adapt_max(S, dc);
        const int shape = adapt ? SHAPE_LANE64 : list_shape(S, exp_ ? exp_ : cap);
        rpw = (shape == SHAPE_LANE64) ? 64u : 16u;  // (the quad shapes of a short list: 16 rays per wave here)
        blocks = cap ? lane_grid(cap, 64u, rpw, adapt) : 0u;
        if (adapt) rpw = 64u;
    };
    ListPairDev P{};
    P.rays_a = srays, P.dist_a = sdist, P.hits_a = shits, P.dcount_a = sdcount, P.n_a = ns, P.dmul_a = sdmul ? sdmul : 1u;
    lane_shape(ns, sexpected, sdcount, P.rpw_a, P.adapt_a, P.blocks_a);
    P.rays_b = rays, P.hits_b = hits, P.normals_b = normals, P.dcount_b = dcount, P.n_b = n;
    lane_shape(n, expected, dcount, P.rpw_b, P.adapt_b, P.blocks_b);
    hipLaunchKernelGGL((k_trace_pair<true>), dim3(P.blocks_a + P.blocks_b), dim3(64), lds_bytes(64), stream, S, P);
    return hipGetLastError();
}
hipError_t launch_trace_primary_compact(const SceneDev& S, const CameraDev& C, const FrameDev& F, float* rays, CgrtHitDev* hits, float* normals,
                                        int* pixels, uint32_t* count, hipStream_t stream, unsigned long long* counters, float* rgb,
                                        const SpawnDev* spawn) {
    const SpawnDev* SP = spawn;  // (a DEVICE address)
    if (F.nblocks == 0) return hipSuccess;
    const unsigned block = (unsigned)F.block;
    const bool fast = S.fast_root != REF_NONE;
    if (F.block == 64 && quad_shape_for(S, (unsigned long long)F.nblocks * 64ull)) {
        if (counters)
            CGRT_LAUNCHQ(k_trace_primary_compact, true, 4u * F.nblocks, stream, S, C, F, rays, hits, normals, pixels, count, counters, rgb, SP);
        else
            CGRT_LAUNCHQ(k_trace_primary_compact, false, 4u * F.nblocks, stream, S, C, F, rays, hits, normals, pixels, count, counters, rgb, SP);
        return hipGetLastError();
    }
    if (counters)
        CGRT_LAUNCH2(k_trace_primary_compact, true, fast, F.nblocks, block, stream, S, C, F, rays, hits, normals, pixels, count, counters, rgb, SP);
    else
        CGRT_LAUNCH2(k_trace_primary_compact, false, fast, F.nblocks, block, stream, S, C, F, rays, hits, normals, pixels, count, counters, rgb, SP);
    return hipGetLastError();
}
hipError_t launch_clear_owned(const FrameDev& F, float* rgb, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_clear_owned, dim3(F.nblocks), dim3((unsigned)F.block), 0, stream, F, rgb);
    return hipGetLastError();
}
hipError_t launch_soft_shadow(const SceneDev& S, const SoftDev& Q, const float* rays, const CgrtHitDev* hits, const int* item_pixels,
                              unsigned long long nitems, uint32_t* lit, int anyhit, hipStream_t stream) {
    const unsigned long long nthreads = nitems * Q.nlights * Q.samples;
    if (nthreads == 0) return hipSuccess;
    const unsigned block = (unsigned)trace_block(S);
    const unsigned long long blocks = (nthreads + block - 1) / block;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const bool fast = S.fast_root != REF_NONE;
    if (anyhit)
        CGRT_LAUNCH2(k_soft_shadow, true, fast, (unsigned)blocks, block, stream, S, Q, rays, hits, item_pixels, nthreads, lit);
    else
        CGRT_LAUNCH2(k_soft_shadow, false, fast, (unsigned)blocks, block, stream, S, Q, rays, hits, item_pixels, nthreads, lit);
    return hipGetLastError();
}
// One word into host-visible memory, behind whatever the stream holds: lets a host thread wait for a launch by looking at its own
// memory instead of sleeping in a runtime call (capi.cpp combined_intersect).
__global__ void k_signal(uint32_t* __restrict__ flag, uint32_t value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_signal(uint32_t* flag, uint32_t value, hipStream_t stream) {
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, stream, flag, value);
    return hipGetLastError();
}
hipError_t launch_generate_rays(const CameraDev& C, int W, int H, int x0, int y0, int x1, int y1, float* rays, hipStream_t stream) {
    const unsigned long long n = (unsigned long long)(x1 - x0) * (unsigned long long)(y1 - y0);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_rays, dim3(grid_for(n, CGRT_BLOCK)), dim3(CGRT_BLOCK), 0, stream, C, W, H, x0, y0, x1, y1, rays);
    return hipGetLastError();
}

}  // namespace cgrt
