// shade_kernels.hip -- the reference's shading / recursion driver (src/main.cpp:61-310) on the device, as a wavefront:
// per recursion level one kernel spawns the shadow rays of every hit (pointInShadow, :104-135), the batch traversal
// kernel answers them, one kernel evaluates the Phong terms (:61-98, :219-232) and spawns the mirror rays (shade,
// :241-264), and after the last level one kernel folds the levels back (color = direct + reflected * ks, :262).
// Every level is a COMPACT list of live paths: level 0 is the frame in the primary kernel's order (tiles, super-tiles),
// a hit appends its shadow rays to the level's shadow list and its mirror ray to the next level's list (one wave-level
// atomic per append site, lanes ranked by ballot), so that the traversal kernel only ever sees real rays, still roughly
// in tile order.  Links: an entry's level record holds the index of its child entry on the next level; sslot[entry * L + l]
// is the index of its shadow ray towards light l.  Spherical lights (:168-218) are sampled by k_soft_shadow (trace_kernels.hip), which
// leaves the number of unoccluded samples per (item, light); the draws of randomUnitVector() are a caller-supplied table.
// Arithmetic follows the reference's expression order (cgrt_math.h); pow(float, float) is powf (device libm: the last
// ulp may differ from glibc's -- the RGB parity bar is 1e-5 absolute).
#include <hip/hip_runtime.h>

#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "spawn_rays.h"
#include "trace_kernels.h"

namespace cgrt {

__device__ __forceinline__ F3 ldv(const float* p) { return f3(p[0], p[1], p[2]); }

// Block-aggregated append: threads with `want` get consecutive indices of the list behind `counter`, ONE atomic per
// workgroup (same-address atomics serialise at the L2, ~12 ns each: one per wave made these streaming kernels
// atomic-bound).  Every thread of the block must call it; s_tmp = blockDim.x / 64 + 1 LDS words.
__device__ __forceinline__ uint32_t block_append(uint32_t* counter, const bool want, uint32_t* s_tmp) {
    const unsigned long long m = __ballot(want);
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
    if (lane == 0) s_tmp[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (unsigned k = 0; k < nw; k++) {
            const uint32_t c = s_tmp[k];
            s_tmp[k] = tot;
            tot += c;
        }
        s_tmp[nw] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t idx = s_tmp[nw] + s_tmp[w] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // s_tmp is reused by the next call
    return idx;
}
#define CGRT_SHADE_BLOCK 1024

// Everything a hit spawns, in one pass: pointInShadow's rays (main.cpp:104-111) for every light, appended to the level's
// shadow list, and shade's mirror ray (:246-258), appended to the next level's list -- the mirror ray does not depend on
// the shadow results, so its batch can be traversed on a second stream while this level's shadow batch runs.
// lvl[2 i + 1] = {ks.xyz, child}; child = index of the mirror ray's entry on the next level, -1 when none was spawned.
// counters[0] += shadow rays, counters[1] += mirror rays, counters[2] += hits.
__global__ __launch_bounds__(CGRT_SHADE_BLOCK) void k_spawn(const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits,
                                                            const float* __restrict__ normals, const int* __restrict__ pixels,
                                                            unsigned long long n, const float* __restrict__ materials,
                                                            const float* __restrict__ lights, unsigned nlights, int spawn,
                                                            float* __restrict__ srays, float* __restrict__ sdist, int* __restrict__ sslot,
                                                            float4* __restrict__ lvl, float* __restrict__ next_rays,
                                                            int* __restrict__ next_pixels, uint32_t* __restrict__ counters,
                                                            const uint32_t* __restrict__ dcount) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount) {  // the list's length is only known on the device (the grid covers its capacity n)
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    const bool in = i < n;
    const bool hit = in && hits[i].hit != 0;
    F3 pointOn = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 0.f), ks = f3(0.f, 0.f, 0.f);
    if (hit) {
        const float* r = rays + 7 * i;
        d = ldv(r + 3);
        pointOn = add(ldv(r), scale(d, hits[i].t));
        const int mid = hits[i].material_id;
        // a hit that never wrote hitInfo.material (sphere only) reads an indeterminate Material upstream; default Material here
        ks = mid >= 0 ? ldv(materials + 8 * mid + 3) : f3(0.f, 0.f, 0.f);
    }
    __shared__ uint32_t s_tmp[CGRT_SHADE_BLOCK / 64 + 1];
    for (unsigned l = 0; l < nlights; l++) {
        const uint32_t idx = block_append(counters + 0, hit, s_tmp);
        if (in) sslot[i * nlights + l] = hit ? (int)idx : -1;
        if (hit) spawn_shadow_ray(lights, l, pointOn, idx, srays, sdist);
    }
    // :246 tests ks.z only (comma operator); `spawn` = level + 1 < maxLevel (:267)
    const bool wants_mirror = hit && !(ks.z <= 0.01f) && spawn;
    const uint32_t child = block_append(counters + 1, wants_mirror, s_tmp);
    if (wants_mirror) {
        spawn_mirror_ray(pointOn, d, ldv(normals + 3 * i), child, next_rays);
        next_pixels[child] = pixels[i];
    }
    if (in) lvl[2 * i + 1] = make_float4(ks.x, ks.y, ks.z, __int_as_float(wants_mirror ? (int)child : -1));
    // hits of the level: one atomic per workgroup
    const uint32_t h = (uint32_t)__popcll(__ballot(hit));
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (unsigned k = 0; k < (blockDim.x >> 6); k++) tot += s_tmp[k];
        if (tot) atomicAdd(counters + 2, tot);
    }
}

// shading (main.cpp:160-235) for one level: lvl[2 i] = {direct light.xyz, flags}, flags bit0 = hit.
__global__ __launch_bounds__(CGRT_SHADE_BLOCK) void k_shade(const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits,
                                                            const float* __restrict__ normals, const CgrtHitDev* __restrict__ shits,
                                                            const float* __restrict__ sdist, const int* __restrict__ sslot,
                                                            unsigned long long n, const float* __restrict__ materials,
                                                            const float* __restrict__ lights, unsigned nlights,
                                                            const float* __restrict__ slights, unsigned nslights,
                                                            const uint32_t* __restrict__ lit, unsigned samples, float4* __restrict__ lvl,
                                                            const uint32_t* __restrict__ dcount) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount) {
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    if (i >= n) return;
    float4 out0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hits[i].hit) {
        const float* r = rays + 7 * i;
        const F3 o = ldv(r), d = ldv(r + 3);
        const F3 nrm = ldv(normals + 3 * i);
        const F3 pointOn = add(o, scale(d, hits[i].t));
        const int mid = hits[i].material_id;
        const F3 kd = mid >= 0 ? ldv(materials + 8 * mid) : f3(0.f, 0.f, 0.f);
        const F3 ks = mid >= 0 ? ldv(materials + 8 * mid + 3) : f3(0.f, 0.f, 0.f);
        const float shininess = mid >= 0 ? materials[8 * mid + 6] : 1.0f;
        const float eps = 0.001f;
        F3 result = f3(0.f, 0.f, 0.f);
        const float dn = dot(nrm, d);  // glm::reflect(I, N) = I - N * dot(N, I) * 2
        const F3 refl = normalize(sub(d, scale(scale(nrm, dn), 2.0f)));
        for (unsigned l = 0; l < nslights; l++) {  // spherical lights first (main.cpp:168-218)
            const F3 lpos = ldv(slights + 7 * l), lcol = ldv(slights + 7 * l + 4);
            const F3 toLight = normalize(sub(lpos, pointOn));
            const float dc = dot(toLight, nrm);
            const F3 dif = dc <= 0 ? f3(0.f, 0.f, 0.f) : f3(lcol.x * kd.x * dc, lcol.y * kd.y * dc, lcol.z * kd.z * dc);
            const float sc = dot(refl, toLight);
            F3 spec = f3(0.f, 0.f, 0.f);
            if (!(sc <= 0)) {
                const float p = __builtin_powf(sc, shininess);
                spec = f3(lcol.x * ks.x * p, lcol.y * ks.y * p, lcol.z * ks.z * p);
            }
            // softShadowCounter: `samples` additions of 1.0f (exact), then / 200.0f (:200)
            const float counter = (float)lit[i * nslights + l] / (float)samples;
            result = add(result, scale(dif, counter));
            result = add(result, scale(spec, counter));
        }
        for (unsigned l = 0; l < nlights; l++) {
            const F3 lpos = ldv(lights + 6 * l), lcol = ldv(lights + 6 * l + 3);
            const F3 toLight = normalize(sub(lpos, pointOn));
            const int slot = sslot[i * nlights + l];
            const CgrtHitDev sh = shits[slot];
            const bool inShadow = sh.hit && !(sh.t + eps >= sdist[slot]);  // main.cpp:118-130
            if (inShadow) continue;
            const float dc = dot(toLight, nrm);  // diffuseOneLight :84-98
            const F3 dif = dc <= 0 ? f3(0.f, 0.f, 0.f) : f3(lcol.x * kd.x * dc, lcol.y * kd.y * dc, lcol.z * kd.z * dc);
            const float sc = dot(refl, toLight);  // specularOneLight :61-82
            F3 spec = f3(0.f, 0.f, 0.f);
            if (!(sc <= 0)) {
                const float p = __builtin_powf(sc, shininess);
                spec = f3(lcol.x * ks.x * p, lcol.y * ks.y * p, lcol.z * ks.z * p);
            }
            result = add(result, dif);
            result = add(result, spec);
        }
        out0 = make_float4(result.x, result.y, result.z, __uint_as_float(1u));
    }
    lvl[2 * i] = out0;
}

// colour = !hit ? 0 : (ks.z <= 0.01 ? direct : direct + childColour * ks)   (main.cpp:248, :262, :293); the child's
// record already holds its folded colour (levels are folded deepest first).
__global__ void k_fold(float4* __restrict__ lvl, const float4* __restrict__ child_lvl, unsigned long long n,
                       const uint32_t* __restrict__ dcount) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount) {  // (as k_spawn: the grid covers the list's capacity)
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    if (i >= n) return;
    const float4 a = lvl[2 * i], b = lvl[2 * i + 1];
    const int child = __float_as_int(b.w);
    if (!(__float_as_uint(a.w) & 1u) || (b.z <= 0.01f) || child < 0) return;  // no child: colour + 0 * ks = colour
    const float4 c = child_lvl[2 * (unsigned long long)child];  // a child that missed holds colour 0
    lvl[2 * i] = make_float4(a.x + c.x * b.x, a.y + c.y * b.y, a.z + c.z * b.z, a.w);
}

// child_lvl (optional): the fold of level 0 with level 1 (k_fold's arithmetic) happens here, one launch less
__global__ void k_write_rgb(const float4* __restrict__ lvl0, const float4* __restrict__ child_lvl, unsigned long long n,
                            const int* __restrict__ item_pixels, float* __restrict__ rgb, const uint32_t* __restrict__ dcount) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount) {
        const unsigned long long present = *dcount;
        n = present < n ? present : n;
    }
    if (i >= n) return;
    const long long pix = item_pixels[i];
    if (pix < 0) return;  // item outside the frame
    float4 a = lvl0[2 * i];
    if (child_lvl) {
        const float4 b = lvl0[2 * i + 1];
        const int child = __float_as_int(b.w);
        if ((__float_as_uint(a.w) & 1u) && !(b.z <= 0.01f) && child >= 0) {
            const float4 c = child_lvl[2 * (unsigned long long)child];
            a = make_float4(a.x + c.x * b.x, a.y + c.y * b.y, a.z + c.z * b.z, a.w);
        }
    }
    rgb[3 * pix] = a.x;
    rgb[3 * pix + 1] = a.y;
    rgb[3 * pix + 2] = a.z;
}

static inline unsigned grid_for(unsigned long long n, unsigned block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_spawn(const float* rays, const CgrtHitDev* hits, const float* normals, const int* pixels, unsigned long long n,
                        const float* materials, const float* lights, unsigned nlights, int spawn, float* srays, float* sdist, int* sslot,
                        float* lvl, float* next_rays, int* next_pixels, uint32_t* counters, hipStream_t s, const uint32_t* dcount) {
    if (n)
        hipLaunchKernelGGL(k_spawn, dim3(grid_for(n, CGRT_SHADE_BLOCK)), dim3(CGRT_SHADE_BLOCK), 0, s, rays, hits, normals, pixels, n, materials,
                           lights, nlights, spawn, srays, sdist, sslot, reinterpret_cast<float4*>(lvl), next_rays, next_pixels, counters, dcount);
    return hipGetLastError();
}
hipError_t launch_shade(const float* rays, const CgrtHitDev* hits, const float* normals, const CgrtHitDev* shits, const float* sdist,
                        const int* sslot, unsigned long long n, const float* materials, const float* lights, unsigned nlights,
                        const float* slights, unsigned nslights, const uint32_t* lit, unsigned samples, float* lvl, hipStream_t s,
                        const uint32_t* dcount) {
    if (n)
        hipLaunchKernelGGL(k_shade, dim3(grid_for(n, CGRT_SHADE_BLOCK)), dim3(CGRT_SHADE_BLOCK), 0, s, rays, hits, normals, shits, sdist, sslot, n,
                           materials, lights, nlights, slights, nslights, lit, samples, reinterpret_cast<float4*>(lvl), dcount);
    return hipGetLastError();
}
hipError_t launch_fold(float* lvl, const float* child_lvl, unsigned long long n, hipStream_t s, const uint32_t* dcount) {
    if (n)
        hipLaunchKernelGGL(k_fold, dim3(grid_for(n, 256)), dim3(256), 0, s, reinterpret_cast<float4*>(lvl),
                           reinterpret_cast<const float4*>(child_lvl), n, dcount);
    return hipGetLastError();
}
hipError_t launch_write_rgb(const float* lvl0, const float* child_lvl, unsigned long long n, const int* item_pixels, float* rgb, hipStream_t s,
                            const uint32_t* dcount) {
    if (n)
        hipLaunchKernelGGL(k_write_rgb, dim3(grid_for(n, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(lvl0),
                           reinterpret_cast<const float4*>(child_lvl), n, item_pixels, rgb, dcount);
    return hipGetLastError();
}

}  // namespace cgrt
