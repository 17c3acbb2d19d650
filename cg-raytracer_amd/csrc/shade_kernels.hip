// shade_kernels.hip -- the reference's shading / recursion driver (src/main.cpp:61-310) on the device, as a wavefront:
// per recursion level one kernel spawns the shadow rays of every hit (pointInShadow, :104-135), the batch traversal
// kernel answers them, one kernel evaluates the Phong terms (:61-98, :219-232) and spawns the mirror rays (shade,
// :241-264), and after the last level one kernel folds the levels back (color = direct + reflected * ks, :262).
// Paths are never compacted: item i of every level belongs to the same pixel (items are in the primary kernel's
// frame order -- tiles, super-tiles -- so every batch stays tile-coherent); dead paths carry a "null ray" that fails the
// root gate of the traversal at once.  Spherical lights (:168-218) are sampled by k_soft_shadow (trace_kernels.hip), which
// leaves the number of unoccluded samples per (item, light); the draws of randomUnitVector() are a caller-supplied table.
// Arithmetic follows the reference's expression order (cgrt_math.h); pow(float, float) is powf (device libm: the last
// ulp may differ from glibc's -- the RGB parity bar is 1e-5 absolute).
#include <hip/hip_runtime.h>

#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "trace_kernels.h"

namespace cgrt {

__device__ __forceinline__ F3 ldv(const float* p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ void null_ray(float* r) {  // fails intersectDataStructure's gate for any finite box
    r[0] = r[1] = r[2] = 3.402823466e+38f;
    r[3] = 1.0f;
    r[4] = r[5] = 0.0f;
    r[6] = 0.0f;
}

// pointInShadow's ray construction (main.cpp:104-111) for every (item, light).
__global__ void k_spawn_shadow(const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits, unsigned long long n,
                               const float* __restrict__ lights, unsigned nlights, float* __restrict__ srays, float* __restrict__ sdist) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool hit = hits[i].hit != 0;
    const float* r = rays + 7 * i;
    const F3 pointOn = add(ldv(r), scale(ldv(r + 3), hits[i].t));
    const float eps = 0.001f;
    for (unsigned l = 0; l < nlights; l++) {
        float* s = srays + 7 * (i * nlights + l);
        if (!hit) {
            null_ray(s);
            sdist[i * nlights + l] = 0.0f;
            continue;
        }
        const F3 toLight = sub(ldv(lights + 6 * l), pointOn);
        const F3 dir = normalize(toLight);
        const F3 o = add(pointOn, f3(eps * dir.x, eps * dir.y, eps * dir.z));  // ray.origin += epsilon * ray.direction
        s[0] = o.x;
        s[1] = o.y;
        s[2] = o.z;
        s[3] = dir.x;
        s[4] = dir.y;
        s[5] = dir.z;
        s[6] = 3.402823466e+38f;
        sdist[i * nlights + l] = length(toLight);
    }
}

// shading (main.cpp:219-232) + shade (:241-264) for one level.  lvl: per item {direct.xyz, flags} {ks.xyz, 0};
// flags bit0 = hit, bit1 = a mirror ray was spawned into next_rays[i].  stats[0..3] += hits, real shadow rays, mirror rays, soft-shadow samples.
__global__ void k_shade(const float* __restrict__ rays, const CgrtHitDev* __restrict__ hits, const float* __restrict__ normals,
                        const CgrtHitDev* __restrict__ shits, const float* __restrict__ sdist, unsigned long long n,
                        const float* __restrict__ materials, const float* __restrict__ lights, unsigned nlights,
                        const float* __restrict__ slights, unsigned nslights, const uint32_t* __restrict__ lit, unsigned samples, int spawn,
                        float4* __restrict__ lvl, float* __restrict__ next_rays, unsigned long long* __restrict__ stats) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 out0 = make_float4(0.f, 0.f, 0.f, 0.f), out1 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool spawned = false;
    if (hits[i].hit) {
        const float* r = rays + 7 * i;
        const F3 o = ldv(r), d = ldv(r + 3);
        const F3 nrm = ldv(normals + 3 * i);
        const F3 pointOn = add(o, scale(d, hits[i].t));
        const int mid = hits[i].material_id;
        // a hit that never wrote hitInfo.material (sphere only) reads an indeterminate Material upstream; default Material here
        const F3 kd = mid >= 0 ? ldv(materials + 8 * mid) : f3(0.f, 0.f, 0.f);
        const F3 ks = mid >= 0 ? ldv(materials + 8 * mid + 3) : f3(0.f, 0.f, 0.f);
        const float shininess = mid >= 0 ? materials[8 * mid + 6] : 1.0f;
        const float eps = 0.001f;
        F3 result = f3(0.f, 0.f, 0.f);
        const float dn0 = dot(nrm, d);  // glm::reflect(I, N) = I - N * dot(N, I) * 2
        const F3 refl0 = normalize(sub(d, scale(scale(nrm, dn0), 2.0f)));
        for (unsigned l = 0; l < nslights; l++) {  // spherical lights first (main.cpp:168-218)
            const F3 lpos = ldv(slights + 7 * l), lcol = ldv(slights + 7 * l + 4);
            const F3 toLight = normalize(sub(lpos, pointOn));
            const float dc = dot(toLight, nrm);
            const F3 dif = dc <= 0 ? f3(0.f, 0.f, 0.f) : f3(lcol.x * kd.x * dc, lcol.y * kd.y * dc, lcol.z * kd.z * dc);
            const float sc = dot(refl0, toLight);
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
            const CgrtHitDev sh = shits[i * nlights + l];
            const bool inShadow = sh.hit && !(sh.t + eps >= sdist[i * nlights + l]);  // main.cpp:118-130
            if (inShadow) continue;
            const float dc = dot(toLight, nrm);  // diffuseOneLight :84-98
            const F3 dif = dc <= 0 ? f3(0.f, 0.f, 0.f) : f3(lcol.x * kd.x * dc, lcol.y * kd.y * dc, lcol.z * kd.z * dc);
            const float dn = dot(nrm, d);  // glm::reflect(I, N) = I - N * dot(N, I) * 2
            const F3 refl = normalize(sub(d, scale(scale(nrm, dn), 2.0f)));
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
        out1 = make_float4(ks.x, ks.y, ks.z, 0.f);
        if (!(ks.z <= 0.01f) && spawn) {  // :246 tests ks.z only (comma operator); `spawn` = level + 1 < maxLevel (:267)
            const float dn = dot(nrm, d);
            const F3 refl = normalize(sub(d, scale(scale(nrm, dn), 2.0f)));
            const F3 ro = add(pointOn, f3(eps * refl.x, eps * refl.y, eps * refl.z));
            float* q = next_rays + 7 * i;
            q[0] = ro.x;
            q[1] = ro.y;
            q[2] = ro.z;
            q[3] = refl.x;
            q[4] = refl.y;
            q[5] = refl.z;
            q[6] = length(d);  // :254: t = |direction| of the parent ray
            spawned = true;
            out0.w = __uint_as_float(3u);
        }
    }
    if (spawn && !spawned) null_ray(next_rays + 7 * i);
    lvl[2 * i] = out0;
    lvl[2 * i + 1] = out1;
    // statistics: one atomic per wave and counter
    const unsigned long long h = __popcll(__ballot(hits[i].hit != 0)), s = __popcll(__ballot(spawned));
    if ((threadIdx.x & 63) == 0) {
        if (h) {
            atomicAdd(stats + 0, h);
            atomicAdd(stats + 1, h * nlights);
            if (nslights) atomicAdd(stats + 3, h * nslights * samples);
        }
        if (s) atomicAdd(stats + 2, s);
    }
}

// color_l = !hit ? 0 : (ks.z <= 0.01 ? direct : direct + color_{l+1} * ks)   (main.cpp:248, :262, :293)
__global__ void k_combine(const float4* __restrict__ levels, int nlevels, unsigned long long n, const int* __restrict__ item_pixels,
                          float* __restrict__ rgb) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long pix = item_pixels[i];
    if (pix < 0) return;  // item outside the frame
    F3 below = f3(0.f, 0.f, 0.f);
    for (int l = nlevels - 1; l >= 0; l--) {
        const float4 a = levels[((unsigned long long)l * n + i) * 2], b = levels[((unsigned long long)l * n + i) * 2 + 1];
        const unsigned flags = __float_as_uint(a.w);
        F3 c = f3(0.f, 0.f, 0.f);
        if (flags & 1u) {
            c = f3(a.x, a.y, a.z);
            if (!(b.z <= 0.01f)) {
                const F3 child = (flags & 2u) ? below : f3(0.f, 0.f, 0.f);
                c = add(c, f3(child.x * b.x, child.y * b.y, child.z * b.z));
            }
        }
        below = c;
    }
    rgb[3 * pix] = below.x;
    rgb[3 * pix + 1] = below.y;
    rgb[3 * pix + 2] = below.z;
}

static inline unsigned grid_for(unsigned long long n, unsigned block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_spawn_shadow(const float* rays, const CgrtHitDev* hits, unsigned long long n, const float* lights, unsigned nlights,
                               float* srays, float* sdist, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_spawn_shadow, dim3(grid_for(n, 256)), dim3(256), 0, s, rays, hits, n, lights, nlights, srays, sdist);
    return hipGetLastError();
}
hipError_t launch_shade(const float* rays, const CgrtHitDev* hits, const float* normals, const CgrtHitDev* shits, const float* sdist,
                        unsigned long long n, const float* materials, const float* lights, unsigned nlights, const float* slights,
                        unsigned nslights, const uint32_t* lit, unsigned samples, int spawn, float* lvl, float* next_rays,
                        unsigned long long* stats, hipStream_t s) {
    if (n)
        hipLaunchKernelGGL(k_shade, dim3(grid_for(n, 256)), dim3(256), 0, s, rays, hits, normals, shits, sdist, n, materials, lights, nlights,
                           slights, nslights, lit, samples, spawn, reinterpret_cast<float4*>(lvl), next_rays, stats);
    return hipGetLastError();
}
hipError_t launch_combine(const float* levels, int nlevels, unsigned long long n, const int* item_pixels, float* rgb, hipStream_t s) {
    if (n)
        hipLaunchKernelGGL(k_combine, dim3(grid_for(n, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(levels), nlevels, n,
                           item_pixels, rgb);
    return hipGetLastError();
}

}  // namespace cgrt
