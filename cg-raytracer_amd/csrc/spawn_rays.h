// spawn_rays.h -- what a hit spawns (main.cpp:104-111 pointInShadow's ray, :246-258 shade's mirror ray), as the two kernels that
// do it share it: k_spawn (shade_kernels.hip; any level, from the level's compact list) and the primary kernel of a predicted
// frame (trace_kernels.hip k_trace_primary_compact; level 0, straight from the registers of the lane that found the hit).
// One set of expressions, so a frame does not depend on which of the two ran.
#pragma once
#include <hip/hip_runtime.h>

#include "cgrt_layout.h"
#include "cgrt_math.h"

namespace cgrt {

// Level-0 spawn fused into the primary kernel (which takes a pointer to one of these in device memory, nullptr = not fused).  Every
// entry of level 0 is a hit, so its shadow rays need no append: the ray towards light l of entry i is shadow ray i * nlights + l,
// and the list holds hits x nlights rays.  Mirror rays are counted in the word behind the kernel's hit counter (one 64-bit atomic).
struct SpawnDev {
    const float* materials;
    const float* lights;  // nlights x {position, colour}
    unsigned nlights;
    int spawn;            // level + 1 < maxLevel (main.cpp:267): mirror rays are wanted
    float* srays;
    float* sdist;
    int* sslot;
    float4* lvl;
    float* next_rays;
    int* next_pixels;
};

// the shadow ray from pointOn towards the light at lights[6 l]: srays[7 idx ..], sdist[idx]
__device__ __forceinline__ void spawn_shadow_ray(const float* __restrict__ lights, unsigned l, const F3 pointOn, unsigned long long idx,
                                                 float* __restrict__ srays, float* __restrict__ sdist) {
    const float eps = 0.001f;
    const float* lp = lights + 6 * l;
    const F3 toLight = sub(f3(lp[0], lp[1], lp[2]), pointOn);
    const F3 dir = normalize(toLight);
    const F3 o = add(pointOn, f3(eps * dir.x, eps * dir.y, eps * dir.z));  // ray.origin += epsilon * ray.direction
    float* s = srays + 7ull * idx;
    s[0] = o.x;
    s[1] = o.y;
    s[2] = o.z;
    s[3] = dir.x;
    s[4] = dir.y;
    s[5] = dir.z;
    s[6] = 3.402823466e+38f;
    sdist[idx] = length(toLight);
}

// the mirror ray of a ray with direction d that hit at pointOn with normal nrm: next_rays[7 child ..]
__device__ __forceinline__ void spawn_mirror_ray(const F3 pointOn, const F3 d, const F3 nrm, unsigned long long child, float* __restrict__ next_rays) {
    const float eps = 0.001f;
    const float dn = dot(nrm, d);  // glm::reflect(I, N) = I - N * dot(N, I) * 2
    const F3 refl = normalize(sub(d, scale(scale(nrm, dn), 2.0f)));
    const F3 ro = add(pointOn, f3(eps * refl.x, eps * refl.y, eps * refl.z));
    float* q = next_rays + 7ull * child;
    q[0] = ro.x;
    q[1] = ro.y;
    q[2] = ro.z;
    q[3] = refl.x;
    q[4] = refl.y;
    q[5] = refl.z;
    q[6] = length(d);  // :254: t = |direction| of the parent ray
}

}  // namespace cgrt
