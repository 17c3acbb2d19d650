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
// The per-lane stack (<= 11 entries, bvh.cpp:48) lives in LDS, lane-interleaved so that every access
// is bank-conflict free; runtime-indexed register arrays would go to scratch.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no FMA contraction), default IEEE div/sqrt,
// denormals on -- see cgrt_math.h for why.
#include <hip/hip_runtime.h>

#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "trace_kernels.h"

namespace cgrt {

#define CGRT_BLOCK 256
#define CGRT_STACK (MAX_LEVELS)

struct LaneCounters {
    uint32_t inner = 0, leaf = 0, tri = 0, sub = 0;
};

__device__ __forceinline__ F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }

// Ordered closest-hit walk of the reference tree for one ray.
//   t        in/out ray.t
//   hit_rec  leaf-order index of the last accepted triangle (REF_NONE if none)
template <bool COUNT>
__device__ __forceinline__ void walk_tree(const SceneDev& S, const F3 o, const F3 d, float& t, uint32_t& hit_rec,
                                          uint32_t* __restrict__ stk_ref, float* __restrict__ stk_t, LaneCounters& cnt) {
    uint32_t cur = REF_NONE;
    if (S.root_ref != REF_NONE) {
        // intersectDataStructure, bvh.cpp:831-844 (the box test's write to ray.t is undone there)
        const F3 lo = f3(S.root_box.lo[0], S.root_box.lo[1], S.root_box.lo[2]);
        const F3 hi = f3(S.root_box.hi[0], S.root_box.hi[1], S.root_box.hi[2]);
        float tb;
        if (starts_in_box(o, lo, hi) || ray_box(lo, hi, o, d, t, tb)) cur = S.root_ref;
    }
    int sp = 0;
    for (;;) {
        if (cur == REF_NONE) {
            bool found = false;
            while (sp > 0) {
                --sp;
                const float ts = stk_t[sp * CGRT_BLOCK];
                const uint32_t r = stk_ref[sp * CGRT_BLOCK];
                if (!(t < ts)) {  // bvh.cpp:582: skip the deferred child iff ray.t < tSecond
                    cur = r;
                    found = true;
                    break;
                }
            }
            if (!found) break;
        }
        if (cur & REF_LEAF) {
            // intersectLeaf, bvh.cpp:535-553: linear scan in leaf order
            const LeafRec L = S.leaves[cur & ~REF_LEAF];
            if (COUNT) {
                cnt.leaf++;
                cnt.tri += L.count;
            }
            for (uint32_t k = 0; k < L.count; k++) {
                const float4* q = reinterpret_cast<const float4*>(S.tris + (L.first + k));
                const float4 a = q[0], b = q[1], c = q[2], e = q[3];
                const F3 v0 = f3(a.x, a.y, a.z), v1 = f3(a.w, b.x, b.y), v2 = f3(b.z, b.w, c.x);
                const F3 n = f3(c.y, c.z, c.w);
                if (ray_triangle_geom(v0, v1, v2, n, e.x, o, d, t)) hit_rec = L.first + k;
            }
            cur = REF_NONE;
            continue;
        }
        // intersectNonLeaf, bvh.cpp:715-736
        if (COUNT) cnt.inner++;
        const float4* q = reinterpret_cast<const float4*>(S.packets + cur);
        const float4 a = q[0], b = q[1], c = q[2];
        const uint4 m = *reinterpret_cast<const uint4*>(q + 3);
        const F3 llo = f3(a.x, a.y, a.z), lhi = f3(a.w, b.x, b.y);
        const F3 rlo = f3(b.z, b.w, c.x), rhi = f3(c.y, c.z, c.w);
        float tL = -1.0f, tR = -1.0f, tb;
        if (ray_box(llo, lhi, o, d, t, tb)) tL = tb;
        if (ray_box(rlo, rhi, o, d, t, tb)) tR = tb;
        const bool inL = starts_in_box(o, llo, lhi), inR = starts_in_box(o, rlo, rhi);
        uint32_t first = REF_NONE, second = REF_NONE;
        float tsec = 0.0f;
        if (inL && inR) {  // intersectDeeper :685-688, left then right, no culling
            first = m.x;
            second = m.y;
            tsec = -__builtin_inff();
        } else if (inL) {  // :689-692
            first = m.x;
            if (!(tR < 0)) {
                second = m.y;
                tsec = tR;
            }
        } else if (inR) {  // :693-696
            first = m.y;
            if (!(tL < 0)) {
                second = m.x;
                tsec = tL;
            }
        } else {  // intersectRayThatStartsOutsideBoxes :611-635
            const bool ml = tL < 0, mr = tR < 0;
            if (ml && mr) {
            } else if (ml) {
                first = m.y;
            } else if (mr) {
                first = m.x;
            } else if (tL < tR) {
                first = m.x;
                second = m.y;
                tsec = tR;
            } else {
                first = m.y;
                second = m.x;
                tsec = tL;
            }
        }
        if (second != REF_NONE) {
            stk_ref[sp * CGRT_BLOCK] = second;
            stk_t[sp * CGRT_BLOCK] = tsec;
            ++sp;
        }
        cur = first;
    }
}

// Spheres (bvh.cpp:878-879), result assembly and the accepted hit's interpolated normal
// (ray_tracing.cpp:94-107), which depends only on the final (triangle, t).
__device__ __forceinline__ void finish_ray(const SceneDev& S, const F3 o, const F3 d, float t, uint32_t hit_rec, CgrtHitDev* out,
                                           float* out_normal) {
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
    CgrtHitDev h;
    h.t = t;
    h.prim_id = prim;
    h.material_id = mat;
    h.hit = hit ? 1u : 0u;
    *out = h;
    if (out_normal && hit) {
        F3 nn = sn;
        if (!sphere_last) {
            const TriRecord* T = S.tris + hit_rec;
            const TriNormals* N = S.tri_normals + hit_rec;
            nn = hit_normal(ld3(T->v0), ld3(T->v1), ld3(T->v2), ld3(T->n), ld3(N->n1), ld3(N->n2), ld3(N->n3), o, d, t);
        }
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

// Wave w of the launch owns tile `rank + nranks * w` of the rectangle's 8x8 tiling; lane -> pixel.
__device__ __forceinline__ bool tile_pixel(const FrameDev& F, uint32_t wave_global, int lane, int& x, int& y) {
    if (wave_global >= F.ntiles_rank) return false;
    const uint32_t tile = (uint32_t)F.rank + (uint32_t)F.nranks * wave_global;
    const int tx = (int)(tile % (uint32_t)F.tiles_x), ty = (int)(tile / (uint32_t)F.tiles_x);
    x = F.x0 + tx * 8 + (lane & 7);
    y = F.y0 + ty * 8 + (lane >> 3);
    return x < F.x1 && y < F.y1;
}

template <bool COUNT>
__global__ __launch_bounds__(CGRT_BLOCK) void k_trace_primary(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits,
                                                              float* __restrict__ normals, unsigned long long* counters) {
    __shared__ uint32_t s_ref[CGRT_STACK * CGRT_BLOCK];
    __shared__ float s_t[CGRT_STACK * CGRT_BLOCK];
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = blockIdx.x * (CGRT_BLOCK / 64) + (threadIdx.x >> 6);
    int x = 0, y = 0;
    const bool active = tile_pixel(F, wave_global, lane, x, y);
    LaneCounters cnt;
    if (active) {
        F3 o, d;
        primary_ray(C, F.W, F.H, x, y, o, d);
        float t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
        uint32_t hit_rec = REF_NONE;
        walk_tree<COUNT>(S, o, d, t, hit_rec, s_ref + threadIdx.x, s_t + threadIdx.x, cnt);
        const size_t pix = (size_t)y * F.W + x;
        finish_ray(S, o, d, t, hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
    }
    if (COUNT) flush_counters(cnt, active, counters);
}

template <bool COUNT>
__global__ __launch_bounds__(CGRT_BLOCK) void k_trace_batch(SceneDev S, const float* __restrict__ rays, unsigned long long n,
                                                            CgrtHitDev* __restrict__ hits, float* __restrict__ normals,
                                                            unsigned long long* counters) {
    __shared__ uint32_t s_ref[CGRT_STACK * CGRT_BLOCK];
    __shared__ float s_t[CGRT_STACK * CGRT_BLOCK];
    const unsigned long long i = (unsigned long long)blockIdx.x * CGRT_BLOCK + threadIdx.x;
    const bool active = i < n;
    LaneCounters cnt;
    if (active) {
        const float* r = rays + 7 * i;
        const F3 o = f3(r[0], r[1], r[2]), d = f3(r[3], r[4], r[5]);
        float t = r[6];
        uint32_t hit_rec = REF_NONE;
        walk_tree<COUNT>(S, o, d, t, hit_rec, s_ref + threadIdx.x, s_t + threadIdx.x, cnt);
        finish_ray(S, o, d, t, hit_rec, hits + i, normals ? normals + 3 * i : nullptr);
    }
    if (COUNT) flush_counters(cnt, active, counters);
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

// ---------------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(unsigned long long n, unsigned block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_trace_primary(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                unsigned long long* counters, hipStream_t stream) {
    if (F.ntiles_rank == 0) return hipSuccess;
    const unsigned blocks = grid_for(F.ntiles_rank, CGRT_BLOCK / 64);
    if (counters)
        hipLaunchKernelGGL(k_trace_primary<true>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters);
    else
        hipLaunchKernelGGL(k_trace_primary<false>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters);
    return hipGetLastError();
}
hipError_t launch_trace_batch(const SceneDev& S, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                              unsigned long long* counters, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = grid_for(n, CGRT_BLOCK);
    if (counters)
        hipLaunchKernelGGL(k_trace_batch<true>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, rays, n, hits, normals, counters);
    else
        hipLaunchKernelGGL(k_trace_batch<false>, dim3(blocks), dim3(CGRT_BLOCK), 0, stream, S, rays, n, hits, normals, counters);
    return hipGetLastError();
}
hipError_t launch_generate_rays(const CameraDev& C, int W, int H, int x0, int y0, int x1, int y1, float* rays, hipStream_t stream) {
    const unsigned long long n = (unsigned long long)(x1 - x0) * (unsigned long long)(y1 - y0);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_rays, dim3(grid_for(n, CGRT_BLOCK)), dim3(CGRT_BLOCK), 0, stream, C, W, H, x0, y0, x1, y1, rays);
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
