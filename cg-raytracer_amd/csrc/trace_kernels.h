// trace_kernels.h -- host-callable launchers of the gfx950 kernels in trace_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cgrt_layout.h"
#include "spawn_rays.h"

namespace cgrt {

static const unsigned CGRT_QUEUE_BLOCK_WORDS = 32 * 9;

// Device mirror of CgrtHit (include/cgrt.h), 16 B.
struct CgrtHitDev {
    float t;
    uint32_t prim_id;
    int32_t material_id;
    uint32_t hit;
};

// Soft-shadow sampling of spherical lights (main.cpp:168-218) for one recursion level.
struct SoftDev {
    const float* lights;  // nlights x 7 {position, radius, color} (SphericalLight, scene.h:47-51)
    const float* units;   // nunits x 3 unit vectors: the randomUnitVector() draws (main.cpp:46-59)
    uint32_t nlights, samples, nunits, seed, level;
};

// threads per workgroup the ray-list kernels (batch, soft shadow) are launched with for this scene; frames carry theirs in FrameDev::block
int trace_block(const SceneDev& S);
// kernel shape per launch (walk_quad.h): mode -1 = by size (launches of at most max_rays rays take the quad shape), 0 = never, 1 = always;
// max_rays 0 keeps the current threshold.  Results do not depend on the shape.
void set_quad_shape(int mode, unsigned long long max_rays);
void get_quad_shape(int* mode, unsigned long long* max_rays);
bool quad_shape_for(const SceneDev& S, unsigned long long rays);
// counters (optional): 5 x u64 device words {rays, inner_visits, leaf_visits, tri_tests, sub_visits}, accumulated.
hipError_t launch_trace_primary(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                unsigned long long* counters, hipStream_t stream);
// persistent variant: queue = CGRT_QUEUE_BLOCK_WORDS zeroed u32 (8 heads + exit counter, one 128-B line each) owned by this launch; blocks = persistent grid size
hipError_t launch_trace_primary_persistent(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                           unsigned long long* counters, unsigned int* queue, unsigned blocks, hipStream_t stream);
// diagnostic: stamps = 4 x ntiles_rank u64 {memtime start, end, memrealtime start, end} per wave
hipError_t launch_trace_primary_stamped(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits,
                                        unsigned long long* stamps, hipStream_t stream);
// dcount (optional): device word with the number of rays actually present (<= n, the capacity the grid is sized for)
hipError_t launch_trace_batch(const SceneDev& S, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                              unsigned long long* counters, hipStream_t stream, const uint32_t* dcount = nullptr,
                              unsigned long long expected = 0);  // expected (with dcount): the host's estimate of *dcount picks the kernel shape
// every triangle, no tree (ray_tracing.cpp:202-213); mesh < 0: all meshes + spheres
hipError_t launch_brute_batch(const SceneDev& S, const float* rays, unsigned long long n, int mesh, CgrtHitDev* hits, float* normals, hipStream_t s);
// pointInShadow's rays (main.cpp:104-135): hits[i] decides `hit && !(t + 0.001f >= dist[i])` like the reference's closest hit does
hipError_t launch_trace_shadow(const SceneDev& S, const float* rays, const float* dist, unsigned long long n, CgrtHitDev* hits, hipStream_t stream,
                               const uint32_t* dcount = nullptr, unsigned long long* counters = nullptr, unsigned long long expected = 0,
                               unsigned dmul = 1);  // dmul: the list holds *dcount x dmul rays
// a level's shadow list (srays .. sexpected, as launch_trace_shadow) and its mirror list (rays .. expected, as launch_trace_batch) in ONE
// launch, workgroups dealt alternately; can_trace_pair: the scene has a fast tree and the forced shape (if any) is a lane shape
bool can_trace_pair(const SceneDev& S);
hipError_t launch_trace_pair(const SceneDev& S, const float* srays, const float* sdist, unsigned long long ns, CgrtHitDev* shits, const uint32_t* sdcount,
                             unsigned sdmul, unsigned long long sexpected, const float* rays, unsigned long long n, CgrtHitDev* hits, float* normals,
                             const uint32_t* dcount, unsigned long long expected, hipStream_t stream);
// *flag = value (system scope) once everything queued on the stream before it has finished
hipError_t launch_signal(uint32_t* flag, uint32_t value, hipStream_t stream);
hipError_t launch_generate_rays(const CameraDev& C, int W, int H, int x0, int y0, int x1, int y1, float* rays, hipStream_t stream);
// lit[item * nlights + l] += samples of spherical light l that reach it from item's hit point (zeroed by the caller)
hipError_t launch_soft_shadow(const SceneDev& S, const SoftDev& Q, const float* rays, const CgrtHitDev* hits, const int* item_pixels,
                              unsigned long long nitems, uint32_t* lit, int anyhit, hipStream_t stream);

// primary frame for the shading wavefront: only the hits, appended to a compact list; count = one zeroed device word
hipError_t launch_trace_primary_compact(const SceneDev& S, const CameraDev& C, const FrameDev& F, float* rays, CgrtHitDev* hits, float* normals,
                                        int* pixels, uint32_t* count, hipStream_t stream, unsigned long long* counters = nullptr,
                                        float* rgb = nullptr, const SpawnDev* spawn = nullptr);  // spawn (device memory): level 0's k_spawn fused in (spawn_rays.h)  // rgb (optional): the rank's pixels are cleared by the same kernel
hipError_t launch_clear_owned(const FrameDev& F, float* rgb, hipStream_t stream);
// shading wavefront (shade_kernels.hip); every level is a compact list of live paths
// counters: 3 device words {shadow rays appended, mirror rays appended, hits}, zeroed by the caller
// k_spawn: shadow rays of every hit -> the level's shadow list; mirror rays -> the next level's list; lvl[2i+1] = {ks, child}
hipError_t launch_spawn(const float* rays, const CgrtHitDev* hits, const float* normals, const int* pixels, unsigned long long n,
                        const float* materials, const float* lights, unsigned nlights, int spawn, float* srays, float* sdist, int* sslot,
                        float* lvl, float* next_rays, int* next_pixels, uint32_t* counters, hipStream_t s, const uint32_t* dcount = nullptr);
// k_shade: lvl[2i] = {direct light, flags}
hipError_t launch_shade(const float* rays, const CgrtHitDev* hits, const float* normals, const CgrtHitDev* shits, const float* sdist,
                        const int* sslot, unsigned long long n, const float* materials, const float* lights, unsigned nlights,
                        const float* slights, unsigned nslights, const uint32_t* lit, unsigned samples, float* lvl, hipStream_t s,
                        const uint32_t* dcount = nullptr);  // dcount (optional): device word with the list's length (<= n, the capacity the grid covers)
// colour of level `lvl` entries += colour of their child (level lvl + 1) * ks  (main.cpp:262)
// dcount (optional, also launch_write_rgb): the list's length on the device; n is then its capacity
hipError_t launch_fold(float* lvl, const float* child_lvl, unsigned long long n, hipStream_t s, const uint32_t* dcount = nullptr);
// child_lvl (optional): level 0 is folded with level 1 on the fly (colour + childColour * ks, main.cpp:262) instead of by launch_fold
hipError_t launch_write_rgb(const float* lvl0, const float* child_lvl, unsigned long long n, const int* item_pixels, float* rgb, hipStream_t s,
                            const uint32_t* dcount = nullptr);
hipError_t launch_gather_calib(const void* table, unsigned long long nrecords, unsigned long long mult, unsigned long long add, float* sink,
                               hipStream_t s);
hipError_t launch_fastdiv_check(const float* a, const float* d, unsigned long long n, unsigned long long* mismatches, float* first_bad,
                                hipStream_t s);
hipError_t launch_ray_triangle(const float* tri, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, float* normals,
                               hipStream_t s);
hipError_t launch_ray_plane(const float* plane, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, hipStream_t s);
hipError_t launch_ray_box(const float* box, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, uint8_t* inside,
                          hipStream_t s);
hipError_t launch_ray_sphere(const float* sph, const float* rays, unsigned long long n, float* t_out, uint8_t* hit, float* normals,
                             hipStream_t s);
hipError_t launch_triangle_plane(const float* tri, unsigned long long n, float* plane, hipStream_t s);
hipError_t launch_point_in_triangle(const float* in, unsigned long long n, uint8_t* out, hipStream_t s);

}  // namespace cgrt
