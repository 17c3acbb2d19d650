// renderRayTracing / getFinalColor / trace / shade / shading / pointInShadow (src/main.cpp:61-310, :648-720) as a
// host-driven WAVEFRONT: per recursion level, all shadow rays and all reflection rays of the frame go through the
// GPU path in batches (BoundingVolumeHierarchy::intersectBatch); the Phong arithmetic runs on the host in the
// reference's expression order.  Spherical lights (soft shadows, :168-218) send `samples` rays per hit through the same
// batched entry; upstream draws their directions from std::random_device (:46-59), here they come from a
// SoftShadowSampler: a table of randomUnitVector() draws + the integer hash documented at cgrt_render_soft
// (include/cgrt.h), so that a frame can be reproduced (and compared with the device driver and the oracle).
#pragma once
#include <vector>

#include "bounding_volume_hierarchy.h"
#include "screen.h"
#include "trackball.h"

struct RenderStats {
    uint64_t primary = 0, shadow = 0, reflection = 0, softShadow = 0;
    double seconds_total = 0, seconds_device = 0;
};

struct SoftShadowSampler {
    std::vector<cgrt::vec3> units;  // the draws of randomUnitVector()
    uint32_t samples = 200;         // main.cpp:176
    uint32_t seed = 0;
    // n draws the way upstream makes one (three std::normal_distribution<float>(0,1) values in the order y, x, s, then
    // glm::normalize, main.cpp:46-59), from a seeded engine instead of std::random_device.
    static SoftShadowSampler gaussian(uint32_t n = 1u << 16, uint32_t engineSeed = 1, uint32_t samples = 200, uint32_t seed = 0);
    const cgrt::vec3& draw(uint32_t pixel, uint32_t level, uint32_t light, uint32_t smp) const;
};

// maxLevel = 2 reproduces main.cpp:267 (`level >= 2` -> black): primary + one mirror bounce.
// The same frame with the whole driver ON THE DEVICE (cgrt_render_soft: primary rays, shadow and mirror batches and the
// Phong terms never leave the GPU; one download of W*H*3 floats): the fast path for a caller that wants pixels.
// RGB agrees with renderToBuffer to the 1e-5 parity bar (powf is the device's).
RenderStats renderToBufferOnDevice(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb,
                                   int maxLevel = 2, const SoftShadowSampler* sampler = nullptr);
RenderStats renderRayTracingOnDevice(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen,
                                     int maxLevel = 2, const SoftShadowSampler* sampler = nullptr);
// One frame on several devices: bvhs[i] is a replica of the scene's BVH on its own device
// (BoundingVolumeHierarchy(&scene, device)); replica i renders the 64x64 super-tiles i % n, shading included, and the
// frame is assembled in ONE Screen / buffer, as the reference's renderRayTracing yields one (main.cpp:648-720).
RenderStats renderToBufferOnDevices(const Scene& scene, const Trackball& camera, const std::vector<const BoundingVolumeHierarchy*>& bvhs, int W, int H,
                                    float* rgb, int maxLevel = 2, const SoftShadowSampler* sampler = nullptr);
RenderStats renderRayTracingOnDevices(const Scene& scene, const Trackball& camera, const std::vector<const BoundingVolumeHierarchy*>& bvhs,
                                      Screen& screen, int maxLevel = 2, const SoftShadowSampler* sampler = nullptr);
// The reference's driver LITERALLY (src/main.cpp:265-310, :648-696): an `omp parallel for` over the rows, per pixel the recursive
// getFinalColor -> trace -> shade -> shading -> pointInShadow, ONE BoundingVolumeHierarchy::intersect call per ray -- what an
// unchanged main.cpp does to the library.  The library combines the rays of concurrent callers into shared launches
// (include/cgrt.h cgrt_set_call_combining), so the frame's speed grows with the number of calling threads; `threads` = 0 takes
// OpenMP's default, and MORE threads than cores pay: a caller spends its time waiting for the GPU round trip.
RenderStats renderToBufferPerRay(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb,
                                 int maxLevel = 2, const SoftShadowSampler* sampler = nullptr, int threads = 0);
RenderStats renderRayTracingPerRay(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel = 2,
                                   const SoftShadowSampler* sampler = nullptr, int threads = 0);
// sampler: required when the scene has spherical lights (nullptr -> SoftShadowSampler::gaussian()).
RenderStats renderRayTracing(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel = 2,
                             const SoftShadowSampler* sampler = nullptr);
// Same, into a plain W*H rgb float buffer indexed y*W+x (not flipped).
RenderStats renderToBuffer(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb, int maxLevel = 2,
                           const SoftShadowSampler* sampler = nullptr);
// renderToBuffer keeps its per-level arrays (up to 1 GiB) for the next frame; this hands them back to the C library.
void releaseRenderBuffers();
