// renderRayTracing / getFinalColor / trace / shade / shading / pointInShadow (src/main.cpp:61-310, :648-720) as a
// host-driven WAVEFRONT: per recursion level, all shadow rays and all reflection rays of the frame go through the
// GPU path in batches (BoundingVolumeHierarchy::intersectBatch); the Phong arithmetic runs on the host in the
// reference's expression order.  Point lights only (spherical lights are random upstream: main.cpp:46-59).
#pragma once
#include "bounding_volume_hierarchy.h"
#include "screen.h"
#include "trackball.h"

struct RenderStats {
    uint64_t primary = 0, shadow = 0, reflection = 0;
    double seconds_total = 0, seconds_device = 0;
};

// maxLevel = 2 reproduces main.cpp:267 (`level >= 2` -> black): primary + one mirror bounce.
RenderStats renderRayTracing(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel = 2);
// Same, into a plain W*H rgb float buffer indexed y*W+x (not flipped).
RenderStats renderToBuffer(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb, int maxLevel = 2);
