/* cgrt.h -- C-ABI of the MI355X-native BVH traversal + ray/triangle path.
 *
 * The reference (mgokbulut/CG-RayTracer) has no FFI / plugin layer: its hot path is the C++ class
 * BoundingVolumeHierarchy (src/bounding_volume_hierarchy.h:15-56) plus the free functions of
 * src/ray_tracing.h:10-20, called one ray at a time from src/main.cpp:115,182,276.  This header is
 * the boundary a maintainer would bind instead; every entry names the reference
 * interface it replaces.  Plain C types only, caller-allocated outputs, no exceptions cross it.
 *
 * Every function that returns int returns 0 on success and a negative CGRT_E_* code on failure;
 * cgrt_last_error() then describes the failure (thread-local string).  There is NO CPU fallback:
 * without a usable HIP device cgrt_scene_create fails with CGRT_E_NO_DEVICE.
 *
 * Threads.  The reference calls BoundingVolumeHierarchy::intersect on one const object from an `omp parallel for`
 * (main.cpp:653-656), so:
 *   - cgrt_intersect_batch, cgrt_trace_primary, cgrt_generate_rays, cgrt_count_* (host pointers) may be called
 *     concurrently on ONE scene from any number of threads: each call runs on a private stream with private device
 *     scratch and pinned staging taken from a per-scene pool and waits for that stream only (no hipMalloc / hipFree /
 *     hipDeviceSynchronize once the pool has grown to the call sizes in use); cgrt_intersect_batch calls of at most 64 rays --
 *     BoundingVolumeHierarchy::intersect, one ray per call -- are COMBINED: concurrent callers append their rays to a shared
 *     pinned ring, one of them launches one kernel for all of them and everybody copies its own hits out (cgrt_set_call_combining);
 *   - the *_device entries only enqueue work on the caller's stream: concurrent too.  (cgrt_trace_primary_device keeps one piece of
 *     per-scene state, the frame hints -- see cgrt_set_frame_hints: a short mutex while the launch is issued, and launches that
 *     arrive on changing streams simply run without hints;)
 *   - cgrt_render* use one per-scene workspace: calls on the same scene are serialised by a mutex inside the library;
 *   - cgrt_set_* are process-wide options (mutex / atomic inside); cgrt_scene_set_walk and cgrt_scene_destroy must not
 *     race with calls on that scene.
 */
#ifndef CGRT_H
#define CGRT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CGRT_OK 0
#define CGRT_E_ARG (-1)       /* bad argument (null pointer, inconsistent counts, index out of range) */
#define CGRT_E_NO_DEVICE (-2) /* no HIP device / device index out of range */
#define CGRT_E_HIP (-3)       /* a HIP runtime call failed */
#define CGRT_E_ALLOC (-4)     /* host or device allocation failed */
#define CGRT_E_LIMIT (-5)     /* scene exceeds a structural limit (BVH deeper than 12 levels) */

#define CGRT_NO_PRIM 0xffffffffu
#define CGRT_DEVICE_NONE (-1) /* cgrt_scene_create: build on the host only, no device upload */

/* Ray: bit-compatible with the reference's `Ray` (framework/include/ray.h:9-13): 28 bytes. */
typedef struct CgrtRay {
    float origin[3];
    float direction[3];
    float t; /* in: current closest t (FLT_MAX for a fresh ray) */
} CgrtRay;

/* Result of BoundingVolumeHierarchy::intersect for one ray (bounding_volume_hierarchy.cpp:850-881).
 *   t           final ray.t (unchanged input t on a miss)
 *   prim_id     last accepted primitive: global triangle index (prefix over Scene::meshes in load
 *               order + index in Mesh::triangles), or ntris + sphere index; CGRT_NO_PRIM on a miss.
 *               The reference has no primitive id (HitInfo = normal + material, ray_tracing.h:4-8);
 *               this is the definition of SURVEY.md section 8(c).
 *   material_id index of the mesh whose Material the reference copies into hitInfo.material
 *               (bvh.cpp:547), -1 if never written (miss, or sphere-only hit: bvh.cpp:878-879).
 *   hit         the bool the reference returns. */
typedef struct CgrtHit {
    float t;
    uint32_t prim_id;
    int32_t material_id;
    uint32_t hit;
} CgrtHit;

/* Trackball camera state (framework/include/trackball.h:48-56; defaults src/main.cpp:730-731). */
typedef struct CgrtCamera {
    float look_at[3];
    float euler[3]; /* radians */
    float distance;
    float fovy;   /* radians, vertical */
    float aspect; /* Window::aspectRatio(), framework/src/window.cpp:334-337 */
} CgrtCamera;

/* Totals of the traversal work for a batch (SURVEY.md section 8(d) algorithmic-bytes definition). */
typedef struct CgrtCounters {
    uint64_t rays;
    uint64_t inner_visits; /* reference inner nodes visited (intersectNonLeaf calls, bvh.cpp:715) */
    uint64_t leaf_visits;  /* reference leaves visited (intersectLeaf calls, bvh.cpp:535) */
    uint64_t tri_tests;    /* triangle records tested on the device */
    uint64_t sub_visits;   /* 4-wide nodes visited: in-leaf accelerators (exact walk) and the fast tree (certified walk) */
    uint64_t cert_boxes;   /* certified walk: path boxes tested by certificates (24 B each) */
    uint64_t fallback_rays; /* certified walk: rays that got no certificate and took the exact walk afterwards */
    uint64_t tree_rays;    /* rays that passed the root gate (intersectDataStructure, bvh.cpp:835-836) and walked the tree */
} CgrtCounters;

typedef struct CgrtScene CgrtScene;

/* Replaces BoundingVolumeHierarchy::BoundingVolumeHierarchy(Scene*) (bvh.cpp:42-76): builds the
 * reference's 12-level median-split tree on the host (same topology, same leaf order), flattens it
 * and uploads it to `device`.  With device == CGRT_DEVICE_NONE nothing is uploaded: the tree can be
 * inspected (cgrt_num_levels, cgrt_get_nodes, cgrt_leaf_prims) but every trace/intersect entry fails with
 * CGRT_E_NO_DEVICE.
 *   pos_nrm   nverts x 6 floats (Vertex{p, n}, mesh.h:12-15), all meshes concatenated
 *   tri       ntris x 3 indices into pos_nrm, meshes concatenated in load order
 *   tri_mesh  ntris mesh indices, non-decreasing
 *   materials nmesh x 8 floats (Material{kd, ks, shininess, transparency}, mesh.h:17-23)
 *   spheres   nspheres x 5 floats {center, radius, material index or -1} (scene.h:36-40), may be NULL */
int cgrt_scene_create(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh,
                      uint32_t ntris, const float* materials, uint32_t nmesh, const float* spheres,
                      uint32_t nspheres, int device, CgrtScene** out);
void cgrt_scene_destroy(CgrtScene* scene);

/* In-leaf accelerator (no counterpart upstream; DESIGN.md "In-leaf accelerator").  The reference scans
 * a leaf's triangles linearly (bvh.cpp:535-553, ~390 per leaf at 800 K triangles).  By default every
 * scene also gets, per reference leaf, a small BVH that lets the kernel skip triangles the ray cannot
 * hit; results are identical with it on or off (that is tested), only the work differs.  Process-wide,
 * applies to scenes created afterwards.  sub_leaf_tris: triangles per accelerator leaf, 0 = default (4). */
int cgrt_set_leaf_accel(int enabled, int sub_leaf_tris);
int cgrt_num_subnodes(const CgrtScene* scene);
/* Scheduling of the fused primary-frame kernel (results are identical; tested): 0 = one wave per 8x8 tile,
 * 1 = persistent waves that pull tiles from per-XCD queues and refill finished lanes.  Process-wide. */
int cgrt_set_primary_mode(int mode);
/* Kernel shape of the certified walk, chosen per launch (no counterpart upstream; DESIGN.md "Latency shapes"; results are
 * identical, tested).  A launch ends when its hardest rays end, and a hard ray's dependent chain runs ~2x faster when it does not
 * share its wave with 63 others, so SHORT RAY LISTS are laid out sparsely:
 *   mode -1 (default) by size: lists of at most 8192 rays -> 4 rays per wave, 16 lanes per ray (four stack entries x four child
 *            boxes per round); of at most max_rays (default 131072) -> one ray per lane, 16 rays per single-wave workgroup (the wave
 *            is in the quad tail, four lanes per ray, from its first step); larger lists and every frame: 64 rays per wave.
 *            Lists sized on the device (cgrt_render's wavefront) choose between 16 and 64 per wave on the device.
 *   mode 0 = 64 rays per wave always (round 2's behaviour), 1 = quad per ray with 16 rays per wave (frames too), 2 = 16 rays per
 *            wave for every list, 3 = 4 rays per wave for every list: for tests and measurements.
 * max_rays = 0 keeps the current threshold.  Process-wide. */
int cgrt_set_kernel_shape(int mode, uint64_t max_rays);
int cgrt_get_kernel_shape(int* mode, uint64_t* max_rays);
/* Frame hints of the primary frame entries (cgrt_trace_primary*, one scene, frames of one shape after another): every wave of a
 * frame measures its own wall time, and the 8x8 tiles whose wave took long are traced differently by the NEXT frame of the same
 * shape -- first (mode 1: the long waves no longer start in the frame's last round) or as four waves of 16 rays (mode 2: for
 * small frames, whose time is the time of their longest wave).  -1 (default) = by frame size (<= 0.8 M rays: 2; a rank's share
 * of <= 1.3 M rays of a frame split over >= 4 ranks: 2; anything else: none -- mode 1 is only ever asked for), 0 = off.  The time from which a wave
 * counts as long starts at 45 us and follows the scene (it rises while more than 2 % of the frame's tiles are listed); a scene
 * whose lists stay empty is traced without hints for 56 of every 64 frames.  Only the order and the layout of the work change: every pixel is traced once, by the same arithmetic
 * (tested bit for bit); the first frames of a shape (hints are set up for a shape that has been traced three times in a row), the
 * instrumented and the multi-device (packed) launches take no hints.  The hint buffers belong to the scene: frames of one scene issued on ONE stream use them; when the caller changes
 * streams the library falls back to unhinted launches for a few frames (the *_device entries stay safe to call from several
 * threads, they are just not accelerated then).  Process-wide. */
int cgrt_set_frame_hints(int mode);
/* Tests: the wall time (s_memrealtime ticks, 100 MHz) from which a 64-ray / a 16-ray wave puts its tile on the hard list; 0 = the
 * default (4500 = 45 us, the floor of the adaptive threshold; a 16-ray wave counts from 5/9 of the threshold in force, the second
 * argument is ignored).  With a few ticks every tile that reaches the tree is "hard" and the lists overflow. */
int cgrt_debug_set_hint_thresholds(unsigned dense_ticks, unsigned sparse_ticks);
/* Diagnostics: the lengths of the scene's three rotating hard lists (synchronises the device). */
int cgrt_debug_hint_counts(CgrtScene* scene, uint32_t* out3);
/* cgrt_render*: 1 (default) = a frame of the same shape (size, rank, depth, light count) as the scene's previous one is issued in one
 * go, its launches sized from what that frame found per level, the counts checked once behind the frame (and the frame drawn again
 * the exact way when a list outgrew its launch); 0 = every frame waits for the device's hit count before it sizes the lists, as the
 * first frame always does.  The pixels are identical (tested); only the host round trips inside the frame differ.  Process-wide. */
int cgrt_set_render_prediction(int enabled);
/* How the scene's last cgrt_render* frame was drawn: 0 = exactly sized, 1 = as predicted, 2 = predicted, a list outgrew its launch,
 * drawn again exactly; -1 for a NULL scene.  (Tests, diagnostics.) */
int cgrt_debug_render_path(const CgrtScene* scene);
/* Certified walk (no counterpart upstream; DESIGN.md "Certified walk").  The exact walk takes every step of the
 * reference's ordered descent (bvh.cpp:572-758) because its culling quirks are part of the result.  A scene may also
 * carry a "fast tree" (a 4-wide tree over the reference LEAVES) and per-leaf box paths: a ray then searches the fast
 * tree with conservative tests, and its answer is kept only if a certificate holds -- unique strict minimum, no
 * origin-on-plane acceptance, and every box the reference tests on its way to that leaf is entered at the final t
 * (the reference's own box arithmetic); any other ray is walked exactly.  Results are identical either way (tested).
 *   cgrt_set_fast_tree: process-wide, for scenes created afterwards: -1 (default) = build it whenever the structures
 *     allow it, except for scenes of fewer than 16 triangles and scenes created with the in-leaf accelerator off;
 *     0 = never; 1 = whenever the structures allow it (finite geometry, tame float planes, every leaf accelerated or small).
 *   cgrt_scene_set_walk: per scene, 1 = certified walk (needs the fast tree: CGRT_E_ARG otherwise), 0 = exact walk only.
 *     Not to be changed while launches of the scene are being issued from other threads.
 *   cgrt_scene_walk: the current setting (1 / 0). */
int cgrt_set_fast_tree(int mode);
int cgrt_scene_set_walk(CgrtScene* scene, int certified);
int cgrt_scene_walk(const CgrtScene* scene);
/* What the host builder decided (also for host-only scenes): out4 = {1 if the fast tree was built, number of reference leaves
 * that hold a triangle whose float plane degenerates (such leaves keep the linear scan and veto the fast tree), 1 if every vertex
 * coordinate is finite, number of reference leaves}. */
int cgrt_scene_build_info(const CgrtScene* scene, uint32_t* out4);

/* BoundingVolumeHierarchy::numLevels() (bvh.cpp:214-224). */
int cgrt_num_levels(const CgrtScene* scene);
/* Tree introspection for builder-parity tests (the reference keeps std::vector<Node>, bvh.h:6-13).
 * meta: nnodes x 5 int32 {is_leaf, level, child0, child1, ntris}; boxes: nnodes x 6 floats. */
int cgrt_num_nodes(const CgrtScene* scene);
int cgrt_get_nodes(const CgrtScene* scene, int32_t* meta, float* boxes);
/* prim ids of leaf `node` in the order intersectLeaf scans them (bvh.cpp:538-551). Returns count. */
int64_t cgrt_leaf_prims(const CgrtScene* scene, int node, uint32_t* out, uint32_t cap);
double cgrt_build_seconds(const CgrtScene* scene);
uint64_t cgrt_device_bytes(const CgrtScene* scene);

/* Batched BoundingVolumeHierarchy::intersect(Ray&, HitInfo&) (bvh.cpp:850-881): n independent rays.
 * normals (optional, n x 3) receives hitInfo.normal for rays that hit (left untouched otherwise, as
 * the reference leaves HitInfo untouched on a miss).  Host pointers; synchronous. */
int cgrt_intersect_batch(CgrtScene* scene, const CgrtRay* rays, uint64_t n, CgrtHit* hits, float* normals);
/* Call combining for small cgrt_intersect_batch calls (see "Threads" above; DESIGN.md "Per-ray boundary"): 1 = on (default),
 * 0 = every call launches for itself (round 2's behaviour).  Results are identical.  Process-wide. */
int cgrt_set_call_combining(int enabled);
/* Diagnostic: out5 = {combined generations launched on this scene, rays in them, rays of the largest one, nanoseconds their
 * leaders spent between closing a generation and holding its results (waiting for the joiners' rays + launch + kernel + stream
 * wait), the part of that up to the return of the launch call}. */
int cgrt_debug_combiner_stats(const CgrtScene* scene, uint64_t* out5);
/* intersectRayWithShape(const Mesh&, Ray&, HitInfo&) (ray_tracing.cpp:202-213) for n rays: every triangle is tested, no
 * tree -- the reference's ground truth over all triangles (its BVH misses hits this loop finds: SURVEY.md F4).
 *   mesh >= 0: that mesh only (index into the scene's meshes); material_id stays -1, the loop never writes hitInfo.material.
 *   mesh <  0: every mesh in load order, then the spheres, material written: the pre-BVH body of
 *              BoundingVolumeHierarchy::intersect (bvh.cpp:854-868, commented out upstream).
 * Same outputs and conventions as cgrt_intersect_batch; O(n * ntris) work: a validation path, not a fast one. */
int cgrt_intersect_brute_batch(CgrtScene* scene, const CgrtRay* rays, uint64_t n, int mesh, CgrtHit* hits, float* normals);
/* Same, all pointers are DEVICE pointers on the scene's device; asynchronous on `stream`
 * (a hipStream_t, NULL = default stream). */
int cgrt_intersect_batch_device(CgrtScene* scene, const CgrtRay* d_rays, uint64_t n, CgrtHit* d_hits,
                                float* d_normals, void* stream);

/* Primary frame: fuses renderRayTracing's ray generation (main.cpp:691-694 + Trackball::generateRay,
 * trackball.cpp:92-103) with intersect; no rays are uploaded.  Pixel (x, y), 0 <= x < W, 0 <= y < H,
 * result index y*W + x (not y-flipped).  The rectangle [x0,x1) x [y0,y1) is cut into 64x64-pixel super-tiles
 * (row-major from (x0, y0)); only pixels whose super-tile index % nranks == rank are traced and written
 * (image tiling across GPUs). */
int cgrt_trace_primary(CgrtScene* scene, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1,
                       int rank, int nranks, CgrtHit* hits, float* normals);
int cgrt_trace_primary_device(CgrtScene* scene, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1,
                              int y1, int rank, int nranks, CgrtHit* d_hits, float* d_normals, void* stream);
/* The rays the fused kernel generates, written out for "same rays" parity checks (row-major over the
 * rectangle). Host pointer. */
int cgrt_generate_rays(CgrtScene* scene, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1,
                       CgrtRay* rays);

/* renderRayTracing / getFinalColor (src/main.cpp:298-310, :648-720) for a whole frame, entirely on the device: primary
 * rays, then per recursion level the shadow rays of every hit (pointInShadow, :104-135), the Phong terms (:61-98,
 * :219-232) and the mirror rays (shade, :241-264).  max_level = 2 is the reference (`level >= 2` -> black, :267).
 * lights: nlights x 6 floats {position, color} (PointLight, scene.h:42-45); rgb: W*H*3 floats, index y*W+x, not
 * y-flipped and not clamped (Screen::setPixel / writeBitmapToFile do that, screen.cpp:30-49).  Host pointers. */
/* The scene keeps the device workspace of these calls between frames (about 0.3 KB per pixel at 1920x1080 with one light
 * and max_level 2; it grows with the frame, the lights and the depth) and releases it in cgrt_scene_destroy. */
typedef struct CgrtRenderStats {
    uint64_t primary_rays, shadow_rays, reflection_rays; /* rays that exist upstream */
    int32_t levels;                                      /* recursion levels actually evaluated */
    float device_ms;                                     /* HIP-event time of all kernels of the frame */
    uint64_t soft_shadow_rays;                           /* samples towards spherical lights (cgrt_render_soft) */
} CgrtRenderStats;
int cgrt_render(CgrtScene* scene, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, int max_level,
                float* rgb, CgrtRenderStats* stats);

/* The same frame with the spherical lights of `shading` (src/main.cpp:168-218): per hit and light, `samples` shadow rays
 * (200 upstream, :176) towards position + radius * randomUnitVector(); the light's diffuse and specular terms are scaled
 * by the fraction that reaches it (!intersect || ray.t > lightT, :183-199); spherical lights are accumulated before the
 * point lights (:168, :219).  Upstream draws randomUnitVector() from std::random_device (:46-59), which cannot be
 * reproduced by anyone; here the draws are DATA: `unit_vectors` holds nunits x 3 floats (normalised gaussian triples
 * give upstream's distribution) and sample `smp` of spherical light `l` at pixel p = y*W+x, recursion level `lv`, uses
 * entry  mix(mix(mix(mix(seed ^ 0x9e3779b9) ^ p) ^ (lv * 0x01000193 + l)) ^ smp) % nunits,  mix = murmur3's 32-bit
 * finaliser (h ^= h>>16; h *= 0x85ebca6b; h ^= h>>13; h *= 0xc2b2ae35; h ^= h>>16).  With the same table and seed the
 * frame is reproducible to the RGB parity bar; with another table it agrees statistically.
 * closest_hit = 0: a sample ray stops at the first leaf that accepts a triangle (the count needs the hit flag only and
 * that flag is the reference's); 1: full closest-hit walk, for checking that claim. */
typedef struct CgrtSoftShadows {
    const float* spherical;    /* nspherical x 7 {position, radius, color} (SphericalLight, scene.h:47-51) */
    const float* unit_vectors; /* nunits x 3 */
    uint32_t nspherical, samples, nunits, seed;
    int32_t closest_hit;
} CgrtSoftShadows;
int cgrt_render_soft(CgrtScene* scene, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights,
                     const CgrtSoftShadows* soft, int max_level, float* rgb, CgrtRenderStats* stats);
/* cgrt_render_soft without the last copy: *rgb receives a pointer to the frame in PINNED host memory owned by the scene (W*H*3
 * floats, index y*W+x), into which the device frame was downloaded with one asynchronous copy; it stays valid until the next
 * cgrt_render* call on this scene or cgrt_scene_destroy.  For callers that convert the frame anyway (Screen::setPixel's flip,
 * screen.cpp:30-36): a 1920x1080 float frame is 25 MB, and copying it once more costs several device frames.  soft may be NULL. */
int cgrt_render_mapped(CgrtScene* scene, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights,
                       const CgrtSoftShadows* soft, int max_level, const float** rgb, CgrtRenderStats* stats);
/* Image tiling across GPUs for whole frames: this call traces and shades only the pixels whose
 * 64x64 super-tile index % nranks == rank (the ownership rule of cgrt_trace_primary); every secondary ray of a pixel
 * stays on the GPU that owns the pixel, so ranks exchange nothing.  rgb of pixels owned by other ranks is left as the
 * caller passed it; the ranks' frames merge by ownership into exactly the single-rank frame.  soft may be NULL. */
int cgrt_render_rank(CgrtScene* scene, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights,
                     const CgrtSoftShadows* soft, int max_level, int rank, int nranks, float* rgb, CgrtRenderStats* stats);

/* cgrt_render with the instrumented kernels (a separate, never timed frame): work3[0] / [1] / [2] receive the traversal work of
 * the primary rays, of all shadow lists and of all mirror lists of the frame (SURVEY.md section 8(d) algorithmic bytes). */
int cgrt_render_counted(CgrtScene* scene, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights,
                        int max_level, float* rgb, CgrtRenderStats* stats, CgrtCounters* work3);

/* One caller, N devices, ONE framebuffer (the reference's renderRayTracing fills one Screen,
 * main.cpp:648-720, its only parallel construct being rows of that frame, :653-656).  scenes[i] is a replica of the scene
 * created on its own device (cgrt_scene_create with different `device` arguments; the same device may repeat, every
 * replica must be a distinct handle).  Replica i traces / renders the 64x64 super-tiles i % nscenes; nothing is exchanged
 * between devices.  trace: all launches are issued first, each on a private stream of its replica; every device downloads
 * its pixels with one asynchronous copy and a host thread per replica scatters them into `hits` / `normals` (normals
 * only where hit == 1).  render: one host thread per replica runs cgrt_render_rank's wavefront into a frame of its own
 * and the owned pixels are merged into `rgb`; stats: ray counts summed, levels / device_ms = maximum over replicas.
 * The bytes written equal those of the single-device entries (tested). */
typedef struct CgrtMultiStats {
    int32_t replicas;
    float kernel_ms_max;   /* slowest replica's traversal kernel (HIP events on its stream) */
    float download_ms_max; /* slowest replica's device -> pinned host copy */
    double wall_ms;        /* first launch to last pixel scattered, host clock */
    uint64_t rays[64];     /* pixels traced by each replica */
} CgrtMultiStats;
int cgrt_trace_primary_multi(CgrtScene* const* scenes, int nscenes, const CgrtCamera* cam, int W, int H, CgrtHit* hits,
                             float* normals, CgrtMultiStats* stats);
int cgrt_render_multi(CgrtScene* const* scenes, int nscenes, const CgrtCamera* cam, int W, int H, const float* lights,
                      uint32_t nlights, const CgrtSoftShadows* soft, int max_level, float* rgb, CgrtRenderStats* stats);

/* Work counters of the same traversal (separate instrumented launch; not part of any timed region). */
int cgrt_count_primary(CgrtScene* scene, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1,
                       int rank, int nranks, CgrtCounters* out);
int cgrt_count_batch(CgrtScene* scene, const CgrtRay* rays, uint64_t n, CgrtCounters* out);
/* Diagnostic (not part of the reference surface): one stamped + instrumented full-frame launch; out holds 16 u64
 * per 8x8 tile (= wave): s_memtime start, end; s_memrealtime start, end; lane-summed inner/leaf/tri/sub steps;
 * wave-level iterations of the inner, sub-node and triangle bodies; per-lane maxima of inner/leaf/tri/sub; active
 * lanes.  One record per launched wave: cap_waves >= 4 * 16 * 8 * ceil(ceil(W/64)*ceil(H/64) / 8). */
int cgrt_debug_wave_times(CgrtScene* scene, const CgrtCamera* cam, int W, int H, uint64_t* out, uint64_t cap_waves);
/* Diagnostic: `repeats` launches of a kernel that reads nrecords scattered 64-byte records (one per lane, never twice)
 * from a zeroed table -- a known HBM byte count in the traversal kernels' access shape, for calibrating rocprofv3's
 * FETCH_SIZE on gfx950 (tools/calibrate_fetch_size.sh). */
int cgrt_debug_gather_calibration(int device, uint64_t nrecords, int repeats);
/* Diagnostic: the kernels' 4-operation exact division (trace_kernels.hip fdiv4) against IEEE a[i] / d[i] on the
 * device; mismatches receives the count, first_bad {a, d, got, expected} of the first one. */
int cgrt_debug_fastdiv_check(int device, const float* a, const float* d, uint64_t n, uint64_t* mismatches, float* first_bad);
/* Diagnostic: validates every reference of the scene's record arrays on the host (tree child references, leaf references
 * in both encodings, accelerator nodes and runs, each triangle reachable exactly once).  Works on host-only scenes. */
int cgrt_debug_check_layout(CgrtScene* scene);
/* Diagnostic: FNV-1a hash over every array the device reads (node packets, accelerator + fast-tree nodes, triangle records, leaf
 * table, vertex normals, certificate paths, leaf of every record, roots): two builds of the same scene must agree, whatever the
 * number of builder threads (cgrt_set_build_threads: 0 = hardware concurrency, at most 16 are used; process-wide, for scenes
 * created afterwards).  Works on host-only scenes. */
int cgrt_debug_layout_hash(const CgrtScene* scene, uint64_t* out);
int cgrt_set_build_threads(int threads);
/* Bytes of one inner-node record / one triangle record / one in-leaf accelerator node / one result. */
void cgrt_record_sizes(uint32_t* node_bytes, uint32_t* tri_bytes, uint32_t* sub_bytes, uint32_t* hit_bytes);

/* Element-wise device versions of the free functions of src/ray_tracing.h:10-20 and startsInBox
 * (bvh.cpp:647-661); element i of every array belongs to call i.  Host pointers; synchronous.
 *   cgrt_ray_triangle_batch  intersectRayWithTriangle (ray_tracing.cpp:86-114)
 *       tri: n x 18 floats {v0 v1 v2 n1 n2 n3}; t_io: n floats in/out; hit: n bytes; normals n x 3 (written on hit)
 *   cgrt_ray_plane_batch     intersectRayWithPlane (ray_tracing.cpp:40-72); plane: n x 4 {D, normal}
 *   cgrt_ray_box_batch       intersectRayWithShape(AxisAlignedBox) (ray_tracing.cpp:162-200) and startsInBox;
 *       box: n x 6 {lower, upper}; inside: n bytes (optional)
 *   cgrt_ray_sphere_batch    intersectRayWithShape(Sphere) (ray_tracing.cpp:118-158); sphere: n x 4 {center, radius}
 *   cgrt_triangle_plane_batch trianglePlane (ray_tracing.cpp:74-82); tri: n x 9 -> plane n x 4 {D, normal}
 *   cgrt_point_in_triangle_batch pointInTriangle (ray_tracing.cpp:23-38); in: n x 15 {v0 v1 v2 n p} */
int cgrt_ray_triangle_batch(int device, const float* tri, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit,
                            float* normals);
int cgrt_ray_plane_batch(int device, const float* plane, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit);
int cgrt_ray_box_batch(int device, const float* box, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit,
                       uint8_t* inside);
int cgrt_ray_sphere_batch(int device, const float* sphere, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit,
                          float* normals);
int cgrt_triangle_plane_batch(int device, const float* tri, uint64_t n, float* plane);
int cgrt_point_in_triangle_batch(int device, const float* in, uint64_t n, uint8_t* out);

int cgrt_device_count(void);
const char* cgrt_last_error(void);
const char* cgrt_version(void);
/* First 16 hex digits of the sha256 over the library's sources (every .hip, .cpp and .h file of csrc + this header, in name order) at build time:
 * ties a committed profile to the kernels it was measured on (bench.py nulls roofline.traffic / roofline.measured when it differs). */
const char* cgrt_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* CGRT_H */
