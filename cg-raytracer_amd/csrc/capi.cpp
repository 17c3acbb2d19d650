// capi.cpp -- implementation of the C-ABI declared in include/cgrt.h.
// Host C++ over the HIP runtime; no torch types, no CPU traversal path: every intersect/trace entry
// launches the gfx950 kernels of trace_kernels.hip and fails loudly when no device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/cgrt.h"
#include "bvh_builder.h"
#include "cgrt_layout.h"
#include "cgrt_math.h"
#include "trace_kernels.h"

using namespace cgrt;

static_assert(sizeof(CgrtRay) == 28, "CgrtRay must match the reference Ray (ray.h:9-13)");
static_assert(sizeof(CgrtHit) == sizeof(CgrtHitDev), "CgrtHit layout");

namespace {

thread_local std::string g_err;
// Process-wide options (cgrt_set_leaf_accel / cgrt_set_fast_tree / cgrt_set_primary_mode).  The build options are copied
// under a mutex when a scene is created; the primary mode is read atomically by every launch.
std::mutex g_options_mutex;
BuildOptions g_build_options;
std::atomic<int> g_call_combining{1};  // cgrt_set_call_combining
std::atomic<int> g_render_predict{1};  // cgrt_set_render_prediction
std::atomic<int> g_frame_hints{-1};    // cgrt_set_frame_hints: -1 auto, 0 off, 1 hard tiles first, 2 hard tiles 16 rays per wave
std::atomic<unsigned> g_hint_thr_dense{0}, g_hint_thr_sparse{0};  // cgrt_debug_set_hint_thresholds (0: the defaults)
std::atomic<int> g_primary_mode{0};  // 0 = one wave per tile, 1 = persistent waves with lane refill

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return CGRT_E_HIP;
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

// RAII device buffer for the host-pointer convenience entries.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(CGRT_E_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= n) return fail(CGRT_E_NO_DEVICE, "device index " + std::to_string(device) + " out of range");
    HIP_TRY(hipSetDevice(device));
    return CGRT_OK;
}

// Ray-independent part of Trackball::generateRay / position (trackball.cpp:70-73, :92-103),
// glm::qua(eulerAngles) as in glm 0.9.9.8 type_quat.inl.
CameraDev make_camera(const CgrtCamera& c) {
    const float hx = c.euler[0] * 0.5f, hy = c.euler[1] * 0.5f, hz = c.euler[2] * 0.5f;
    const float cx = std::cos(hx), cy = std::cos(hy), cz = std::cos(hz);
    const float sx = std::sin(hx), sy = std::sin(hy), sz = std::sin(hz);
    Q4 q;
    q.w = cx * cy * cz + sx * sy * sz;
    q.x = sx * cy * cz - cx * sy * sz;
    q.y = cx * sy * cz + sx * cy * sz;
    q.z = cx * cy * sz - sx * sy * cz;
    const F3 p = add(f3(c.look_at[0], c.look_at[1], c.look_at[2]), quat_rotate(q, f3(0.0f, 0.0f, -c.distance)));
    CameraDev d;
    d.pos[0] = p.x;
    d.pos[1] = p.y;
    d.pos[2] = p.z;
    d.q[0] = q.w;
    d.q[1] = q.x;
    d.q[2] = q.y;
    d.q[3] = q.z;
    d.half_h = std::tan(c.fovy / 2.0f);
    d.half_w = c.aspect * d.half_h;
    return d;
}

bool make_frame(int W, int H, int x0, int y0, int x1, int y1, int rank, int nranks, int block, FrameDev& F) {
    if (W <= 0 || H <= 0 || x0 < 0 || y0 < 0 || x1 > W || y1 > H || x0 > x1 || y0 > y1 || nranks <= 0 || rank < 0 || rank >= nranks)
        return false;
    F.W = W;
    F.H = H;
    F.x0 = x0;
    F.y0 = y0;
    F.x1 = x1;
    F.y1 = y1;
    F.tiles_x = (x1 - x0 + 7) / 8;
    F.tiles_y = (y1 - y0 + 7) / 8;
    F.st_x = (F.tiles_x + ST_TILES - 1) / ST_TILES;
    F.st_y = (F.tiles_y + ST_TILES - 1) / ST_TILES;
    F.rank = rank;
    F.nranks = nranks;
    const uint64_t nst = (uint64_t)F.st_x * (uint64_t)F.st_y;
    F.nst_rank = (uint32_t)((nst + (uint64_t)nranks - 1 - (uint64_t)rank) / (uint64_t)nranks);
    F.block = block;
    F.nblocks = ((F.nst_rank + 7u) / 8u) * 8u * (64u / ((uint32_t)block / 64u));
    F.packed = 0;
    F.hint = nullptr;
    F.hint_blocks = F.hint_rgen = F.hint_wgen = 0;
    return true;
}

// pixels of the rectangle that belong to F.rank (its super-tiles, clipped)
unsigned long long owned_pixels(const FrameDev& F) {
    unsigned long long n = 0;
    const uint64_t nst = (uint64_t)F.st_x * (uint64_t)F.st_y;
    for (uint64_t i = (uint64_t)F.rank; i < nst; i += (uint64_t)F.nranks) {
        const int sx = (int)(i % (uint64_t)F.st_x), sy = (int)(i / (uint64_t)F.st_x);
        const int w = std::min(64, (F.x1 - F.x0) - 64 * sx), h = std::min(64, (F.y1 - F.y0) - 64 * sy);
        n += (unsigned long long)w * (unsigned long long)h;
    }
    return n;
}

}  // namespace

struct CgrtScene {
    int device = 0;
    BuiltBvh bvh;
    uint32_t ntris = 0;
    SceneDev dev = [] {
        SceneDev d{};
        d.fast_root = REF_NONE;
        d.root_ref = REF_NONE;
        return d;
    }();
    void* d_records = nullptr;  // [packets | subnodes | tris], 64 B each
    void* d_leaves = nullptr;
    void* d_tri_normals = nullptr;
    void* d_spheres = nullptr;
    void* d_materials = nullptr;  // nmesh x 8 floats, for the shading wavefront
    void* d_tri_leaf = nullptr;   // certified walk: leaf of every record, per-leaf box paths (SceneDev::tri_leaf, paths)
    void* d_paths = nullptr;
    uint32_t fast_root = REF_NONE;  // the scene's fast tree (REF_NONE: none); dev.fast_root is this or REF_NONE by cgrt_scene_set_walk
    uint32_t nmesh = 0;
    unsigned int* d_queues = nullptr;  // ring of 8 queue blocks (CGRT_QUEUE_BLOCK_WORDS u32 each) for the persistent kernel, reset by every launch
    // the persistent kernel's launches take the queue blocks in turn; a block is handed to a new launch only behind the
    // launch that used it last (an event per block), so any number of frames may be in flight on any streams
    std::mutex queue_mutex;
    unsigned launch_seq = 0;
    // Frame hints (cgrt_layout.h HintDev; attach_hints below): the hard-tile lists a primary frame leaves for the next frame of
    // the same shape.  Guarded by hints_mutex while a launch is being issued; the buffers themselves are only touched by kernels.
    struct FrameHints {
        std::mutex mu;
        bool ready = false;           // buffers allocated for `key`
        int key[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // W, H, x0, y0, x1, y1, rank, nranks
        int wanted[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // the shape (+ mode, threshold) of the last launches that asked for other buffers
        int wanted_count = 0;         // ... and how many in a row did
        int per_tile = 0;             // 1 / 4 (the policy the buffers were made for)
        unsigned thr[2] = {0, 0};     // the thresholds in the device structs
        uint32_t cap = 0;
        void* mem = nullptr;          // 3 x {flag[ntiles], list[cap], count} + 3 HintDev
        size_t mem_bytes = 0;
        HintDev* phase[3] = {nullptr, nullptr, nullptr};  // device addresses
        HintDev phase_host[3];        // what they hold
        uint32_t* mailbox = nullptr;  // 64 pinned, device-mapped bytes: {generation, length of the list that frame read, threshold}
        uint32_t* mailbox_dev = nullptr;
        uint32_t seen_gen = 0;        // the last mailbox generation the host has looked at
        uint32_t last_listed = 0;     // ... and what it said
        int empty_streak = 0;         // hinted frames in a row whose list was empty
        int dormant = 0;              // frames still to run plain because of that
        uint64_t seq = 0;             // frames issued with these buffers
        bool have_prev = false;       // the set this frame would read was written by frame seq - 1
        hipStream_t last_stream = nullptr;
        bool last_stream_valid = false;
        hipStream_t hint_stream = nullptr;  // the stream of the last launch that used the buffers
        bool hint_stream_valid = false;
        int cooldown = 0;             // frames to run without hints after the caller changed streams
    } hints;
    hipEvent_t queue_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // Host-pointer entries (cgrt_intersect_batch, cgrt_trace_primary, cgrt_count_*, ...) run on "call lanes": a private
    // stream + device scratch + pinned staging + a counter block, taken from this pool for the duration of one call and
    // kept afterwards.  Concurrent callers (the reference calls intersect from an omp parallel for, main.cpp:653-656) get
    // different lanes; no call allocates, frees or synchronises the device once its lane has grown to the call's size.
    struct CallLane {
        hipStream_t stream = nullptr;
        struct Buf {
            void* p = nullptr;
            size_t cap = 0;
        } dev[3], pin[3];  // rays / hits / normals
        void* bounce[2] = {nullptr, nullptr};  // pinned halves of the large-transfer pipeline (lane_upload / lane_download), made on first use
        hipEvent_t bounce_ev[2] = {nullptr, nullptr};
        unsigned long long* d_counters = nullptr;
    };
    std::mutex lanes_mutex;
    std::vector<CallLane*> lanes_free, lanes_all;
    // Call combining (cgrt_intersect_batch with a handful of rays, i.e. BoundingVolumeHierarchy::intersect as the reference's
    // `omp parallel for` issues it, main.cpp:653-656: one ray per call from many threads at once).  A launch per ray costs the
    // GPU round trip per RAY; here concurrent callers append their rays to the open GENERATION of one of two pinned, device-mapped
    // rings; the first caller of a generation is its leader: it closes the generation, launches ONE kernel over all its rays
    // (the kernel reads the rays from and writes the hits to host memory directly: no copy commands), waits for that stream and
    // publishes the results; the other callers wait on the generation's state and copy their own hits out.  While a generation
    // is on the GPU the next one fills up, so the batch size adapts to the load.  Nothing stays resident on the device.
#ifndef CGRT_COMBINE_RINGS
#define CGRT_COMBINE_RINGS 16  // most generations that can be open / in flight at a time; `nrings` of them are used
#endif
    struct Combiner {
        static const int NRINGS = CGRT_COMBINE_RINGS;
        int nrings = 8;  // set before the rings are made (combined_intersect; CGRT_COMBINE_NRINGS)
        static const uint32_t CAP = 32768;       // ray slots per ring
        static const uint32_t MAX_N = 64;        // calls with more rays than this take the direct path
        static const int MAX_CALLERS = 256;      // callers inside the entry at a time (more take the direct path)
        static const uint32_t JOIN_MAX = CAP - MAX_N * MAX_CALLERS;  // a generation is joined while it holds at most this many rays:
                                                                     // MAX_CALLERS joins of MAX_N rays in flight cannot overflow it
        enum State : uint64_t { FREE = 0, OPEN = 1, RUNNING = 2, DONE = 3 };
        // One 64-bit word per ring says everything a caller needs, so joining is ONE atomic add, leader election and closing single
        // compare-and-swaps, and nobody takes a lock on the way in (with a mutex 64 callers formed a convoy: 139 K calls/s):
        //   bits 0..1 state | 2..15 callers that joined | 16..31 rays appended | 32..63 generation number
        // An add that arrives after the generation was closed lands on a RUNNING / DONE / FREE word: the adder sees the old state in
        // the value it gets back and tries again elsewhere; the stray counts are overwritten by the next transition (the leader
        // keeps its own copy of the counts it closed with), and the caller limit keeps them inside their bit fields.
        static uint64_t pack(uint64_t st, uint64_t joined, uint64_t count, uint64_t gen) { return st | (joined << 2) | (count << 16) | (gen << 32); }
        static uint64_t st_of(uint64_t w) { return w & 3u; }
        static uint32_t joined_of(uint64_t w) { return (uint32_t)((w >> 2) & 0x3fffu); }
        static uint32_t count_of(uint64_t w) { return (uint32_t)((w >> 16) & 0xffffu); }
        struct alignas(64) Ring {
            std::atomic<uint64_t> word{0};     // FREE, generation 0
            char pad0[56];
            // what the WAITERS of a generation spin on -- a line of its own, written once per generation: spinning on `word`
            // made every join fight 60 readers for the line (64 callers: 0.46 M calls/s)
            std::atomic<uint32_t> done_gen{0}; // generations of this ring whose results are published
            char pad1[60];
            std::atomic<uint32_t> copied{0};   // joiners whose rays are in the ring (the leader launches when copied == joined)
            std::atomic<uint32_t> readers{0};  // callers that still have to copy their results out (the last one frees the ring)
            char pad2[56];
            void* host = nullptr;  // pinned + mapped: [CgrtRay x CAP | CgrtHit x CAP | normals 3 x CAP]
            void* dev = nullptr;   // the same memory as the device sees it
            hipStream_t stream = nullptr;
            int rc = 0;            // the leader's status for the whole generation (written before done_gen is published)
            std::string err;
        } ring[CGRT_COMBINE_RINGS];
        std::mutex init_mu;
        // Callers that SLEEP (futex) instead of spinning -- taken when more callers are inside than the process has CPUs (see
        // combined_intersect): `epoch` counts rings set free (what callers without a ring wait for), the sleeper counts tell the
        // thread that publishes whether a wake-up call is needed at all.
        alignas(64) std::atomic<uint32_t> epoch{0};
        std::atomic<int> epoch_sleepers{0}, done_sleepers{0};
        alignas(64) std::atomic<int> inside{0};   // callers currently inside the combining entry
        alignas(64) std::atomic<int> ready{0};    // 0 = rings not allocated yet, 1 = usable, -1 = allocation failed (direct path for good)
        // diagnostics (cgrt_debug_combiner_stats): generations launched, rays in them, the largest generation, nanoseconds the
        // leaders spent from closing a generation to its results (launch + kernel + stream wait)
        std::atomic<uint64_t> n_gen{0}, n_rays{0}, max_gen{0}, ns_gpu{0}, ns_launch{0};  // ns_launch: the part of ns_gpu spent issuing the launch
    } comb;
    std::mutex render_mutex;  // cgrt_render* share the workspace below: one frame per scene at a time
    unsigned persistent_blocks = 1024;  // 4 workgroups per CU
    // Device workspace of cgrt_render*: kept between frames (a frame of the same shape then allocates nothing; hipMalloc and
    // hipFree of ~20 buffers cost more than the frame itself), grown on demand, released with the scene.
    struct WorkSlot {
        void* p = nullptr;
        size_t cap = 0;
    } work[32];  // slots 0..29 are in use (render_impl)
    // pinned host staging of cgrt_render*'s frame (grown on demand, guarded by render_mutex): the device frame comes down with ONE
    // asynchronous copy at PCIe speed; cgrt_render_mapped hands this memory to the caller instead of copying it once more
    void* pin_frame = nullptr;
    size_t pin_frame_cap = 0;
    // streams and events of cgrt_render* (created once per scene, guarded by render_mutex: creating and destroying a stream and
    // five events per frame cost more host time than the Cornell frame's device time)
    struct RenderAux {
        hipStream_t s = nullptr, copy = nullptr;  // second traversal stream; read-backs that must not wait for queued kernels
        hipEvent_t spawned = nullptr, traced = nullptr, e0 = nullptr, e1 = nullptr, primary_done = nullptr;
        uint32_t* pin_counts = nullptr;  // 64 pinned bytes for counter read-backs
        SpawnDev spawn_host{};           // what the workspace's SpawnDev (fused level-0 spawn of predicted frames) holds
        bool spawn_valid = false;
    } raux;
    // What the previous cgrt_render* frame of this shape found, per level (entries of the level's compact list): the next frame's
    // launches are sized from it and issued WITHOUT waiting for the device to say how many primary rays hit (render_impl).
    struct RenderPred {
        bool valid = false;
        int W = 0, H = 0, rank = 0, nranks = 0, max_level = 0;
        unsigned L = 0;
        std::vector<uint32_t> counts;
        int last_path = 0;  // how the last frame was drawn: 0 exact, 1 as predicted, 2 predicted, found too small, drawn again exactly
    } rpred;
    uint64_t device_bytes = 0;
    ~CgrtScene() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        for (void* p : {d_records, d_leaves, d_tri_normals, d_spheres, d_materials, d_tri_leaf, d_paths, (void*)d_queues, hints.mem})
            if (p) (void)hipFree(p);
        if (hints.mailbox) (void)hipHostFree(hints.mailbox);
        if (pin_frame) (void)hipHostFree(pin_frame);
        for (hipEvent_t e : {raux.spawned, raux.traced, raux.e0, raux.e1, raux.primary_done})
            if (e) (void)hipEventDestroy(e);
        for (hipStream_t st : {raux.s, raux.copy})
            if (st) (void)hipStreamDestroy(st);
        if (raux.pin_counts) (void)hipHostFree(raux.pin_counts);
        for (auto& r : comb.ring) {
            if (r.host) (void)hipHostFree(r.host);
            if (r.stream) (void)hipStreamDestroy(r.stream);
        }
        for (CallLane* L : lanes_all) {
            for (auto& b : L->dev)
                if (b.p) (void)hipFree(b.p);
            for (auto& b : L->pin)
                if (b.p) (void)hipHostFree(b.p);
            for (void* b : L->bounce)
                if (b) (void)hipHostFree(b);
            for (hipEvent_t e : L->bounce_ev)
                if (e) (void)hipEventDestroy(e);
            if (L->d_counters) (void)hipFree(L->d_counters);
            if (L->stream) (void)hipStreamDestroy(L->stream);
            delete L;
        }
        for (WorkSlot& w : work)
            if (w.p) (void)hipFree(w.p);
        for (hipEvent_t e : queue_done)
            if (e) (void)hipEventDestroy(e);
    }
};

struct WsBuf {  // a slot of the scene's workspace, with DevBuf's interface
    CgrtScene* sc;
    int slot;
    void* p = nullptr;
    hipError_t alloc(size_t bytes) {
        CgrtScene::WorkSlot& w = sc->work[slot];
        if (w.cap < bytes || !w.p) {
            if (w.p) (void)hipFree(w.p);
            w.p = nullptr;
            w.cap = 0;
            const hipError_t e = hipMalloc(&w.p, bytes ? bytes : 1);
            if (e != hipSuccess) return e;
            w.cap = bytes ? bytes : 1;
        }
        p = w.p;
        return hipSuccess;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

namespace {
void parallel_copy(void* dst, const void* src, size_t bytes);  // (below: large host copies on a few threads)
// Host -> device for the scene's arrays (113 MB for the 800 K-triangle bench scene): through two alternating 8 MB pinned buffers kept
// for the process, the host copying piece k + 1 on a few threads while the DMA engine moves piece k.  Handing the runtime the
// std::vector's pageable memory took 60-90 ms of the 0.41 s a scene took to create (profiles/r3_build_times.txt).
hipError_t staged_h2d(void* dst, const void* src, size_t bytes) {
    const size_t piece_bytes = 8u << 20;
    if (bytes <= (1u << 20)) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    static std::mutex mu;
    static void* pin[2] = {nullptr, nullptr};
    static hipEvent_t ev[2] = {nullptr, nullptr};
    std::lock_guard<std::mutex> lk(mu);
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        if (!pin[b]) e = hipHostMalloc(&pin[b], piece_bytes, hipHostMallocDefault);
        if (e == hipSuccess && !ev[b]) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) {  // (no pinned memory to be had: the plain copy still works)
        (void)hipGetLastError();
        return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    }
    size_t piece = 0;
    for (size_t off = 0; off < bytes; off += piece_bytes, piece++) {
        const int b = (int)(piece & 1);
        const size_t m = std::min(piece_bytes, bytes - off);
        if (piece >= 2 && (e = hipEventSynchronize(ev[b])) != hipSuccess) return e;
        parallel_copy(pin[b], static_cast<const char*>(src) + off, m);
        if ((e = hipMemcpyAsync(static_cast<char*>(dst) + off, pin[b], m, hipMemcpyHostToDevice, nullptr)) != hipSuccess) return e;
        if ((e = hipEventRecord(ev[b], nullptr)) != hipSuccess) return e;
    }
    return hipStreamSynchronize(nullptr);
}
template <class T>
int upload(const std::vector<T>& v, void** dptr, uint64_t& total) {
    const size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    if (bytes) HIP_TRY(staged_h2d(*dptr, v.data(), bytes));
    total += bytes;
    return CGRT_OK;
}
}  // namespace

extern "C" {

const char* cgrt_last_error(void) { return g_err.c_str(); }
const char* cgrt_version(void) { return "cgrt-mi355x 0.1 (gfx950)"; }

int cgrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cgrt_scene_create(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                      const float* materials, uint32_t nmesh, const float* spheres, uint32_t nspheres, int device, CgrtScene** out) {
    if (!out) return fail(CGRT_E_ARG, "out is NULL");
    *out = nullptr;
    if ((nverts && !pos_nrm) || (ntris && (!tri || !tri_mesh)) || (nmesh && !materials) || (nspheres && !spheres))
        return fail(CGRT_E_ARG, "NULL array with non-zero count");
    if (ntris && (!nverts || !nmesh)) return fail(CGRT_E_ARG, "triangles without vertices or meshes");
    // device == CGRT_DEVICE_NONE builds the tree on the host only (introspection, builder tests);
    // every trace/intersect entry then fails with CGRT_E_NO_DEVICE -- there is no CPU traversal.
    int rc = (device == CGRT_DEVICE_NONE) ? CGRT_OK : select_device(device);
    if (rc) return rc;

    HostScene hs;
    hs.nverts = nverts;
    hs.ntris = ntris;
    hs.nmesh = nmesh;
    hs.nspheres = nspheres;
    try {
        hs.pos_nrm.assign(pos_nrm, pos_nrm + 6 * (size_t)nverts);
        hs.tri.assign(tri, tri + 3 * (size_t)ntris);
        hs.tri_mesh.assign(tri_mesh, tri_mesh + ntris);
        hs.materials.assign(materials, materials + 8 * (size_t)nmesh);
        if (nspheres) hs.spheres.assign(spheres, spheres + 5 * (size_t)nspheres);
        CgrtScene* s = new CgrtScene();
        s->device = device;
        s->ntris = ntris;
        std::string err;
        BuildOptions bo;
        {
            std::lock_guard<std::mutex> lk(g_options_mutex);
            bo = g_build_options;
        }
        if (const char* e = getenv("CGRT_FAST_OPEN")) bo.fast_open = atoi(e);  // experiment knob
        if (!build_reference_bvh(hs, bo, s->bvh, err)) {
            delete s;
            return fail(err.find("deeper") != std::string::npos ? CGRT_E_LIMIT : CGRT_E_ARG, err);
        }
        if (device == CGRT_DEVICE_NONE) {
            *out = s;
            return CGRT_OK;
        }
        uint64_t total = 0;
        {
            static_assert(sizeof(NodePacket) == 64 && sizeof(SubNode) == 64 && sizeof(TriRecord) == 64, "record size");
            const BuiltBvh& B = s->bvh;
            const size_t nrec = B.packets.size() + B.subnodes.size() + B.tris.size();
            hipError_t e = hipMalloc(&s->d_records, nrec ? nrec * 64 : 64);
            char* base = static_cast<char*>(s->d_records);
            if (e == hipSuccess && !B.packets.empty()) e = staged_h2d(base, B.packets.data(), B.packets.size() * 64);
            if (e == hipSuccess && !B.subnodes.empty())
                e = staged_h2d(base + (size_t)B.sub_base * 64, B.subnodes.data(), B.subnodes.size() * 64);
            if (e == hipSuccess && !B.tris.empty())
                e = staged_h2d(base + (size_t)B.tri_base * 64, B.tris.data(), B.tris.size() * 64);
            if (e != hipSuccess) {
                delete s;
                return hip_fail(e, "uploading the record array");
            }
            total += nrec * 64;
        }
        s->nmesh = nmesh;
        if ((rc = upload(hs.materials, &s->d_materials, total))) {
            delete s;
            return rc;
        }
        if ((rc = upload(s->bvh.leaves, &s->d_leaves, total)) || (rc = upload(s->bvh.tri_normals, &s->d_tri_normals, total)) ||
            (rc = upload(s->bvh.spheres, &s->d_spheres, total)) || (rc = upload(s->bvh.tri_leaf, &s->d_tri_leaf, total)) ||
            (rc = upload(s->bvh.paths, &s->d_paths, total))) {
            delete s;
            return rc;
        }
        hipError_t e = hipMalloc((void**)&s->d_queues, 8 * CGRT_QUEUE_BLOCK_WORDS * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(s->d_queues, 0, 8 * CGRT_QUEUE_BLOCK_WORDS * sizeof(unsigned int));
        if (e != hipSuccess) {
            delete s;
            return hip_fail(e, "hipMalloc(queues)");
        }
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                s->persistent_blocks = (unsigned)prop.multiProcessorCount * 4u;
        }
        s->device_bytes = total;
        SceneDev& D = s->dev;
        D.packets = static_cast<const NodePacket*>(s->d_records);
        D.subnodes = static_cast<const SubNode*>(s->d_records);
        D.tris = static_cast<const TriRecord*>(s->d_records);
        D.tri_base = s->bvh.tri_base;
        D.leaves = static_cast<const LeafRec*>(s->d_leaves);
        D.tri_normals = static_cast<const TriNormals*>(s->d_tri_normals);
        D.spheres = static_cast<const SphereRecord*>(s->d_spheres);
        D.tri_leaf = static_cast<const uint32_t*>(s->d_tri_leaf);
        D.paths = static_cast<const float*>(s->d_paths);
        D.scene_eps = s->bvh.scene_absmax * 1.52587890625e-05f;  // 2^-16
        D.fast_boxes = 1;
        for (const TopoNode& n : s->bvh.nodes)
            for (int k = 0; k < 6; k++) {
                const float c = std::fabs(k < 3 ? n.box.lo[k] : n.box.hi[k - 3]);
                if (!(c == 0.0f || (c >= 9.094947017729282e-13f && c <= 1099511627776.0f))) D.fast_boxes = 0;
            }
        // the certificates use the exact fast division, so they need the box envelope too (RayFast::fd is false without it)
        s->fast_root = D.fast_boxes ? s->bvh.fast_root : REF_NONE;
        D.fast_root = s->fast_root;
        D.root_box = s->bvh.root_box;
        D.root_ref = s->bvh.root_ref;
        D.ntris = ntris;
        D.nspheres = nspheres;
        D.npackets = (uint32_t)s->bvh.packets.size();
        D.nleaves = (uint32_t)s->bvh.leaves.size();
        *out = s;
    } catch (const std::bad_alloc&) {
        return fail(CGRT_E_ALLOC, "host allocation failed");
    }
    return CGRT_OK;
}

void cgrt_scene_destroy(CgrtScene* scene) { delete scene; }

int cgrt_set_leaf_accel(int enabled, int sub_leaf_tris) {
    if (sub_leaf_tris < 0 || sub_leaf_tris > 64) return fail(CGRT_E_ARG, "sub_leaf_tris must be in 0..64 (0 = default)");
    std::lock_guard<std::mutex> lk(g_options_mutex);
    g_build_options.leaf_accel = enabled != 0;
    g_build_options.sub_leaf_tris = sub_leaf_tris ? sub_leaf_tris : SUB_LEAF_TRIS;
    return CGRT_OK;
}
int cgrt_set_primary_mode(int mode) {
    if (mode != 0 && mode != 1) return fail(CGRT_E_ARG, "mode must be 0 (wave per tile) or 1 (persistent waves, lane refill)");
    g_primary_mode.store(mode);
    return CGRT_OK;
}
int cgrt_set_kernel_shape(int mode, uint64_t max_rays) {
    if (mode < -1 || mode > 3) return fail(CGRT_E_ARG, "mode must be -1 (by launch size), 0 (lane per ray, 64 per wave), 1 (quad per ray), 2 (lane per ray, 16 per wave) or 3 (4 rays per wave)");
    set_quad_shape(mode, max_rays);
    return CGRT_OK;
}
int cgrt_get_kernel_shape(int* mode, uint64_t* max_rays) {
    int m = 0;
    unsigned long long r = 0;
    get_quad_shape(&m, &r);
    if (mode) *mode = m;
    if (max_rays) *max_rays = r;
    return CGRT_OK;
}
int cgrt_set_fast_tree(int mode) {
    if (mode < -1 || mode > 1) return fail(CGRT_E_ARG, "mode must be -1 (default policy), 0 (never) or 1 (whenever possible)");
    std::lock_guard<std::mutex> lk(g_options_mutex);
    g_build_options.fast_tree = mode;
    return CGRT_OK;
}
int cgrt_scene_set_walk(CgrtScene* s, int certified) {
    if (!s) return fail(CGRT_E_ARG, "scene is NULL");
    if (certified && s->fast_root == REF_NONE) return fail(CGRT_E_ARG, "the scene has no fast tree");
    s->dev.fast_root = certified ? s->fast_root : REF_NONE;
    return CGRT_OK;
}
int cgrt_scene_build_info(const CgrtScene* s, uint32_t* out4) {
    if (!s || !out4) return fail(CGRT_E_ARG, "NULL argument");
    uint32_t wild = 0;
    for (uint8_t w : s->bvh.leaf_wild) wild += w;
    out4[0] = s->bvh.fast_root != REF_NONE ? 1u : 0u;
    out4[1] = wild;
    out4[2] = s->bvh.geometry_finite ? 1u : 0u;
    out4[3] = (uint32_t)s->bvh.leaves.size();
    return CGRT_OK;
}
int cgrt_scene_walk(const CgrtScene* s) { return s ? (s->dev.fast_root != REF_NONE ? 1 : 0) : fail(CGRT_E_ARG, "scene is NULL"); }
int cgrt_num_subnodes(const CgrtScene* s) { return s ? (int)s->bvh.subnodes.size() : fail(CGRT_E_ARG, "scene is NULL"); }

int cgrt_num_levels(const CgrtScene* s) { return s ? s->bvh.levels : fail(CGRT_E_ARG, "scene is NULL"); }
int cgrt_num_nodes(const CgrtScene* s) { return s ? (int)s->bvh.nodes.size() : fail(CGRT_E_ARG, "scene is NULL"); }
double cgrt_build_seconds(const CgrtScene* s) { return s ? s->bvh.build_seconds : 0.0; }
uint64_t cgrt_device_bytes(const CgrtScene* s) { return s ? s->device_bytes : 0; }

int cgrt_get_nodes(const CgrtScene* s, int32_t* meta, float* boxes) {
    if (!s || !meta || !boxes) return fail(CGRT_E_ARG, "NULL argument");
    for (size_t i = 0; i < s->bvh.nodes.size(); i++) {
        const TopoNode& n = s->bvh.nodes[i];
        meta[5 * i] = n.leaf;
        meta[5 * i + 1] = n.level;
        meta[5 * i + 2] = n.child[0];
        meta[5 * i + 3] = n.child[1];
        meta[5 * i + 4] = (int32_t)n.count;
        std::memcpy(boxes + 6 * i, &n.box, 24);
    }
    return CGRT_OK;
}

int64_t cgrt_leaf_prims(const CgrtScene* s, int node, uint32_t* out, uint32_t cap) {
    if (!s || node < 0 || (size_t)node >= s->bvh.nodes.size()) return fail(CGRT_E_ARG, "bad node");
    const TopoNode& n = s->bvh.nodes[node];
    for (uint32_t k = 0; k < n.count && k < cap; k++) out[k] = s->bvh.order[n.first + k];
    return n.count;
}

// Walks the host copy of the device records and checks every reference: child references of the reference tree, leaf
// references in both encodings (REF_LEAF_ACCEL), accelerator child references and runs, alignment of 4-wide nodes, and
// that every triangle is reachable exactly once through the accelerator of its leaf.  0 = consistent.
int cgrt_set_build_threads(int threads) {
    if (threads < 0 || threads > 256) return fail(CGRT_E_ARG, "threads must be in 0..256 (0 = hardware concurrency)");
    std::lock_guard<std::mutex> lk(g_options_mutex);
    g_build_options.threads = threads;
    return CGRT_OK;
}
int cgrt_debug_layout_hash(const CgrtScene* s, uint64_t* out) {
    if (!s || !out) return fail(CGRT_E_ARG, "NULL argument");
    const BuiltBvh& B = s->bvh;
    uint64_t h = 1469598103934665603ull;  // FNV-1a over every array the device reads, in a fixed order
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* q = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; i++) h = (h ^ q[i]) * 1099511628211ull;
        h = (h ^ (uint64_t)n) * 1099511628211ull;
    };
    mix(B.packets.data(), B.packets.size() * sizeof(NodePacket));
    mix(B.subnodes.data(), B.subnodes.size() * sizeof(SubNode));
    mix(B.tris.data(), B.tris.size() * sizeof(TriRecord));
    mix(B.leaves.data(), B.leaves.size() * sizeof(LeafRec));
    mix(B.tri_normals.data(), B.tri_normals.size() * sizeof(TriNormals));
    mix(B.paths.data(), B.paths.size() * sizeof(float));
    mix(B.tri_leaf.data(), B.tri_leaf.size() * sizeof(uint32_t));
    mix(&B.fast_root, sizeof(B.fast_root));
    mix(&B.root_ref, sizeof(B.root_ref));
    *out = h;
    return CGRT_OK;
}
int cgrt_debug_check_layout(CgrtScene* s) {
    if (!s) return fail(CGRT_E_ARG, "NULL scene");
    const BuiltBvh& B = s->bvh;
    if (B.packets.empty() && B.leaves.empty()) return CGRT_OK;
    const uint32_t npk = (uint32_t)B.packets.size(), nsub = (uint32_t)B.subnodes.size(), ntri = (uint32_t)B.tris.size();
    const uint32_t nleaf = (uint32_t)B.leaves.size();
    if (B.sub_base != npk || B.tri_base != npk + nsub) return fail(CGRT_E_ARG, "record bases do not match the array sizes");
    std::vector<uint8_t> leaf_seen(nleaf, 0);
    std::vector<uint32_t> tri_seen(ntri, 0);
    auto check_leaf_ref = [&](uint32_t r) -> bool {
        uint32_t li;
        if (r & REF_LEAF_ACCEL) {
            const uint32_t root = r & REF_INDEX26;
            if (root < B.sub_base || root >= B.tri_base || ((root - B.sub_base) & 1u)) return false;
            li = B.subnodes[root - B.sub_base + 1].pad[0];
            if (li >= nleaf || B.leaves[li].sub_root != root) return false;
        } else {
            li = r & ~REF_LEAF;
            if (li >= nleaf) return false;
            if (B.leaves[li].sub_root != REF_NONE) return false;  // an accelerated leaf must be referenced directly
        }
        if (leaf_seen[li]) return false;  // a tree: every leaf has one parent
        leaf_seen[li] = 1;
        return true;
    };
    auto check_topo_ref = [&](uint32_t r) -> bool {
        if (r == REF_NONE) return false;
        return (r & REF_LEAF) ? check_leaf_ref(r) : (r < npk);
    };
    if (!check_topo_ref(B.root_ref)) return fail(CGRT_E_ARG, "bad root reference");
    const uint32_t real_packets = (uint32_t)(B.nodes.size() - nleaf);
    for (uint32_t i = 0; i < real_packets; i++)
        if (!check_topo_ref(B.packets[i].left) || !check_topo_ref(B.packets[i].right)) return fail(CGRT_E_ARG, "bad child reference in a node packet");
    for (uint32_t li = 0; li < nleaf; li++) {
        const LeafRec& L = B.leaves[li];
        if (!leaf_seen[li]) return fail(CGRT_E_ARG, "leaf not referenced by the tree");
        if (L.first < B.tri_base || (uint64_t)L.first + L.count > (uint64_t)B.tri_base + ntri) return fail(CGRT_E_ARG, "leaf record range");
        if (L.sub_root == REF_NONE) {
            for (uint32_t k = 0; k < L.count; k++) tri_seen[L.first - B.tri_base + k]++;
            continue;
        }
        // walk the leaf's accelerator
        std::vector<uint32_t> todo{L.sub_root};
        while (!todo.empty()) {
            const uint32_t r = todo.back();
            todo.pop_back();
            if (r == REF_NONE) continue;
            if (r & REF_LEAF) {
                const uint32_t first = r & REF_INDEX26, cnt = ((r >> 26) & 31u) + 1u;
                if (first < L.first || first + cnt > L.first + L.count) return fail(CGRT_E_ARG, "run outside its leaf");
                for (uint32_t k = 0; k < cnt; k++) tri_seen[first - B.tri_base + k]++;
                continue;
            }
            if (r < B.sub_base || r >= B.tri_base) return fail(CGRT_E_ARG, "accelerator node reference out of range");
            const uint32_t width_recs = 2u;
            if ((r - B.sub_base) & 1u) return fail(CGRT_E_ARG, "4-wide node not 128-byte aligned");
            for (uint32_t h = 0; h < width_recs; h++) {
                const SubNode& N = B.subnodes[r - B.sub_base + h];
                todo.push_back(N.ref0);
                todo.push_back(N.ref1);
            }
        }
    }
    for (uint32_t t = 0; t < ntri; t++)
        if (tri_seen[t] != 1) return fail(CGRT_E_ARG, "a triangle record is reachable " + std::to_string(tri_seen[t]) + " times");
    if (B.fast_root != REF_NONE) {
        // the fast tree: every reference leaf hangs under it exactly once (as its accelerator root, or as the run of its
        // records), at most TOP_MAX_DEPTH levels deep; paths and tri_leaf describe the reference tree
        if (B.tri_leaf.size() != ntri || B.paths.size() != (size_t)nleaf * PATH_BOXES * 6) return fail(CGRT_E_ARG, "fast tree tables have the wrong size");
        std::vector<uint32_t> seen2(ntri, 0);
        std::vector<std::pair<uint32_t, int>> todo{{B.fast_root, 0}};
        while (!todo.empty()) {
            const uint32_t r = todo.back().first;
            const int depth = todo.back().second;
            todo.pop_back();
            if (r == REF_NONE) continue;
            if (r & REF_LEAF) {
                const uint32_t first = r & REF_INDEX26, cnt = ((r >> 26) & 31u) + 1u;
                if (first < B.tri_base || (uint64_t)first - B.tri_base + cnt > ntri) return fail(CGRT_E_ARG, "fast tree: run out of range");
                for (uint32_t k = 0; k < cnt; k++) seen2[first - B.tri_base + k]++;
                continue;
            }
            if (r < B.sub_base || r >= B.tri_base || ((r - B.sub_base) & 1u)) return fail(CGRT_E_ARG, "fast tree: node reference out of range");
            if (depth >= (int)(FAST_STACK_ENTRIES / (SUB_WIDTH - 1))) return fail(CGRT_E_ARG, "fast tree deeper than its stack allows");
            for (uint32_t h = 0; h < 2; h++) {
                const SubNode& N = B.subnodes[r - B.sub_base + h];
                todo.push_back({N.ref0, depth + 1});
                todo.push_back({N.ref1, depth + 1});
            }
        }
        for (uint32_t t = 0; t < ntri; t++)
            if (seen2[t] != 1) return fail(CGRT_E_ARG, "fast tree: a triangle record is reachable " + std::to_string(seen2[t]) + " times");
        for (size_t i = 0; i < B.nodes.size(); i++) {
            if (!B.nodes[i].leaf) continue;
            const uint32_t li = (uint32_t)B.node_to_ref_index[i];
            const LeafRec& L = B.leaves[li];
            const uint32_t plen = L.path_len & 0xffu, need = L.path_len >> 8;
            if ((int)plen != B.nodes[i].level) return fail(CGRT_E_ARG, "path length != leaf level");
            if (plen && std::memcmp(&B.paths[((size_t)li * PATH_BOXES + plen - 1) * 6], &B.nodes[i].box, 24) != 0)
                return fail(CGRT_E_ARG, "a path does not end in its leaf's box");
            if (plen && (!(need & (1u << (plen - 1))) || (need >> plen))) return fail(CGRT_E_ARG, "a certificate must test its leaf's box and nothing beyond the path");
            for (uint32_t k = 0; k + 1 < plen; k++) {  // a box the certificate skips contains the next one
                if (need & (1u << k)) continue;
                const float* a = &B.paths[((size_t)li * PATH_BOXES + k) * 6];
                const float* b = a + 6;
                for (int ax = 0; ax < 3; ax++)
                    if (!(a[ax] <= b[ax] && a[3 + ax] >= b[3 + ax])) return fail(CGRT_E_ARG, "a skipped path box does not contain its successor");
            }
            for (uint32_t k = 0; k < L.count; k++)
                if (B.tri_leaf[L.first - B.tri_base + k] != li) return fail(CGRT_E_ARG, "tri_leaf does not match the leaf table");
        }
    }
    return CGRT_OK;
}

void cgrt_record_sizes(uint32_t* node_bytes, uint32_t* tri_bytes, uint32_t* sub_bytes, uint32_t* hit_bytes) {
    if (node_bytes) *node_bytes = sizeof(NodePacket);
    if (tri_bytes) *tri_bytes = sizeof(TriRecord);
    if (sub_bytes) *sub_bytes = sizeof(SubNode) * 2 - 16;  // bytes read per accelerator node visit: four boxes (96 B) + four references (16 B)
    if (hit_bytes) *hit_bytes = sizeof(CgrtHit);
}

// ------------------------------------------------------------------------------------------------
#define NEED_DEVICE(s) \
    if ((s)->device < 0) return fail(CGRT_E_NO_DEVICE, "scene was created host-only (CGRT_DEVICE_NONE); there is no CPU traversal path")

// A call lane of the scene for the duration of one host-pointer call (RAII).
namespace {
struct LaneGuard {
    CgrtScene* sc;
    CgrtScene::CallLane* L = nullptr;
    explicit LaneGuard(CgrtScene* s) : sc(s) {}
    ~LaneGuard() {
        if (!L) return;
        // An entry that returns early (an error after work was queued) must not leave copies in flight into its caller's
        // memory or hand a busy lane's staging buffers to the next caller: a finished stream answers the query at once.
        if (hipSetDevice(sc->device) == hipSuccess && hipStreamQuery(L->stream) != hipSuccess) (void)hipStreamSynchronize(L->stream);
        std::lock_guard<std::mutex> lk(sc->lanes_mutex);
        sc->lanes_free.push_back(L);
    }
    int acquire() {
        {
            std::lock_guard<std::mutex> lk(sc->lanes_mutex);
            if (!sc->lanes_free.empty()) {
                L = sc->lanes_free.back();
                sc->lanes_free.pop_back();
                return CGRT_OK;
            }
        }
        CgrtScene::CallLane* n = new (std::nothrow) CgrtScene::CallLane();
        if (!n) return fail(CGRT_E_ALLOC, "host allocation failed");
        hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void**)&n->d_counters, 8 * sizeof(unsigned long long));
        if (e != hipSuccess) {
            if (n->stream) (void)hipStreamDestroy(n->stream);
            delete n;
            return hip_fail(e, "creating a call lane");
        }
        {
            std::lock_guard<std::mutex> lk(sc->lanes_mutex);
            sc->lanes_all.push_back(n);
        }
        L = n;
        return CGRT_OK;
    }
    // scratch that only grows (geometrically): k = 0 rays, 1 hits, 2 normals
    hipError_t dev(int k, size_t bytes, void** out) {
        auto& b = L->dev[k];
        if (b.cap < bytes) {
            if (b.p) (void)hipFree(b.p);
            b.p = nullptr;
            b.cap = 0;
            const size_t want = std::max<size_t>(bytes, 4096) * 3 / 2;
            const hipError_t e = hipMalloc(&b.p, want);
            if (e != hipSuccess) return e;
            b.cap = want;
        }
        *out = b.p;
        return hipSuccess;
    }
    hipError_t pin(int k, size_t bytes, void** out) {
        auto& b = L->pin[k];
        if (b.cap < bytes) {
            if (b.p) (void)hipHostFree(b.p);
            b.p = nullptr;
            b.cap = 0;
            const size_t want = std::max<size_t>(bytes, 4096) * 3 / 2;
            const hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            b.cap = want;
        }
        *out = b.p;
        return hipSuccess;
    }
};
// large host copies on a few threads (one thread moves ~10 GB/s: a 1080p float frame would take as long as 8 device frames)
void parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t chunk = 2u << 20;
    if (bytes < 2 * chunk) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const unsigned nt = (unsigned)std::min<size_t>(4, bytes / chunk);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++)
        pool.emplace_back([=] {
            const size_t b = bytes * t / nt, e = bytes * (t + 1) / nt;
            std::memcpy(static_cast<char*>(dst) + b, static_cast<const char*>(src) + b, e - b);
        });
    std::memcpy(dst, src, bytes / nt);
    for (std::thread& th : pool) th.join();
}

// Transfers below this size go through the lane's pinned staging buffers (a pageable hipMemcpyAsync of a few bytes costs
// far more than copying them twice).  Larger ones -- the lists of a host-driven wavefront, whole frames -- are cut into
// kBounceBytes pieces that alternate between two pinned buffers: the host copies piece k+1 (on a few threads) while the DMA engine
// moves piece k.  Handing the runtime a pageable pointer instead moved ~3 GB/s (66 MB of rays + 56 MB of hits and normals for a
// 1080p list: 40-50 ms around a 0.3 ms kernel, profiles/r3_host_mirror.txt).
const size_t kStageBytes = 1u << 20;
const size_t kBounceBytes = 8u << 20;

hipError_t lane_bounce(LaneGuard& g) {
    for (int b = 0; b < 2; b++) {
        if (!g.L->bounce[b]) {
            const hipError_t e = hipHostMalloc(&g.L->bounce[b], kBounceBytes, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        if (!g.L->bounce_ev[b]) {
            const hipError_t e = hipEventCreateWithFlags(&g.L->bounce_ev[b], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// host -> device on the lane's stream
hipError_t lane_upload(LaneGuard& g, int k, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes <= kStageBytes) {
        void* st = nullptr;
        const hipError_t e = g.pin(k, bytes, &st);
        if (e != hipSuccess) return e;
        std::memcpy(st, src, bytes);
        return hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, g.L->stream);
    }
    hipError_t e = lane_bounce(g);
    if (e != hipSuccess) return e;
    size_t piece = 0;
    for (size_t off = 0; off < bytes; off += kBounceBytes, piece++) {
        const int b = (int)(piece & 1);
        const size_t m = std::min(kBounceBytes, bytes - off);
        if (piece >= 2 && (e = hipEventSynchronize(g.L->bounce_ev[b])) != hipSuccess) return e;  // the DMA out of this half has finished
        parallel_copy(g.L->bounce[b], static_cast<const char*>(src) + off, m);
        if ((e = hipMemcpyAsync(static_cast<char*>(dst) + off, g.L->bounce[b], m, hipMemcpyHostToDevice, g.L->stream)) != hipSuccess) return e;
        if ((e = hipEventRecord(g.L->bounce_ev[b], g.L->stream)) != hipSuccess) return e;
    }
    // the halves are reused by the next transfer of this call: it must not overwrite a piece still being read
    for (int b = 0; b < 2; b++)
        if ((e = hipEventSynchronize(g.L->bounce_ev[b])) != hipSuccess) return e;
    return hipSuccess;
}
// device -> host.  Small: returns the pinned address to copy from after the stream has been synchronised.  Large: the data is in
// dst when the call returns (the stream's earlier work has been waited for), *staged stays null.  `keep` (optional, large path
// only): one flag word per `stride` bytes -- elements whose word is zero are NOT written (normals of rays that missed).
hipError_t lane_download(LaneGuard& g, int k, void* dst, const void* src, size_t bytes, void** staged, const CgrtHit* keep = nullptr,
                         size_t stride = 0) {
    *staged = nullptr;
    if (bytes == 0) return hipSuccess;
    if (bytes <= kStageBytes && !keep) {
        void* st = nullptr;
        const hipError_t e = g.pin(k, bytes, &st);
        if (e != hipSuccess) return e;
        *staged = st;
        return hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, g.L->stream);
    }
    hipError_t e = lane_bounce(g);
    if (e != hipSuccess) return e;
    const size_t piece_bytes = stride ? kBounceBytes / stride * stride : kBounceBytes;
    const size_t npieces = (bytes + piece_bytes - 1) / piece_bytes;
    for (size_t piece = 0; piece <= npieces; piece++) {
        if (piece < npieces) {
            const int b = (int)(piece & 1);
            const size_t off = piece * piece_bytes, m = std::min(piece_bytes, bytes - off);
            if ((e = hipMemcpyAsync(g.L->bounce[b], static_cast<const char*>(src) + off, m, hipMemcpyDeviceToHost, g.L->stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(g.L->bounce_ev[b], g.L->stream)) != hipSuccess) return e;
        }
        if (piece >= 1) {  // copy the previous piece out while this one is on the wire
            const int b = (int)((piece - 1) & 1);
            const size_t off = (piece - 1) * piece_bytes, m = std::min(piece_bytes, bytes - off);
            if ((e = hipEventSynchronize(g.L->bounce_ev[b])) != hipSuccess) return e;
            if (!keep) {
                parallel_copy(static_cast<char*>(dst) + off, g.L->bounce[b], m);
            } else {
                const size_t first = off / stride, cnt = m / stride;
                const unsigned nt = cnt >= 65536 ? 4 : 1;
                auto part = [&](unsigned t) {
                    const char* from = static_cast<const char*>(g.L->bounce[b]);
                    char* to = static_cast<char*>(dst) + off;
                    for (size_t i = cnt * t / nt, e2 = cnt * (t + 1) / nt; i < e2; i++)
                        if (keep[first + i].hit) std::memcpy(to + i * stride, from + i * stride, stride);
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nt; t++) pool.emplace_back(part, t);
                part(0);
                for (std::thread& th : pool) th.join();
            }
        }
    }
    return hipSuccess;
}
}  // namespace

int cgrt_intersect_batch_device(CgrtScene* s, const CgrtRay* d_rays, uint64_t n, CgrtHit* d_hits, float* d_normals, void* stream) {
    if (!s || (n && (!d_rays || !d_hits))) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_trace_batch(s->dev, reinterpret_cast<const float*>(d_rays), n, reinterpret_cast<CgrtHitDev*>(d_hits), d_normals,
                               nullptr, static_cast<hipStream_t>(stream)));
    return CGRT_OK;
}

namespace {
// CPUs this process may use: the affinity mask, cut to the cgroup's quota when /sys/fs/cgroup/cpu.max states one (a container that
// shows 256 hardware threads and grants 16 CPUs throttles a process whose 64 threads all spin).
int usable_cpus() {
    static const int n = [] {
        cpu_set_t set;
        int c = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) c = std::min<long long>(c, (quota + period - 1) / period);
            std::fclose(f);
        }
        return std::max(1, c);
    }();
    return n;
}
inline void futex_wait(std::atomic<uint32_t>& word, uint32_t expected) {  // returns at once when the word no longer holds `expected`
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>& word) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
inline void cpu_relax(unsigned& spins) {  // a wait of a few microseconds is the common case: spin first, then give the core away
    if (++spins < 4096)
        __builtin_ia32_pause();
    else
        std::this_thread::yield();
}
// Returns 1 when the call was served by a combined generation (rc_out = its status), 0 when the caller should take the direct path.
int combined_intersect(CgrtScene* s, const CgrtRay* rays, uint32_t n, CgrtHit* hits, float* normals, int& rc_out) {
    typedef CgrtScene::Combiner C;
    C& cb = s->comb;
    const size_t off_hits = sizeof(CgrtRay) * C::CAP, off_nrm = off_hits + sizeof(CgrtHit) * C::CAP, total = off_nrm + 12 * (size_t)C::CAP;
    int rdy = cb.ready.load(std::memory_order_acquire);
    if (rdy == 0) {  // first call on this scene: the two rings
        std::lock_guard<std::mutex> lk(cb.init_mu);
        rdy = cb.ready.load(std::memory_order_acquire);
        if (rdy == 0) {
            bool ok = hipSetDevice(s->device) == hipSuccess;
            if (const char* e = getenv("CGRT_COMBINE_NRINGS")) cb.nrings = std::max(1, std::min(atoi(e), (int)C::NRINGS));
            for (int k = 0; k < cb.nrings; k++) {
                C::Ring& r = cb.ring[k];
                if (!ok) break;
                hipError_t e = hipHostMalloc(&r.host, total + 64, hipHostMallocMapped);  // (+ the completion word)
                if (e == hipSuccess) std::memset(static_cast<char*>(r.host) + total, 0, 64);
                if (e == hipSuccess) e = hipHostGetDevicePointer(&r.dev, r.host, 0);
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
                ok = e == hipSuccess;  // (whatever was allocated is released with the scene)
            }
            rdy = ok ? 1 : -1;
            cb.ready.store(rdy, std::memory_order_release);
        }
    }
    if (rdy < 0) return 0;
    struct Inside {
        std::atomic<int>& c;
        int before;
        explicit Inside(std::atomic<int>& x) : c(x), before(c.fetch_add(1, std::memory_order_relaxed)) {}
        ~Inside() { c.fetch_sub(1, std::memory_order_relaxed); }
    } inside(cb.inside);
    if (inside.before >= C::MAX_CALLERS) return 0;  // (the bound the word's bit fields and JOIN_MAX rely on)
    // How a caller waits.  With a CPU for every caller, spinning is the fastest hand-over (8 callers: 0.45 M calls/s).  With more
    // callers than CPUs the spinners take the CPUs from the callers that have work to do -- and inside a CPU quota they get the
    // whole process throttled -- so then a caller spins for a moment only and sleeps on a futex; whoever publishes what the
    // sleepers wait for wakes them.  CGRT_COMBINE_SLEEP=0 / 1 forces one or the other.
    static const int sleep_env = [] {
        const char* e = getenv("CGRT_COMBINE_SLEEP");
        return e ? atoi(e) : -1;
    }();
    const bool sleepy = sleep_env >= 0 ? sleep_env != 0 : inside.before + 1 > usable_cpus();
    static const unsigned spin_first = [] {  // pauses before a sleepy caller goes to sleep (a few microseconds)
        const char* e = getenv("CGRT_COMBINE_SPIN");
        return e ? (unsigned)atoi(e) : 20u;
    }();
    // ---- join the open generation, or open a free ring (and lead it) ----
    const int NR = cb.nrings;
    int b = -1;
    uint32_t at = 0;
    uint32_t gen = 0;
    bool leader = false;
    unsigned spins = 0;
    while (b < 0) {
        const uint32_t epoch = cb.epoch.load(std::memory_order_seq_cst);  // (read BEFORE the rings are looked at)
        for (int k = 0; k < NR && b < 0; k++) {
            C::Ring& r = cb.ring[k];
            const uint64_t seen = r.word.load(std::memory_order_acquire);
            if (C::st_of(seen) != C::OPEN || C::count_of(seen) > C::JOIN_MAX) continue;
            const uint64_t w = r.word.fetch_add(((uint64_t)1 << 2) | ((uint64_t)n << 16), std::memory_order_acq_rel);
            if (C::st_of(w) == C::OPEN) {  // joined: the slots [count, count + n) are this caller's
                b = k;
                at = C::count_of(w);
                gen = (uint32_t)(w >> 32);
            }  // else: closed in between -- the stray counts are harmless (see Combiner)
        }
        for (int k = 0; k < NR && b < 0; k++) {
            C::Ring& r = cb.ring[k];
            uint64_t w = r.word.load(std::memory_order_acquire);
            if (C::st_of(w) != C::FREE) continue;
            // (copied == 0 and readers == 0 here: the last reader of the previous generation reset them before it freed the ring)
            if (r.word.compare_exchange_strong(w, C::pack(C::OPEN, 1u, n, w >> 32), std::memory_order_acq_rel, std::memory_order_acquire)) {
                b = k;
                at = 0;
                gen = (uint32_t)(w >> 32);
                leader = true;
            }
        }
        if (b >= 0) break;
        // both rings are on the GPU or being read out: the next generation opens in a moment
        if (sleepy && ++spins > spin_first) {
            bool open = false;  // an OPEN ring that is merely full does not announce itself: keep polling while there is one
            for (int k = 0; k < NR; k++) open |= C::st_of(cb.ring[k].word.load(std::memory_order_relaxed)) == C::OPEN;
            if (!open) {
                cb.epoch_sleepers.fetch_add(1, std::memory_order_seq_cst);
                futex_wait(cb.epoch, epoch);  // until a ring has been set free since `epoch` was read
                cb.epoch_sleepers.fetch_sub(1, std::memory_order_relaxed);
                continue;
            }
        }
        cpu_relax(spins);
    }
    C::Ring& r = cb.ring[b];
    std::memcpy(static_cast<char*>(r.host) + sizeof(CgrtRay) * (size_t)at, rays, sizeof(CgrtRay) * (size_t)n);
    r.copied.fetch_add(1, std::memory_order_release);
    if (leader) {
        // a short grace period while other callers are on their way in (callers parked on the other ring are not coming)
        // -- until everybody inside the entry has joined, or nobody new has for a moment; a few microseconds at most
        uint32_t last = 0, quiet = 0;
        for (unsigned k = 0; k < 1000; k++) {
            int parked = 0;
            for (int k = 0; k < NR; k++) {
                if (k == b) continue;
                const uint64_t ow = cb.ring[k].word.load(std::memory_order_relaxed);
                if (C::st_of(ow) == C::RUNNING || C::st_of(ow) == C::DONE) parked += (int)cb.ring[k].readers.load(std::memory_order_relaxed);
            }
            const uint32_t j = C::joined_of(r.word.load(std::memory_order_relaxed));
            if ((int)j + parked >= cb.inside.load(std::memory_order_relaxed)) break;
            if (j != last) {
                last = j;
                quiet = 0;
            } else if (++quiet > 100) {
                break;
            }
            __builtin_ia32_pause();
        }
        // close: OPEN -> RUNNING fixes the number of rays and of joiners in one step
        uint64_t w = r.word.load(std::memory_order_acquire);
        while (!r.word.compare_exchange_weak(w, C::pack(C::RUNNING, C::joined_of(w), C::count_of(w), w >> 32), std::memory_order_acq_rel,
                                            std::memory_order_acquire)) {
        }
        const uint32_t cnt = C::count_of(w), joined = C::joined_of(w);
        r.readers.store(joined, std::memory_order_relaxed);
        const auto t_closed = std::chrono::steady_clock::now();
        cb.n_gen.fetch_add(1, std::memory_order_relaxed);
        cb.n_rays.fetch_add(cnt, std::memory_order_relaxed);
        for (uint64_t m = cb.max_gen.load(std::memory_order_relaxed); m < cnt && !cb.max_gen.compare_exchange_weak(m, cnt, std::memory_order_relaxed);) {
        }
        unsigned sp = 0;
        while (r.copied.load(std::memory_order_acquire) != joined) cpu_relax(sp);  // every joiner's rays are in the ring
        int rc = CGRT_OK;
        hipError_t e = hipSetDevice(s->device);
        if (e == hipSuccess)
            e = launch_trace_batch(s->dev, static_cast<const float*>(r.dev), cnt, reinterpret_cast<CgrtHitDev*>(static_cast<char*>(r.dev) + off_hits),
                                   reinterpret_cast<float*>(static_cast<char*>(r.dev) + off_nrm), nullptr, r.stream);
        cb.ns_launch.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_closed).count(),
                               std::memory_order_relaxed);
        if (e == hipSuccess) {
            // How the leader waits.  Sleeping in hipStreamSynchronize serialises the two rings: while one leader sleeps there the
            // other leader's launch call does not return (64 callers: 35 of a generation's 61 us were spent getting the launch
            // out, 2.3 us for a lone caller; profiles/r3_per_ray.txt), and polling hipStreamQuery is worse still.  So a one-thread
            // kernel behind the batch writes the generation's number into the ring's pinned memory and the leader watches that
            // word: no runtime call while it waits.  (A launch that never signals -- a fault -- is caught by the bounded spin:
            // the leader then asks the runtime.)  CGRT_COMBINE_WAIT=1: hipStreamSynchronize, =0: hipStreamQuery polling.
            static const int wait_mode = [] {
                const char* w = getenv("CGRT_COMBINE_WAIT");
                return w ? atoi(w) : 2;
            }();
            if (wait_mode == 1) {
                e = hipStreamSynchronize(r.stream);
            } else if (wait_mode == 0) {
                while ((e = hipStreamQuery(r.stream)) == hipErrorNotReady)
                    for (int k = 0; k < 32; k++) __builtin_ia32_pause();
            } else {
                volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(static_cast<char*>(r.host) + total);
                uint32_t* dflag = reinterpret_cast<uint32_t*>(static_cast<char*>(r.dev) + total);
                const uint32_t token = gen + 1u;  // (the word holds the previous generation's token, never this one's)
                e = launch_signal(dflag, token, r.stream);
                if (e == hipSuccess) {
                    unsigned long long spins = 0;
                    while (*flag != token) {
                        __builtin_ia32_pause();
                        if (++spins > 200000000ull) {  // seconds: something is wrong with the launch -- let the runtime say what
                            e = hipStreamSynchronize(r.stream);
                            break;
                        }
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
            }
        }
        if (e != hipSuccess) {
            rc = CGRT_E_HIP;
            r.err = std::string("combined launch: ") + hipGetErrorString(e);
        }
        r.rc = rc;
        cb.ns_gpu.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_closed).count(),
                            std::memory_order_relaxed);
        r.word.store(C::pack(C::DONE, joined, cnt, gen), std::memory_order_release);
        r.done_gen.store(gen + 1u, std::memory_order_seq_cst);
        if (cb.done_sleepers.load(std::memory_order_seq_cst) > 0) futex_wake_all(r.done_gen);
    } else {
        unsigned sp = 0;
        for (;;) {
            const uint32_t dg = r.done_gen.load(std::memory_order_seq_cst);
            if ((int32_t)(dg - gen) > 0) break;
            if (sleepy && ++sp > spin_first) {
                cb.done_sleepers.fetch_add(1, std::memory_order_seq_cst);
                futex_wait(r.done_gen, dg);  // (returns at once if the leader has published in between)
                cb.done_sleepers.fetch_sub(1, std::memory_order_relaxed);
            } else {
                cpu_relax(sp);
            }
        }
    }
    rc_out = r.rc;
    if (r.rc == CGRT_OK) {
        const CgrtHit* rh = reinterpret_cast<const CgrtHit*>(static_cast<char*>(r.host) + off_hits) + at;
        const float* rn = reinterpret_cast<const float*>(static_cast<char*>(r.host) + off_nrm) + 3 * (size_t)at;
        std::memcpy(hits, rh, sizeof(CgrtHit) * (size_t)n);
        if (normals)
            for (uint32_t i = 0; i < n; i++)
                if (rh[i].hit) std::memcpy(normals + 3 * (size_t)i, rn + 3 * (size_t)i, 12);  // HitInfo stays untouched on a miss
    } else {
        g_err = r.err;
    }
    if (r.readers.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // last one out: the ring is free for generation gen + 1
        r.copied.store(0, std::memory_order_relaxed);
        r.word.store(C::pack(C::FREE, 0, 0, (uint64_t)(gen + 1u)), std::memory_order_release);
        cb.epoch.fetch_add(1, std::memory_order_seq_cst);
        if (cb.epoch_sleepers.load(std::memory_order_seq_cst) > 0) futex_wake_all(cb.epoch);
    }
    return 1;
}
}  // namespace

int cgrt_debug_combiner_stats(const CgrtScene* s, uint64_t* out4) {  // (five words, see include/cgrt.h)
    if (!s || !out4) return fail(CGRT_E_ARG, "NULL argument");
    out4[0] = s->comb.n_gen.load();
    out4[1] = s->comb.n_rays.load();
    out4[2] = s->comb.max_gen.load();
    out4[3] = s->comb.ns_gpu.load();
    out4[4] = s->comb.ns_launch.load();
    return CGRT_OK;
}
int cgrt_debug_render_path(const CgrtScene* s) { return s ? s->rpred.last_path : -1; }
int cgrt_debug_hint_counts(CgrtScene* s, uint32_t* out3) {  // the three hard lists' lengths (after a device synchronise): diagnostics
    if (!s || !out3) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    std::lock_guard<std::mutex> lk(s->hints.mu);
    out3[0] = out3[1] = out3[2] = 0;
    if (!s->hints.ready) return CGRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int k = 0; k < 3; k++) HIP_TRY(hipMemcpy(out3 + k, s->hints.phase_host[k].count_r, 4, hipMemcpyDeviceToHost));
    return CGRT_OK;
}
int cgrt_debug_set_hint_thresholds(unsigned dense_ticks, unsigned sparse_ticks) {
    g_hint_thr_dense.store(dense_ticks);
    g_hint_thr_sparse.store(sparse_ticks);
    return CGRT_OK;
}
int cgrt_set_frame_hints(int mode) {
    if (mode < -1 || mode > 2) return fail(CGRT_E_ARG, "frame hint mode: -1 auto, 0 off, 1 hard tiles first, 2 hard tiles 16 rays per wave");
    g_frame_hints.store(mode);
    return CGRT_OK;
}
int cgrt_set_render_prediction(int enabled) {
    g_render_predict.store(enabled ? 1 : 0);
    return CGRT_OK;
}
int cgrt_set_call_combining(int enabled) {
    g_call_combining.store(enabled ? 1 : 0);
    return CGRT_OK;
}

int cgrt_intersect_batch(CgrtScene* s, const CgrtRay* rays, uint64_t n, CgrtHit* hits, float* normals) {
    if (!s || (n && (!rays || !hits))) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    if (n == 0) return CGRT_OK;
    if (n <= CgrtScene::Combiner::MAX_N && g_call_combining.load(std::memory_order_relaxed)) {
        static const bool env_off = [] {
            const char* e = getenv("CGRT_COMBINE");  // experiment knob: 0 = every call takes the direct path
            return e && e[0] == '0';
        }();
        int rc = CGRT_OK;
        if (!env_off && combined_intersect(s, rays, (uint32_t)n, hits, normals, rc)) return rc;
    }
    HIP_TRY(hipSetDevice(s->device));
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void *dr, *dh, *dn = nullptr;
    HIP_TRY(g.dev(0, n * sizeof(CgrtRay), &dr));
    HIP_TRY(g.dev(1, n * sizeof(CgrtHit), &dh));
    HIP_TRY(lane_upload(g, 0, dr, rays, n * sizeof(CgrtRay)));
    // HitInfo is left untouched on a miss (the kernels write a normal only with a hit): a short list starts from the caller's
    // contents on the device; of a long one only the normals of rays that hit are copied back (lane_download's `keep`)
    const bool big = n * sizeof(CgrtHit) > kStageBytes;
    if (normals) {
        HIP_TRY(g.dev(2, n * 12, &dn));
        if (!big) HIP_TRY(lane_upload(g, 2, dn, normals, n * 12));
    }
    rc = cgrt_intersect_batch_device(s, static_cast<const CgrtRay*>(dr), n, static_cast<CgrtHit*>(dh), static_cast<float*>(dn), g.L->stream);
    if (rc) return rc;
    void *sh = nullptr, *sn = nullptr;
    HIP_TRY(lane_download(g, 1, hits, dh, n * sizeof(CgrtHit), &sh));
    if (normals && !big) HIP_TRY(lane_download(g, 2, normals, dn, n * 12, &sn));
    if (normals && big) HIP_TRY(lane_download(g, 2, normals, dn, n * 12, &sn, hits, 12));  // hits is complete here: only rays that hit
    HIP_TRY(hipStreamSynchronize(g.L->stream));
    if (sh) std::memcpy(hits, sh, n * sizeof(CgrtHit));
    if (sn) std::memcpy(normals, sn, n * 12);
    return CGRT_OK;
}

int cgrt_intersect_brute_batch(CgrtScene* s, const CgrtRay* rays, uint64_t n, int mesh, CgrtHit* hits, float* normals) {
    if (!s || (n && (!rays || !hits))) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    if (mesh >= (int)s->nmesh) return fail(CGRT_E_ARG, "mesh index out of range");
    if (n == 0) return CGRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void *dr, *dh, *dn = nullptr;
    HIP_TRY(g.dev(0, n * sizeof(CgrtRay), &dr));
    HIP_TRY(g.dev(1, n * sizeof(CgrtHit), &dh));
    HIP_TRY(lane_upload(g, 0, dr, rays, n * sizeof(CgrtRay)));
    const bool big = n * sizeof(CgrtHit) > kStageBytes;  // as cgrt_intersect_batch
    if (normals) {
        HIP_TRY(g.dev(2, n * 12, &dn));
        if (!big) HIP_TRY(lane_upload(g, 2, dn, normals, n * 12));
    }
    HIP_TRY(launch_brute_batch(s->dev, static_cast<const float*>(dr), n, mesh < 0 ? -1 : mesh, static_cast<CgrtHitDev*>(dh), static_cast<float*>(dn),
                               g.L->stream));
    void *sh = nullptr, *sn = nullptr;
    HIP_TRY(lane_download(g, 1, hits, dh, n * sizeof(CgrtHit), &sh));
    if (normals && !big) HIP_TRY(lane_download(g, 2, normals, dn, n * 12, &sn));
    if (normals && big) HIP_TRY(lane_download(g, 2, normals, dn, n * 12, &sn, hits, 12));  // hits is complete here: only rays that hit
    HIP_TRY(hipStreamSynchronize(g.L->stream));
    if (sh) std::memcpy(hits, sh, n * sizeof(CgrtHit));
    if (sn) std::memcpy(normals, sn, n * 12);
    return CGRT_OK;
}

// Frame hints: which regime a frame is in goes by its size (profiles/r3_frame_hints.txt; six scenes -- the regular and the irregular
// dragon stand-in, the 87 K dragon, dodgeColorTest, monkey, Cornell -- at eight frame shapes).  Frames of up to ~0.8 M rays (640x360,
// 960x540, 800x800) are as long as their longest wave: their hard tiles are traced 16 rays per wave (0 .. -30 %, never more than
// +3 %).  Around 1 M rays the outcome depends on the scene for whole frames and halves of frames (1280x720, half a 1080p frame:
// -8 .. +7 %) -- no hints -- but a rank's share of a frame split over four or more GPUs still gains (1/8 of the 4K frame: -12 ..
// +5 %).  A 1/4 share (2 M rays) gains 2 - 6 % from tracing its hard tiles first in the usual 64-ray waves (mode 1) once the lists
// have settled, but its first twenty-odd frames are slower than plain ones: mode 1 is there to be asked for, the policy does not
// choose it.  A whole frame of that size does not gain (1080p: +2 %), larger frames gain nothing.  cgrt_set_frame_hints forces a
// mode; CGRT_FRAME_HINTS likewise.
#ifndef CGRT_HINT_SPARSE_MAX_RAYS
#define CGRT_HINT_SPARSE_MAX_RAYS 800000ull
#endif
#ifndef CGRT_HINT_SPARSE_MAX_RAYS_SHARE
#define CGRT_HINT_SPARSE_MAX_RAYS_SHARE 1300000ull  // a rank's share, four or more ranks
#endif
#ifndef CGRT_HINT_THR_DENSE
#define CGRT_HINT_THR_DENSE 4500u  // 45 us of s_memrealtime (swept: profiles/r3_frame_hints.txt)
#endif
#ifndef CGRT_HINT_THR_SPARSE
#define CGRT_HINT_THR_SPARSE 2500u  // (kept for cgrt_debug_set_hint_thresholds' second argument: a 16-ray wave counts from 5/9 of the threshold)
#endif
static int hint_mode_for(const FrameDev& F) {
    static const int env = [] {
        const char* e = getenv("CGRT_FRAME_HINTS");
        return e ? atoi(e) : -2;
    }();
    const int m = env >= -1 ? env : g_frame_hints.load();
    if (m >= 0) return m > 2 ? 0 : m;
    const unsigned long long rays = owned_pixels(F);
    if (rays <= CGRT_HINT_SPARSE_MAX_RAYS) return 2;
    if (F.nranks >= 4 && rays <= CGRT_HINT_SPARSE_MAX_RAYS_SHARE) return 2;
    return 0;  // (mode 1 is never chosen: see above)
}
// Attaches the scene's hint buffers to F for ONE launch on `stream` (or leaves F without hints).  Called with s->hints.mu held
// until the launch has been issued.  What keeps the buffers consistent: a frame reads the set the previous frame wrote, writes the
// next and zeroes the counter of the third -- so frames that use them must run one after the other.  Frames issued on one stream
// do.  When the caller changes streams, frames run WITHOUT hints (they touch no buffer) for a few launches, and before hints are
// taken up again the stream that used them last is waited for (it has long finished).
static int attach_hints(CgrtScene* s, FrameDev& F, hipStream_t stream) {
    CgrtScene::FrameHints& Hs = s->hints;
    const int mode = (F.block == 64 && !F.packed && s->dev.fast_root != REF_NONE) ? hint_mode_for(F) : 0;
    if (Hs.last_stream_valid && stream != Hs.last_stream) {
        Hs.cooldown = 8;
        Hs.have_prev = false;
    }
    const hipStream_t prev_stream = Hs.last_stream;
    const bool prev_valid = Hs.last_stream_valid;
    Hs.last_stream = stream;
    Hs.last_stream_valid = true;
    if (mode == 0) {
        Hs.have_prev = false;
        return CGRT_OK;
    }
    if (Hs.cooldown > 0) {
        if (--Hs.cooldown > 0) return CGRT_OK;
        if (prev_valid) (void)hipStreamSynchronize(prev_stream);  // (a destroyed stream answers with an error: nothing of ours is on it then)
        if (Hs.hint_stream_valid && Hs.hint_stream != stream) (void)hipStreamSynchronize(Hs.hint_stream);
        (void)hipGetLastError();
    }
    const int key[8] = {F.W, F.H, F.x0, F.y0, F.x1, F.y1, F.rank, F.nranks};
    const int per_tile = mode == 2 ? 4 : 1;
    const uint32_t ntiles = (uint32_t)F.tiles_x * (uint32_t)F.tiles_y;
    const unsigned td = g_hint_thr_dense.load(), ts = g_hint_thr_sparse.load();
    const unsigned thr[2] = {td ? td : CGRT_HINT_THR_DENSE, ts ? ts : CGRT_HINT_THR_SPARSE};
    if (!Hs.ready || std::memcmp(key, Hs.key, sizeof(key)) != 0 || Hs.per_tile != per_tile || thr[0] != Hs.thr[0] || thr[1] != Hs.thr[1]) {
        // Other buffers are needed: the frame's shape has changed.  Rebuilding waits for whatever used the old ones, so it is only
        // done for a shape that has been asked for three launches in a row -- a caller that alternates between shapes (two
        // viewports, a tool that walks through the ranks) gets plain launches, not a host synchronisation per frame.
        const int wanted[10] = {key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], per_tile, (int)thr[0]};
        if (std::memcmp(wanted, Hs.wanted, sizeof(wanted)) != 0) {
            std::memcpy(Hs.wanted, wanted, sizeof(wanted));
            Hs.wanted_count = 0;
        }
        if (++Hs.wanted_count < 3) {
            Hs.have_prev = false;
            return CGRT_OK;
        }
        Hs.wanted_count = 0;
        // (re)build the buffers.  Whatever used the old ones must have finished.
        if (Hs.hint_stream_valid) (void)hipStreamSynchronize(Hs.hint_stream);
        (void)hipGetLastError();
        const uint64_t owned_tiles = (owned_pixels(F) + 63) / 64;
        // a list holds 4 % of the frame's tiles: the threshold keeps the listed share at 0.5 - 2 % (HintDev)
        const uint32_t cap = (uint32_t)std::min<uint64_t>(0xfffeu, std::max<uint64_t>(64, owned_tiles * 4 / 100));
        const size_t set_words = (size_t)ntiles + cap + 16;  // flag | list | count (+ padding)
        const size_t bytes = 3 * set_words * 4 + 3 * 256 + 64;
        // (hints are an accelerator: if their buffers cannot be had the frame is traced without them, it does not fail)
        hipError_t e = hipSuccess;
        if (!Hs.mailbox) {
            e = hipHostMalloc((void**)&Hs.mailbox, 64, hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&Hs.mailbox_dev, Hs.mailbox, 0);
        }
        if (e == hipSuccess && Hs.mem_bytes < bytes) {
            if (Hs.mem) (void)hipFree(Hs.mem);
            Hs.mem = nullptr;
            Hs.mem_bytes = 0;
            e = hipMalloc(&Hs.mem, bytes);
            if (e == hipSuccess) Hs.mem_bytes = bytes;
        }
        if (e == hipSuccess) e = hipMemsetAsync(Hs.mem, 0, bytes, stream);  // (generation 0 is never used: every flag is stale; on the launch's stream: ordered before it)
        uint32_t* base = static_cast<uint32_t*>(Hs.mem);
        char* structs = reinterpret_cast<char*>(base + 3 * set_words);
        uint32_t* ctl = reinterpret_cast<uint32_t*>(structs + 3 * 256);
        auto flag = [&](int k) { return base + (size_t)k * set_words; };
        auto list = [&](int k) { return flag(k) + ntiles; };
        auto count = [&](int k) { return list(k) + cap; };
        for (int p = 0; p < 3 && e == hipSuccess; p++) {
            HintDev h{};
            const int r = p, w = (p + 1) % 3, z = (p + 2) % 3;
            h.flag_r = flag(r), h.list_r = list(r), h.count_r = count(r);
            h.flag_w = flag(w), h.list_w = list(w), h.count_w = count(w);
            h.count_z = count(z);
            h.ctl = ctl;
            h.mailbox = Hs.mailbox_dev;
            h.cap = cap;
            h.per_tile = (uint32_t)per_tile;
            h.thr_floor = thr[0];
            h.thr_ceil = thr[0] * 8u;
            h.lo = (uint32_t)std::max<uint64_t>(1, owned_tiles / 200);
            h.hi = (uint32_t)std::max<uint64_t>(4, owned_tiles * 2 / 100);
            h.thr_min = (thr[0] * 5u / 9u) - ((thr[0] * 5u / 9u) >> 2);
            Hs.phase[p] = reinterpret_cast<HintDev*>(structs + 256 * p);
            Hs.phase_host[p] = h;
            e = hipMemcpyAsync(Hs.phase[p], &h, sizeof(h), hipMemcpyHostToDevice, stream);  // (pageable source: staged before the call returns)
        }
        const uint32_t thr_start = thr[0] + thr[0] / 2;  // (HintDev: from the side on which a frame is not slower than a plain one)
        if (e == hipSuccess) e = hipMemcpyAsync(ctl, &thr_start, 4, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            Hs.ready = false;
            Hs.have_prev = false;
            return CGRT_OK;
        }
        Hs.mailbox[0] = Hs.mailbox[1] = Hs.mailbox[2] = 0;
        Hs.seen_gen = 0;
        Hs.last_listed = 0;
        Hs.empty_streak = 0;
        Hs.dormant = 0;
        std::memcpy(Hs.key, key, sizeof(key));
        Hs.per_tile = per_tile;
        Hs.thr[0] = thr[0], Hs.thr[1] = thr[1];
        Hs.cap = cap;
        Hs.seq = 0;
        Hs.have_prev = false;
        Hs.ready = true;
    }
    // What the device has said about the lists so far (never waited for: the mailbox is a frame or two behind).
    {
        volatile uint32_t* mb = Hs.mailbox;
        const uint32_t g = mb[0];
        if (g != 0 && g != Hs.seen_gen) {
            Hs.seen_gen = g;
            Hs.last_listed = mb[1];
            // ("empty": a handful of tiles does not pay for the mechanism either -- a monkey's 1/8 share with one listed tile was 6 % slower)
            Hs.empty_streak = Hs.last_listed <= std::max<uint32_t>(2u, Hs.cap / 80u) ? Hs.empty_streak + 1 : 0;
        }
    }
    // A scene without long waves (a Cornell box, a small mesh) gains nothing and pays the mechanism (3-5 us per frame): after eight
    // frames with (all but) empty lists -- at most 0.05 % of the tiles -- the frames run plain for 56 launches, then the hints are
    // tried again.
    if (Hs.dormant > 0) {
        Hs.dormant--;
        Hs.have_prev = false;
        return CGRT_OK;
    }
    if (Hs.empty_streak >= 8) {
        Hs.empty_streak = 0;
        Hs.dormant = 56;
        Hs.have_prev = false;
        return CGRT_OK;
    }
    const uint64_t seq = Hs.seq++;
    auto gen_of = [](uint64_t q) { return (uint32_t)(q % 65535u) + 1u; };
    F.hint = Hs.phase[seq % 3];
    // The launch takes every entry the list has room for.  (Sizing it from the last list length heard of was tried: a caller that
    // issues its frames back to back is dozens of launches ahead of the device, what it has heard is that much out of date -- a
    // list length of 0 from a shape's first frame kept all those launches at 32 entries, and the hints did nothing for them.)
    const uint32_t entries = Hs.cap;
    F.hint_blocks = entries * (uint32_t)per_tile;
    F.hint_rgen = Hs.have_prev ? gen_of(seq - 1) : 0u;
    F.hint_wgen = gen_of(seq);
    Hs.have_prev = true;
    Hs.hint_stream = stream;
    Hs.hint_stream_valid = true;
    return CGRT_OK;
}

static int launch_primary(CgrtScene* s, const CameraDev& C, const FrameDev& F_in, CgrtHitDev* d_hits, float* d_normals, unsigned long long* counters,
                          hipStream_t stream) {
    FrameDev F = F_in;
    if (g_primary_mode.load() == 0 && !counters) {  // (the instrumented and the persistent kernels take no hints)
        std::lock_guard<std::mutex> lk(s->hints.mu);
        const int rc = attach_hints(s, F, stream);
        if (rc) return rc;
        HIP_TRY(launch_trace_primary(s->dev, C, F, d_hits, d_normals, counters, stream));
        return CGRT_OK;
    }
    if (g_primary_mode.load() == 1) {
        std::lock_guard<std::mutex> lk(s->queue_mutex);
        const unsigned k = s->launch_seq++ & 7u;
        unsigned int* q = s->d_queues + CGRT_QUEUE_BLOCK_WORDS * k;
        if (s->queue_done[k])
            HIP_TRY(hipStreamWaitEvent(stream, s->queue_done[k], 0));  // the block's previous launch, on whatever stream it ran
        else
            HIP_TRY(hipEventCreateWithFlags(&s->queue_done[k], hipEventDisableTiming));
        HIP_TRY(launch_trace_primary_persistent(s->dev, C, F, d_hits, d_normals, counters, q, s->persistent_blocks, stream));
        HIP_TRY(hipEventRecord(s->queue_done[k], stream));
    } else {
        HIP_TRY(launch_trace_primary(s->dev, C, F, d_hits, d_normals, counters, stream));
    }
    return CGRT_OK;
}

int cgrt_trace_primary_device(CgrtScene* s, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1, int rank, int nranks,
                              CgrtHit* d_hits, float* d_normals, void* stream) {
    if (!s || !cam || !d_hits) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    FrameDev F;
    if (!make_frame(W, H, x0, y0, x1, y1, rank, nranks, trace_block(s->dev), F)) return fail(CGRT_E_ARG, "bad frame rectangle or rank");
    HIP_TRY(hipSetDevice(s->device));
    return launch_primary(s, make_camera(*cam), F, reinterpret_cast<CgrtHitDev*>(d_hits), d_normals, nullptr, static_cast<hipStream_t>(stream));
}

int cgrt_trace_primary(CgrtScene* s, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1, int rank, int nranks,
                       CgrtHit* hits, float* normals) {
    if (!s || !cam || !hits) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    if (W <= 0 || H <= 0) return fail(CGRT_E_ARG, "bad frame size");
    HIP_TRY(hipSetDevice(s->device));
    const size_t npix = (size_t)W * (size_t)H;
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void *dh, *dn = nullptr;
    HIP_TRY(g.dev(1, npix * sizeof(CgrtHit), &dh));
    // pixels outside the traced tiles keep caller data: seed the device frame with it -- unless this call writes every pixel
    const bool whole_frame = (x0 == 0 && y0 == 0 && x1 == W && y1 == H && nranks == 1);
    if (!whole_frame) HIP_TRY(lane_upload(g, 1, dh, hits, npix * sizeof(CgrtHit)));
    // normals of pixels that miss keep the caller's contents: seeded on the device, or -- a large whole frame, whose hit flags all
    // come from this call -- only the normals of pixels that hit are copied back
    const bool keep_by_hit = whole_frame && npix * sizeof(CgrtHit) > kStageBytes;
    if (normals) {
        HIP_TRY(g.dev(2, npix * 12, &dn));
        if (!keep_by_hit) HIP_TRY(lane_upload(g, 2, dn, normals, npix * 12));
    }
    rc = cgrt_trace_primary_device(s, cam, W, H, x0, y0, x1, y1, rank, nranks, static_cast<CgrtHit*>(dh), static_cast<float*>(dn), g.L->stream);
    if (rc) return rc;
    void *sh = nullptr, *sn = nullptr;
    HIP_TRY(lane_download(g, 1, hits, dh, npix * sizeof(CgrtHit), &sh));
    if (normals) HIP_TRY(lane_download(g, 2, normals, dn, npix * 12, &sn, keep_by_hit ? hits : nullptr, 12));
    HIP_TRY(hipStreamSynchronize(g.L->stream));
    if (sh) std::memcpy(hits, sh, npix * sizeof(CgrtHit));
    if (sn) std::memcpy(normals, sn, npix * 12);
    return CGRT_OK;
}

int cgrt_generate_rays(CgrtScene* s, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1, CgrtRay* rays) {
    if (!s || !cam || !rays) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    FrameDev F;
    if (!make_frame(W, H, x0, y0, x1, y1, 0, 1, CGRT_BLOCK, F)) return fail(CGRT_E_ARG, "bad frame rectangle");
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = (size_t)(x1 - x0) * (size_t)(y1 - y0);
    if (!n) return CGRT_OK;
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void* dr;
    HIP_TRY(g.dev(0, n * sizeof(CgrtRay), &dr));
    HIP_TRY(launch_generate_rays(make_camera(*cam), W, H, x0, y0, x1, y1, static_cast<float*>(dr), g.L->stream));
    void* sr = nullptr;
    HIP_TRY(lane_download(g, 0, rays, dr, n * sizeof(CgrtRay), &sr));
    HIP_TRY(hipStreamSynchronize(g.L->stream));
    if (sr) std::memcpy(rays, sr, n * sizeof(CgrtRay));
    return CGRT_OK;
}

static int read_counters(LaneGuard& g, CgrtCounters* out) {
    unsigned long long h[8];
    HIP_TRY(hipMemcpyAsync(h, g.L->d_counters, sizeof(h), hipMemcpyDeviceToHost, g.L->stream));
    HIP_TRY(hipStreamSynchronize(g.L->stream));
    out->rays = h[0];
    out->inner_visits = h[1];
    out->leaf_visits = h[2];
    out->tri_tests = h[3];
    out->sub_visits = h[4];
    out->cert_boxes = h[5];
    out->fallback_rays = h[6];
    out->tree_rays = h[7];
    return CGRT_OK;
}

int cgrt_count_primary(CgrtScene* s, const CgrtCamera* cam, int W, int H, int x0, int y0, int x1, int y1, int rank, int nranks,
                       CgrtCounters* out) {
    if (!s || !cam || !out) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    FrameDev F;
    if (!make_frame(W, H, x0, y0, x1, y1, rank, nranks, trace_block(s->dev), F)) return fail(CGRT_E_ARG, "bad frame rectangle or rank");
    HIP_TRY(hipSetDevice(s->device));
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void* dh;
    HIP_TRY(g.dev(1, (size_t)W * (size_t)H * sizeof(CgrtHit), &dh));
    HIP_TRY(hipMemsetAsync(g.L->d_counters, 0, 8 * sizeof(unsigned long long), g.L->stream));
    rc = launch_primary(s, make_camera(*cam), F, static_cast<CgrtHitDev*>(dh), nullptr, g.L->d_counters, g.L->stream);
    if (rc) return rc;
    return read_counters(g, out);
}

int cgrt_debug_wave_times(CgrtScene* s, const CgrtCamera* cam, int W, int H, uint64_t* out, uint64_t cap_waves) {
    if (!s || !cam || !out) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    FrameDev F;
    if (!make_frame(W, H, 0, 0, W, H, 0, 1, trace_block(s->dev), F)) return fail(CGRT_E_ARG, "bad frame");
    const size_t nwaves = (size_t)F.nblocks * (size_t)(F.block / 64);
    if (cap_waves < nwaves) return fail(CGRT_E_ARG, "output too small");
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dh, ds;
    HIP_TRY(dh.alloc((size_t)W * (size_t)H * sizeof(CgrtHit)));
    HIP_TRY(ds.alloc(nwaves * 128));
    HIP_TRY(hipMemset(ds.p, 0, nwaves * 128));
    for (int rep = 0; rep < 3; rep++)  // warm caches, keep the last
        HIP_TRY(launch_trace_primary_stamped(s->dev, make_camera(*cam), F, dh.as<CgrtHitDev>(), ds.as<unsigned long long>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, ds.p, nwaves * 128, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_count_batch(CgrtScene* s, const CgrtRay* rays, uint64_t n, CgrtCounters* out) {
    if (!s || !out || (n && !rays)) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    HIP_TRY(hipSetDevice(s->device));
    LaneGuard g(s);
    int rc = g.acquire();
    if (rc) return rc;
    void *dr, *dh;
    HIP_TRY(g.dev(0, n * sizeof(CgrtRay), &dr));
    HIP_TRY(g.dev(1, n * sizeof(CgrtHit), &dh));
    HIP_TRY(lane_upload(g, 0, dr, rays, n * sizeof(CgrtRay)));
    HIP_TRY(hipMemsetAsync(g.L->d_counters, 0, 8 * sizeof(unsigned long long), g.L->stream));
    HIP_TRY(launch_trace_batch(s->dev, static_cast<const float*>(dr), n, static_cast<CgrtHitDev*>(dh), nullptr, g.L->d_counters, g.L->stream));
    return read_counters(g, out);
}

// ------------------------------------------------------------------------------------------------
// renderRayTracing / getFinalColor (src/main.cpp:298-310, :648-720) as a device wavefront
// counted (optional, 3 blocks): the frame is rendered with the instrumented kernels (never timed) and the work of its primary,
// shadow and mirror traversals is returned separately

// rgb (optional): the caller's frame; mapped (optional): receives the scene's pinned staging frame (valid until the next
// cgrt_render* call on this scene).  With nranks > 1 only the pixels this rank owns are meaningful in the staging frame, and only
// those are copied into rgb (pixels of other ranks keep the caller's contents).
static int render_impl(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, const CgrtSoftShadows* soft,
                       int max_level, int rank, int nranks, float* rgb, CgrtRenderStats* stats, CgrtCounters* counted = nullptr,
                       const float** mapped = nullptr) {
    if (!s || !cam || (!rgb && !mapped) || (nlights && !lights)) return fail(CGRT_E_ARG, "NULL argument");
    NEED_DEVICE(s);
    if (W <= 0 || H <= 0 || max_level < 0 || max_level > 16) return fail(CGRT_E_ARG, "bad frame size or recursion depth");
    const unsigned SL = soft ? soft->nspherical : 0;
    if (SL) {
        if (!soft->spherical || !soft->unit_vectors || soft->nunits == 0 || soft->samples == 0 || soft->samples > (1u << 24))
            return fail(CGRT_E_ARG, "soft shadows need lights, a unit-vector table and 1..2^24 samples");
        if ((unsigned long long)W * H > 0x7fffffffull) return fail(CGRT_E_ARG, "frame too large");
    }
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> one_frame(s->render_mutex);  // the workspace below belongs to one frame at a time
    const unsigned long long npix = (unsigned long long)W * H;
    const unsigned L = nlights;
    CgrtRenderStats st{};
    FrameDev F;
    if (!make_frame(W, H, 0, 0, W, H, rank, nranks, trace_block(s->dev), F)) return fail(CGRT_E_ARG, "bad frame or rank");
    const unsigned long long n = (unsigned long long)F.nblocks * (unsigned long long)F.block;  // items: this rank's part of the frame in the primary kernel's order
    // Every level is a compact list: level 0 = the primary rays that hit, level l + 1 = the mirror rays of level l (at most
    // one per entry, so the number of primary hits bounds every list, and n bounds that); the shadow list of a level
    // holds at most entries * L rays.  hits/normals/rays/pixels alternate between two sets (a level's mirror batch is
    // traversed on a second stream while the level itself is still being shaded).
    // hits/normals/rays/pixels rotate through three sets (level l reads set l % 3 and writes its mirror rays into set (l + 1) % 3;
    // level 1 is evaluated on a second stream while level 0 is still being shaded, so its mirror rays need a third set), the
    // shadow lists through two.
    WsBuf rays[3] = {{s, 0}, {s, 1}, {s, 21}}, hits[3] = {{s, 2}, {s, 3}, {s, 22}}, normals[3] = {{s, 4}, {s, 5}, {s, 23}},
          pix[3] = {{s, 6}, {s, 7}, {s, 24}}, ipix{s, 8}, srays[2] = {{s, 9}, {s, 25}}, shits[2] = {{s, 10}, {s, 26}}, sdist[2] = {{s, 11}, {s, 27}},
          sslot[2] = {{s, 12}, {s, 28}}, dlights{s, 13}, levels{s, 14}, drgb{s, 15}, dctr{s, 16}, dslights{s, 17}, dunits{s, 18}, dlit{s, 19},
          dwork{s, 20}, dspawn{s, 29};
    unsigned long long *cw_primary = nullptr, *cw_shadow = nullptr, *cw_mirror = nullptr;
    if (counted) {
        HIP_TRY(dwork.alloc(3 * 8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(dwork.p, 0, 3 * 8 * sizeof(unsigned long long)));
        cw_primary = dwork.as<unsigned long long>();
        cw_shadow = cw_primary + 8;
        cw_mirror = cw_primary + 16;
    }
    HIP_TRY(dspawn.alloc(sizeof(SpawnDev)));
    HIP_TRY(ipix.alloc(n * 4));  // pixels of level 0, kept to the end
    for (int k = 0; k < 3; k++) {
        HIP_TRY(rays[k].alloc(n * 28));
        HIP_TRY(hits[k].alloc(n * sizeof(CgrtHit)));
        HIP_TRY(normals[k].alloc(n * 12));
        HIP_TRY(pix[k].alloc(n * 4));
    }
    for (int k = 0; k < 2; k++) {
        HIP_TRY(srays[k].alloc(n * L * 28));
        HIP_TRY(shits[k].alloc(n * L * sizeof(CgrtHit)));
        HIP_TRY(sdist[k].alloc(n * L * 4));
        HIP_TRY(sslot[k].alloc(n * L * 4));
    }
    HIP_TRY(dlights.alloc((size_t)L * 24));
    HIP_TRY(levels.alloc((size_t)(max_level > 0 ? max_level : 1) * n * 32));
    HIP_TRY(drgb.alloc(npix * 12));
    const size_t nctr = 4 * (size_t)(max_level + 1);
    HIP_TRY(dctr.alloc(nctr * sizeof(uint32_t)));  // per level {shadow rays, mirror rays, hits, -}; the last block: [3] = primary hits
    if (L) HIP_TRY(hipMemcpy(dlights.p, lights, (size_t)L * 24, hipMemcpyHostToDevice));
    SoftDev Q{};
    if (SL) {
        HIP_TRY(dslights.alloc((size_t)SL * 28));
        HIP_TRY(dunits.alloc((size_t)soft->nunits * 12));
        HIP_TRY(dlit.alloc(n * SL * 4));
        HIP_TRY(hipMemcpy(dslights.p, soft->spherical, (size_t)SL * 28, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dunits.p, soft->unit_vectors, (size_t)soft->nunits * 12, hipMemcpyHostToDevice));
        Q.lights = dslights.as<float>();
        Q.units = dunits.as<float>();
        Q.nlights = SL;
        Q.samples = soft->samples;
        Q.nunits = soft->nunits;
        Q.seed = soft->seed;
    }
    const CameraDev C = make_camera(*cam);
    CgrtScene::RenderAux& aux = s->raux;  // second stream + the events that order it against the default stream
    if (!aux.pin_counts) {
        // The second stream carries the frame's critical path (level 0's mirror list, then all of level 1), the default stream the
        // level-0 shadow list beside it.  A higher queue priority for the critical path was measured: within the noise (Cornell
        // 0.185 / 0.193 ms, dragon 0.468 / 0.455 ms with / without, profiles/r3_config3.txt); CGRT_AUX_PRIORITY=1 selects it.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (numerically lower = higher priority)
        const char* ap = getenv("CGRT_AUX_PRIORITY");
        if (ap && ap[0] == '1')
            HIP_TRY(hipStreamCreateWithPriority(&aux.s, hipStreamNonBlocking, prio_hi));
        else
            HIP_TRY(hipStreamCreateWithFlags(&aux.s, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&aux.copy, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&aux.spawned, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&aux.traced, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&aux.primary_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&aux.e0));
        HIP_TRY(hipEventCreate(&aux.e1));
        HIP_TRY(hipHostMalloc((void**)&aux.pin_counts, 64, hipHostMallocDefault));
    }
    const float* const mats = static_cast<const float*>(s->d_materials);
    uint32_t* const primary_hits = dctr.as<uint32_t>() + 4 * (size_t)max_level + 3;
    auto ctr_of = [&](int level) { return dctr.as<uint32_t>() + 4 * (size_t)level; };
    auto lvl_of = [&](int level) { return levels.as<float>() + (size_t)level * n * 8; };
    // ---- The frame as the previous frame of this shape predicts it ----
    // An interactive renderer draws the same scene again and again (main.cpp:776-797 re-renders on every camera change), and what
    // stands between the kernels of one frame is the HOST learning list lengths: a read-back and a round trip (~25-40 us of an
    // idle GPU, profiles/r3_config3_timeline.txt) after the primary kernel, and another one per level beyond the second.  Here
    // every launch of the frame is issued at once: grids cover 1.125 x the previous frame's entries of that level + 1024, the
    // kernels stop at the counts they read on the device, and levels the previous frame did not reach are not issued.  The
    // counters come back once, behind the frame; if any list outgrew its grid, or a level that was not issued turns out to have
    // entries, the frame is drawn again the exact way below (first frames, resized frames and frames with spherical lights or
    // work counters always are).  Same kernels on the same lists: the frame is bit-identical either way (tests/test_render_prediction_gpu.py).
    bool frame_done = false;
    CgrtScene::RenderPred& P = s->rpred;
    const bool predictable = g_render_predict.load() && P.valid && P.W == W && P.H == H && P.rank == rank && P.nranks == nranks &&
                             P.max_level == max_level && P.L == L && !P.counts.empty() && SL == 0 && !counted && max_level >= 1;
    auto predicted = [&]() -> int {
        const int np = (int)P.counts.size();  // levels the previous frame evaluated (P.counts[l] > 0 entries each)
        // {level 0's entries, level 1's entries}: one 64-bit word, filled by the primary kernel's fused spawn with one atomic
        uint32_t* const pair = dctr.as<uint32_t>() + 4 * (size_t)max_level + 2;
        auto count_of = [&](int level) { return level <= 1 ? pair + level : ctr_of(level - 1) + 1; };  // device word: entries of the level
        auto cap_of = [&](int level) { return std::min<unsigned long long>(n, (unsigned long long)P.counts[level] + P.counts[level] / 8 + 1024); };
        HIP_TRY(hipEventRecord(aux.e0, nullptr));
        HIP_TRY(hipMemsetAsync(dctr.p, 0, nctr * sizeof(uint32_t), nullptr));
        // (level 0's spawn is done by the primary kernel itself, from the registers of the lanes that hit: spawn_rays.h)
        SpawnDev SP{};
        SP.materials = mats;
        SP.lights = dlights.as<float>();
        SP.nlights = L;
        SP.spawn = 1 < max_level;
        SP.srays = srays[0].as<float>();
        SP.sdist = sdist[0].as<float>();
        SP.sslot = sslot[0].as<int>();
        SP.lvl = reinterpret_cast<float4*>(lvl_of(0));
        SP.next_rays = rays[1].as<float>();
        SP.next_pixels = pix[1].as<int>();
        if (!aux.spawn_valid || std::memcmp(&SP, &aux.spawn_host, sizeof(SP)) != 0) {  // (the workspace keeps its addresses from frame to frame)
            HIP_TRY(hipMemcpy(dspawn.p, &SP, sizeof(SP), hipMemcpyHostToDevice));
            aux.spawn_host = SP;
            aux.spawn_valid = true;
        }
        HIP_TRY(launch_trace_primary_compact(s->dev, C, F, rays[0].as<float>(), hits[0].as<CgrtHitDev>(), normals[0].as<float>(), ipix.as<int>(),
                                             pair, nullptr, nullptr, drgb.as<float>(), static_cast<const SpawnDev*>(dspawn.p)));
        const unsigned long long cap0 = cap_of(0);
        // Level 0's two lists.  With a fast tree they go out as ONE launch (k_trace_pair: workgroups dealt alternately) and the whole
        // frame stays on one stream; otherwise the shadow list runs on the default stream and the mirror list, with all of level 1
        // behind it, on the second one.
        const bool paired = np >= 2 && L > 0 && can_trace_pair(s->dev);
        hipStream_t const side = paired ? nullptr : aux.s;  // where level 1 is evaluated
        bool tail_on_aux = false;
        if (paired) {
            HIP_TRY(launch_trace_pair(s->dev, srays[0].as<float>(), sdist[0].as<float>(), cap0 * L, shits[0].as<CgrtHitDev>(), pair, L,
                                      (unsigned long long)P.counts[0] * L, rays[1].as<float>(), cap_of(1), hits[1].as<CgrtHitDev>(), normals[1].as<float>(),
                                      pair + 1, P.counts[1], nullptr));
        } else {
            HIP_TRY(hipEventRecord(aux.spawned, nullptr));
            if (L)
                HIP_TRY(launch_trace_shadow(s->dev, srays[0].as<float>(), sdist[0].as<float>(), cap0 * L, shits[0].as<CgrtHitDev>(), nullptr, pair,
                                            nullptr, (unsigned long long)P.counts[0] * L, L));  // (hits x lights rays)
        }
        if (np >= 2) {  // level 1 (and, unpaired, level 0's mirror list in front of it)
            const unsigned long long cap1 = cap_of(1);
            if (!paired) {
                HIP_TRY(hipStreamWaitEvent(aux.s, aux.spawned, 0));
                HIP_TRY(launch_trace_batch(s->dev, rays[1].as<float>(), cap1, hits[1].as<CgrtHitDev>(), normals[1].as<float>(), nullptr, aux.s, pair + 1,
                                           P.counts[1]));
            }
            HIP_TRY(launch_spawn(rays[1].as<float>(), hits[1].as<CgrtHitDev>(), normals[1].as<float>(), pix[1].as<int>(), cap1, mats, dlights.as<float>(), L,
                                 2 < max_level, srays[1].as<float>(), sdist[1].as<float>(), sslot[1].as<int>(), lvl_of(1), rays[2].as<float>(),
                                 pix[2].as<int>(), ctr_of(1), side, pair + 1));
            if (L)
                HIP_TRY(launch_trace_shadow(s->dev, srays[1].as<float>(), sdist[1].as<float>(), cap1 * L, shits[1].as<CgrtHitDev>(), side, ctr_of(1) + 0,
                                            nullptr, (unsigned long long)P.counts[1] * L));
            HIP_TRY(launch_shade(rays[1].as<float>(), hits[1].as<CgrtHitDev>(), normals[1].as<float>(), shits[1].as<CgrtHitDev>(), sdist[1].as<float>(),
                                 sslot[1].as<int>(), cap1, mats, dlights.as<float>(), L, dslights.as<float>(), 0, dlit.as<uint32_t>(), Q.samples, lvl_of(1),
                                 side, pair + 1));
            if (np >= 3)
                HIP_TRY(launch_trace_batch(s->dev, rays[2].as<float>(), cap_of(2), hits[2].as<CgrtHitDev>(), normals[2].as<float>(), nullptr, side,
                                           ctr_of(1) + 1, P.counts[2]));
            if (!paired) HIP_TRY(hipEventRecord(aux.traced, aux.s));
        }
        HIP_TRY(launch_shade(rays[0].as<float>(), hits[0].as<CgrtHitDev>(), normals[0].as<float>(), shits[0].as<CgrtHitDev>(), sdist[0].as<float>(),
                             sslot[0].as<int>(), cap0, mats, dlights.as<float>(), L, dslights.as<float>(), 0, dlit.as<uint32_t>(), Q.samples, lvl_of(0), nullptr,
                             pair));
        if (!paired) {
            // Two levels (the reference's depth, and most frames at any depth): the frame's last kernels are on the second stream, so
            // the scatter into the frame goes there too, behind level 0's shading -- which finished long before -- instead of the
            // default stream waiting for the second one (a cross-stream wait in front of the last kernel cost ~10 us of idle GPU).
            tail_on_aux = np == 2;
            if (tail_on_aux) {
                HIP_TRY(hipEventRecord(aux.primary_done, nullptr));  // (reused: level 0 is shaded)
                HIP_TRY(hipStreamWaitEvent(aux.s, aux.primary_done, 0));
            } else if (np >= 2) {
                HIP_TRY(hipStreamWaitEvent(nullptr, aux.traced, 0));
            }
        }
        hipStream_t const tail = tail_on_aux ? aux.s : nullptr;
        for (int level = 2; level < np; level++) {  // deeper levels: small, one after the other (buffer sets as in the exact path)
            const int a = level % 3, b = (level + 1) % 3, q = level & 1;
            const unsigned long long cap = cap_of(level);
            HIP_TRY(launch_spawn(rays[a].as<float>(), hits[a].as<CgrtHitDev>(), normals[a].as<float>(), pix[a].as<int>(), cap, mats, dlights.as<float>(), L,
                                 level + 1 < max_level, srays[q].as<float>(), sdist[q].as<float>(), sslot[q].as<int>(), lvl_of(level), rays[b].as<float>(),
                                 pix[b].as<int>(), ctr_of(level), nullptr, count_of(level)));
            if (L)
                HIP_TRY(launch_trace_shadow(s->dev, srays[q].as<float>(), sdist[q].as<float>(), cap * L, shits[q].as<CgrtHitDev>(), nullptr,
                                            ctr_of(level) + 0, nullptr, (unsigned long long)P.counts[level] * L));
            HIP_TRY(launch_shade(rays[a].as<float>(), hits[a].as<CgrtHitDev>(), normals[a].as<float>(), shits[q].as<CgrtHitDev>(), sdist[q].as<float>(),
                                 sslot[q].as<int>(), cap, mats, dlights.as<float>(), L, dslights.as<float>(), 0, dlit.as<uint32_t>(), Q.samples, lvl_of(level),
                                 nullptr, count_of(level)));
            if (level + 1 < np)
                HIP_TRY(launch_trace_batch(s->dev, rays[b].as<float>(), cap_of(level + 1), hits[b].as<CgrtHitDev>(), normals[b].as<float>(), nullptr, nullptr,
                                           ctr_of(level) + 1, P.counts[level + 1]));
        }
        for (int level = np - 2; level >= 1; level--) HIP_TRY(launch_fold(lvl_of(level), lvl_of(level + 1), cap_of(level), nullptr, count_of(level)));
        HIP_TRY(launch_write_rgb(lvl_of(0), np >= 2 ? lvl_of(1) : nullptr, cap0, ipix.as<int>(), drgb.as<float>(), tail, pair));
        HIP_TRY(hipEventRecord(aux.e1, tail));
        HIP_TRY(hipEventSynchronize(aux.e1));
        std::vector<uint32_t> hc(nctr);
        HIP_TRY(hipMemcpy(hc.data(), dctr.p, nctr * sizeof(uint32_t), hipMemcpyDeviceToHost));  // (waits for the frame)
        // ---- did every list fit its grid, and did the frame end where it was expected to? ----
        std::vector<uint32_t> actual;  // entries per level, levels with entries only
        bool fits = true;
        for (int level = 0; level < max_level; level++) {
            const uint32_t cnt = level <= 1 ? hc[4 * (size_t)max_level + 2 + level] : hc[4 * (size_t)(level - 1) + 1];
            if (level < np) {
                fits = fits && cnt <= cap_of(level);
            } else {
                fits = fits && cnt == 0;  // a level that was not issued has entries
            }
            if (cnt == 0 || !fits) break;
            actual.push_back(cnt);
        }
        if (!fits) {
            P.valid = false;
            return CGRT_OK;  // (frame_done stays false: the exact path draws the frame)
        }
        st.primary_rays = owned_pixels(F);
        st.shadow_rays = (uint64_t)actual[0] * L;  // (level 0's counter block is not used by the fused spawn)
        st.reflection_rays = hc[4 * (size_t)max_level + 3];
        for (size_t level = 1; level < actual.size(); level++) {
            st.shadow_rays += hc[4 * level + 0];
            st.reflection_rays += hc[4 * level + 1];
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, aux.e0, aux.e1));
        st.device_ms = ms;
        st.levels = (int)actual.size();
        P.counts = actual;
        P.valid = !actual.empty();
        P.last_path = 1;
        frame_done = true;
        return CGRT_OK;
    };
    if (predictable) {
        const int prc = predicted();
        if (prc != CGRT_OK) return prc;
    }
    auto exact = [&]() -> int {
        HIP_TRY(hipEventRecord(aux.e0, nullptr));
        int nlev = 0;
        bool finished = false;  // the frame's last kernels and its closing event have been issued inside the level loop
        std::vector<unsigned long long> level_count;  // entries per evaluated level
        HIP_TRY(hipMemsetAsync(dctr.p, 0, nctr * sizeof(uint32_t), nullptr));
        if (max_level >= 1) {  // trace(level 0): main.cpp:267 returns black without tracing when level >= maxLevel
            // level 0 = the primary rays that hit something, straight out of the fused primary kernel (pixels that miss are
            // black, main.cpp:293, and spawn nothing)
            HIP_TRY(launch_trace_primary_compact(s->dev, C, F, rays[0].as<float>(), hits[0].as<CgrtHitDev>(), normals[0].as<float>(),
                                                 ipix.as<int>(), primary_hits, nullptr, cw_primary, drgb.as<float>()));  // (also clears this rank's pixels)
            st.primary_rays = owned_pixels(F);
            // Level 0's spawn does not wait for the host to learn how many primary rays hit: it is launched over every item of the
            // rank's frame and stops at the count it reads on the device, while the host fetches that count on a stream of its own
            // (behind the primary kernel only) to size the traversal launches that follow -- the host round trip (~25 us of an idle
            // GPU per frame, profiles/r3_config3_timeline.txt) now overlaps the spawn kernel.
            HIP_TRY(hipEventRecord(aux.primary_done, nullptr));
            HIP_TRY(launch_spawn(rays[0].as<float>(), hits[0].as<CgrtHitDev>(), normals[0].as<float>(), ipix.as<int>(), n, mats, dlights.as<float>(), L,
                                 1 < max_level, srays[0].as<float>(), sdist[0].as<float>(), sslot[0].as<int>(), levels.as<float>(), rays[1].as<float>(),
                                 pix[1].as<int>(), dctr.as<uint32_t>(), nullptr, primary_hits));
            // (Issuing level 0's shadow list here too, over its capacity n * L, was measured: Cornell 0.182 -> 0.179 ms, but the dragon
            // frame 0.46 -> 0.50 ms -- a grid of 32 K workgroups for 318 K rays costs more than the round trip it saves.  It is sized
            // exactly after the read-back, and issued FIRST: it used to start 64 us after the spawn kernel ended, behind the second
            // stream's five launches, profiles/r3_config3_timeline.txt.)
            HIP_TRY(hipEventRecord(aux.spawned, nullptr));
            HIP_TRY(hipStreamWaitEvent(aux.copy, aux.primary_done, 0));
            HIP_TRY(hipMemcpyAsync(aux.pin_counts, primary_hits, sizeof(uint32_t), hipMemcpyDeviceToHost, aux.copy));
            HIP_TRY(hipStreamSynchronize(aux.copy));
            unsigned long long cnt = aux.pin_counts[0];
            int level = 0;
            while (level < max_level && cnt > 0) {
                const int a = level % 3, b = (level + 1) % 3, q = level & 1;  // this level's buffer set, the next level's, this level's shadow set
                const int spawn = level + 1 < max_level;
                const int* cur_pix = level == 0 ? ipix.as<int>() : pix[a].as<int>();
                uint32_t* ctr = dctr.as<uint32_t>() + 4 * (size_t)level;
                float* lvl = levels.as<float>() + (size_t)level * n * 8;
                if (level > 0)  // (level 0's spawn is already in flight, see above)
                    HIP_TRY(launch_spawn(rays[a].as<float>(), hits[a].as<CgrtHitDev>(), normals[a].as<float>(), cur_pix, cnt, mats, dlights.as<float>(), L,
                                         spawn, srays[q].as<float>(), sdist[q].as<float>(), sslot[q].as<int>(), lvl, rays[b].as<float>(), pix[b].as<int>(),
                                         ctr, nullptr));
                // Level 0's mirror batch runs on the second stream, beside level 0's shadow batch (two batches of a few hundred
                // thousand rays each; its grid covers the list's capacity -- one mirror ray per entry -- and the kernel stops at the
                // appended count).  Without spherical lights the whole of level 1 follows it there -- spawn, shadow list, shading,
                // level 2's mirror batch: none of it needs level 0's shading, all of it is sized by counts on the device -- so that
                // level 1's tail overlaps level 0's.  Deeper levels are small and often empty: they run one after the other,
                // exactly sized after each level's read-back, or not at all.
                // the level's shadow list first: the long pole of the default stream must not wait behind the second stream's launches
                if (L)
                    HIP_TRY(launch_trace_shadow(s->dev, srays[q].as<float>(), sdist[q].as<float>(), cnt * L, shits[q].as<CgrtHitDev>(), nullptr, ctr + 0,
                                                cw_shadow));
                const bool overlap = spawn && level == 0;
                const bool pipelined = overlap && SL == 0;
                const int spawn1 = 2 < max_level;
                uint32_t* const ctr1 = dctr.as<uint32_t>() + 4;
                if (overlap) {  // (level 0: aux.spawned was recorded right behind the spawn kernel)
                    HIP_TRY(hipStreamWaitEvent(aux.s, aux.spawned, 0));
                    HIP_TRY(launch_trace_batch(s->dev, rays[b].as<float>(), cnt, hits[b].as<CgrtHitDev>(), normals[b].as<float>(), cw_mirror, aux.s,
                                               ctr + 1));
                    if (pipelined) {
                        const int a1 = b, b1 = 2, q1 = 1;
                        float* lvl1 = levels.as<float>() + (size_t)n * 8;
                        HIP_TRY(launch_spawn(rays[a1].as<float>(), hits[a1].as<CgrtHitDev>(), normals[a1].as<float>(), pix[a1].as<int>(), cnt, mats,
                                             dlights.as<float>(), L, spawn1, srays[q1].as<float>(), sdist[q1].as<float>(), sslot[q1].as<int>(), lvl1,
                                             rays[b1].as<float>(), pix[b1].as<int>(), ctr1, aux.s, ctr + 1));
                        if (L)
                            HIP_TRY(launch_trace_shadow(s->dev, srays[q1].as<float>(), sdist[q1].as<float>(), cnt * L, shits[q1].as<CgrtHitDev>(), aux.s,
                                                        ctr1 + 0, cw_shadow));
                        HIP_TRY(launch_shade(rays[a1].as<float>(), hits[a1].as<CgrtHitDev>(), normals[a1].as<float>(), shits[q1].as<CgrtHitDev>(),
                                             sdist[q1].as<float>(), sslot[q1].as<int>(), cnt, mats, dlights.as<float>(), L, dslights.as<float>(), SL,
                                             dlit.as<uint32_t>(), Q.samples, lvl1, aux.s, ctr + 1));
                        if (spawn1)
                            HIP_TRY(launch_trace_batch(s->dev, rays[b1].as<float>(), cnt, hits[b1].as<CgrtHitDev>(), normals[b1].as<float>(), cw_mirror,
                                                       aux.s, ctr1 + 1));
                    }
                    HIP_TRY(hipEventRecord(aux.traced, aux.s));
                }
                if (SL) {
                    Q.level = (uint32_t)level;
                    HIP_TRY(hipMemsetAsync(dlit.p, 0, cnt * SL * 4, nullptr));
                    HIP_TRY(launch_soft_shadow(s->dev, Q, rays[a].as<float>(), hits[a].as<CgrtHitDev>(), cur_pix, cnt, dlit.as<uint32_t>(),
                                               soft->closest_hit == 0, nullptr));
                }
                HIP_TRY(launch_shade(rays[a].as<float>(), hits[a].as<CgrtHitDev>(), normals[a].as<float>(), shits[q].as<CgrtHitDev>(), sdist[q].as<float>(),
                                     sslot[q].as<int>(), cnt, mats, dlights.as<float>(), L, dslights.as<float>(), SL, dlit.as<uint32_t>(), Q.samples, lvl,
                                     nullptr));
                if (overlap) HIP_TRY(hipStreamWaitEvent(nullptr, aux.traced, 0));  // the next level (and the end of the frame) need the second stream's results
                nlev = level + 1;
                level_count.push_back(cnt);
                if (pipelined && !spawn1) {
                    // Depth 2 (the reference's own depth, main.cpp:267): both levels are in flight or done and nothing further depends
                    // on their counts -- the frame is finished without a host round trip (an entry of level 0 without a mirror ray
                    // carries child = -1, so the scatter kernel can fold with level 1 whether or not level 1 has entries); the
                    // counts are read after the frame's closing event.
                    HIP_TRY(launch_write_rgb(levels.as<float>(), levels.as<float>() + (size_t)n * 8, cnt, ipix.as<int>(), drgb.as<float>(), nullptr));
                    finished = true;
                    HIP_TRY(hipEventRecord(aux.e1, nullptr));
                    uint32_t h2[8];
                    HIP_TRY(hipMemcpy(h2, ctr, sizeof(h2), hipMemcpyDeviceToHost));  // levels 0 and 1, adjacent
                    st.shadow_rays += (uint64_t)h2[0] + h2[4];
                    st.reflection_rays += (uint64_t)h2[1] + h2[5];
                    if (h2[1] > 0) {
                        nlev = 2;
                        level_count.push_back(h2[1]);
                    }
                    break;
                }
                uint32_t h[8];
                HIP_TRY(hipMemcpy(h, ctr, pipelined ? sizeof(h) : sizeof(h) / 2, hipMemcpyDeviceToHost));  // also the level's sync point (pipelined: levels 0 and 1, adjacent)
                st.shadow_rays += h[0];
                st.reflection_rays += h[1];
                st.soft_shadow_rays += (uint64_t)h[2] * SL * Q.samples;
                if (!spawn || h[1] == 0) break;
                if (pipelined) {  // level 1 has been evaluated on the second stream, and level 2's rays traversed
                    const uint32_t* h1 = h + 4;
                    nlev = 2;
                    level_count.push_back(h[1]);
                    st.shadow_rays += h1[0];
                    st.reflection_rays += h1[1];
                    if (!spawn1 || h1[1] == 0) break;
                    cnt = h1[1];
                    level = 2;
                    continue;
                }
                if (!overlap)
                    HIP_TRY(launch_trace_batch(s->dev, rays[b].as<float>(), h[1], hits[b].as<CgrtHitDev>(), normals[b].as<float>(), cw_mirror, nullptr));
                cnt = h[1];
                level += 1;
            }
        }
        if (finished) {
        } else if (max_level < 1) {  // trace() returns black without tracing (main.cpp:267): no primary kernel ran, clear here
            HIP_TRY(launch_clear_owned(F, drgb.as<float>(), nullptr));
        } else if (nlev == 0) {  // nothing was hit: the primary kernel has left this rank's pixels black
        } else {
            // color = directColor + reflectedColor * ks (main.cpp:262), deepest level first; the last fold (level 0 with level 1) is
            // done by the kernel that scatters level 0 over the frame
            for (int level = nlev - 2; level >= 1; level--)
                HIP_TRY(launch_fold(levels.as<float>() + (size_t)level * n * 8, levels.as<float>() + (size_t)(level + 1) * n * 8, level_count[level],
                                    nullptr));
            HIP_TRY(launch_write_rgb(levels.as<float>(), nlev >= 2 ? levels.as<float>() + (size_t)n * 8 : nullptr, level_count[0], ipix.as<int>(),
                                     drgb.as<float>(), nullptr));
        }
        if (!finished) HIP_TRY(hipEventRecord(aux.e1, nullptr));
        HIP_TRY(hipEventSynchronize(aux.e1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, aux.e0, aux.e1));
        st.device_ms = ms;
        st.levels = nlev;
        // what this frame found sizes the next one
        P.valid = g_render_predict.load() && SL == 0 && !counted && max_level >= 1 && !level_count.empty();
        P.W = W, P.H = H, P.rank = rank, P.nranks = nranks, P.max_level = max_level, P.L = L;
        P.counts.clear();
        for (unsigned long long c : level_count) P.counts.push_back((uint32_t)c);
        return CGRT_OK;
    };
    if (!frame_done) {
        P.last_path = predictable ? 2 : 0;
        st = CgrtRenderStats{};
        const int erc = exact();
        if (erc != CGRT_OK) return erc;
    }
    {
        const size_t bytes = (size_t)npix * 12;
        if (s->pin_frame_cap < bytes) {
            if (s->pin_frame) (void)hipHostFree(s->pin_frame);
            s->pin_frame = nullptr;
            s->pin_frame_cap = 0;
            HIP_TRY(hipHostMalloc(&s->pin_frame, bytes, hipHostMallocDefault));
            s->pin_frame_cap = bytes;
        }
        HIP_TRY(hipMemcpyAsync(s->pin_frame, drgb.p, bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        const float* pin = static_cast<const float*>(s->pin_frame);
        if (mapped) *mapped = pin;
        if (rgb && nranks == 1) {
            parallel_copy(rgb, pin, bytes);
        } else if (rgb) {  // this rank's super-tiles only (the ownership rule of cgrt_trace_primary)
            const uint64_t nst = (uint64_t)F.st_x * (uint64_t)F.st_y;
            for (uint64_t k = (uint64_t)rank; k < nst; k += (uint64_t)nranks) {
                const int sx = (int)(k % (uint64_t)F.st_x) * 64, sy = (int)(k / (uint64_t)F.st_x) * 64;
                const int w = std::min(64, W - sx), h = std::min(64, H - sy);
                for (int r = 0; r < h; r++) std::memcpy(rgb + 3 * ((size_t)(sy + r) * W + sx), pin + 3 * ((size_t)(sy + r) * W + sx), (size_t)w * 12);
            }
        }
    }
    if (counted) {
        unsigned long long h[24];
        HIP_TRY(hipMemcpy(h, dwork.p, sizeof(h), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3; k++) {
            const unsigned long long* q = h + 8 * k;
            counted[k] = CgrtCounters{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
        }
    }
    if (stats) *stats = st;
    return CGRT_OK;
}

int cgrt_render(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, int max_level, float* rgb,
                CgrtRenderStats* stats) {
    return render_impl(s, cam, W, H, lights, nlights, nullptr, max_level, 0, 1, rgb, stats);
}
int cgrt_render_counted(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, int max_level, float* rgb,
                        CgrtRenderStats* stats, CgrtCounters* work3) {
    if (!work3) return fail(CGRT_E_ARG, "work3 is NULL");
    return render_impl(s, cam, W, H, lights, nlights, nullptr, max_level, 0, 1, rgb, stats, work3);
}
int cgrt_render_soft(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, const CgrtSoftShadows* soft,
                     int max_level, float* rgb, CgrtRenderStats* stats) {
    return render_impl(s, cam, W, H, lights, nlights, soft, max_level, 0, 1, rgb, stats);
}
int cgrt_render_mapped(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, const CgrtSoftShadows* soft,
                       int max_level, const float** rgb, CgrtRenderStats* stats) {
    if (!rgb) return fail(CGRT_E_ARG, "rgb is NULL");
    return render_impl(s, cam, W, H, lights, nlights, soft, max_level, 0, 1, nullptr, stats, nullptr, rgb);
}
int cgrt_render_rank(CgrtScene* s, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights, const CgrtSoftShadows* soft,
                     int max_level, int rank, int nranks, float* rgb, CgrtRenderStats* stats) {
    return render_impl(s, cam, W, H, lights, nlights, soft, max_level, rank, nranks, rgb, stats);
}

// ------------------------------------------------------------------------------------------------
// One caller, N devices, one framebuffer (main.cpp:648-720 yields ONE Screen).  scenes[i] is a
// replica of the scene on its own device (the same device may appear several times); replica i traces the super-tiles
// i % nscenes.  All launches are issued first, each on a private stream of its replica; every device then downloads its
// pixels with ONE asynchronous copy (the kernel writes them back to back, FrameDev::packed) and a host thread per replica
// scatters them into the caller's frame.
namespace {
// pixel of lane (r, c) = row r, column c of the 8x8 tile of wave w of workgroup b: the host mirror of tile_pixel_of()
inline bool tile_origin(const FrameDev& F, uint32_t b, uint32_t w, int& x, int& y) {
    const uint32_t lane8 = b & 7u, j = b >> 3;
    const uint32_t wpb = (uint32_t)F.block / 64u, bps = 64u / wpb;
    const uint32_t sl = (j / bps) * 8u + lane8;
    if (sl >= F.nst_rank) return false;
    const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * sl;
    const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
    const int idx = (int)((j % bps) * wpb + w);
    x = F.x0 + (stx * ST_TILES + (idx & 7)) * 8;
    y = F.y0 + (sty * ST_TILES + (idx >> 3)) * 8;
    return x < F.x1 && y < F.y1;
}
}  // namespace

int cgrt_trace_primary_multi(CgrtScene* const* scenes, int nscenes, const CgrtCamera* cam, int W, int H, CgrtHit* hits, float* normals,
                             CgrtMultiStats* stats) {
    if (!scenes || nscenes <= 0 || nscenes > 64 || !cam || !hits) return fail(CGRT_E_ARG, "NULL argument or bad replica count");
    for (int i = 0; i < nscenes; i++) {
        if (!scenes[i]) return fail(CGRT_E_ARG, "NULL scene");
        NEED_DEVICE(scenes[i]);
        for (int k = 0; k < i; k++)
            if (scenes[k] == scenes[i]) return fail(CGRT_E_ARG, "the same replica twice: create one scene per rank (the same device may repeat)");
    }
    if (W <= 0 || H <= 0) return fail(CGRT_E_ARG, "bad frame size");
    const CameraDev C = make_camera(*cam);
    struct Part {
        FrameDev F;
        LaneGuard* g = nullptr;
        void *dh = nullptr, *dn = nullptr, *ph = nullptr, *pn = nullptr;
        size_t n = 0;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        ~Part() {
            for (hipEvent_t e : {e0, e1, e2})
                if (e) (void)hipEventDestroy(e);
            delete g;
        }
    };
    std::vector<Part> part(nscenes);
    const auto t_begin = std::chrono::steady_clock::now();
    for (int i = 0; i < nscenes; i++) {  // issue everything, wait for nothing
        Part& P = part[i];
        if (!make_frame(W, H, 0, 0, W, H, i, nscenes, trace_block(scenes[i]->dev), P.F)) return fail(CGRT_E_ARG, "bad frame");
        P.F.packed = 1;
        P.n = (size_t)P.F.nblocks * (size_t)P.F.block;
        if (P.n == 0) continue;
        HIP_TRY(hipSetDevice(scenes[i]->device));
        P.g = new LaneGuard(scenes[i]);
        int rc = P.g->acquire();
        if (rc) return rc;
        HIP_TRY(P.g->dev(1, P.n * sizeof(CgrtHit), &P.dh));
        HIP_TRY(P.g->pin(1, P.n * sizeof(CgrtHit), &P.ph));
        if (normals) {
            HIP_TRY(P.g->dev(2, P.n * 12, &P.dn));
            HIP_TRY(P.g->pin(2, P.n * 12, &P.pn));
        }
        HIP_TRY(hipEventCreate(&P.e0));
        HIP_TRY(hipEventCreate(&P.e1));
        HIP_TRY(hipEventCreate(&P.e2));
        hipStream_t st = P.g->L->stream;
        HIP_TRY(hipEventRecord(P.e0, st));
        HIP_TRY(launch_trace_primary(scenes[i]->dev, C, P.F, static_cast<CgrtHitDev*>(P.dh), static_cast<float*>(P.dn), nullptr, st));
        HIP_TRY(hipEventRecord(P.e1, st));
        HIP_TRY(hipMemcpyAsync(P.ph, P.dh, P.n * sizeof(CgrtHit), hipMemcpyDeviceToHost, st));
        if (normals) HIP_TRY(hipMemcpyAsync(P.pn, P.dn, P.n * 12, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(P.e2, st));
    }
    // one host thread per replica: wait for its stream, scatter its tiles (rows of 8 pixels) into the frame
    std::vector<int> status(nscenes, CGRT_OK);
    std::vector<std::string> errs(nscenes);
    auto finish = [&](int i) {
        Part& P = part[i];
        if (P.n == 0) return;
        if (hipSetDevice(scenes[i]->device) != hipSuccess || hipStreamSynchronize(P.g->L->stream) != hipSuccess) {
            status[i] = CGRT_E_HIP;
            errs[i] = "waiting for a replica's stream failed";
            return;
        }
        const CgrtHit* ph = static_cast<const CgrtHit*>(P.ph);
        const float* pn = static_cast<const float*>(P.pn);
        for (uint32_t b = 0; b < P.F.nblocks; b++)
            for (uint32_t w = 0; w < (uint32_t)P.F.block / 64u; w++) {
                int x, y;
                if (!tile_origin(P.F, b, w, x, y)) continue;
                const int cw = std::min(8, P.F.x1 - x), ch = std::min(8, P.F.y1 - y);
                const size_t base = (size_t)b * (size_t)P.F.block + w * 64u;
                for (int r = 0; r < ch; r++) {
                    std::memcpy(hits + (size_t)(y + r) * W + x, ph + base + 8 * r, (size_t)cw * sizeof(CgrtHit));
                    if (normals) {
                        // hitInfo.normal is only written for rays that hit (HitInfo stays untouched on a miss)
                        for (int c = 0; c < cw; c++)
                            if (ph[base + 8 * r + c].hit) std::memcpy(normals + 3 * ((size_t)(y + r) * W + x + c), pn + 3 * (base + 8 * r + c), 12);
                    }
                }
            }
    };
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < nscenes; i++) pool.emplace_back(finish, i);
        finish(0);
        for (std::thread& t : pool) t.join();
    }
    for (int i = 0; i < nscenes; i++)
        if (status[i]) return fail(status[i], errs[i]);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->replicas = nscenes;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        for (int i = 0; i < nscenes; i++) {
            Part& P = part[i];
            if (P.n == 0) continue;
            float k = 0, c = 0;
            (void)hipSetDevice(scenes[i]->device);
            if (hipEventElapsedTime(&k, P.e0, P.e1) == hipSuccess) stats->kernel_ms_max = std::max(stats->kernel_ms_max, k);
            if (hipEventElapsedTime(&c, P.e1, P.e2) == hipSuccess) stats->download_ms_max = std::max(stats->download_ms_max, c);
            stats->rays[i] = owned_pixels(P.F);
        }
    }
    return CGRT_OK;
}

int cgrt_render_multi(CgrtScene* const* scenes, int nscenes, const CgrtCamera* cam, int W, int H, const float* lights, uint32_t nlights,
                      const CgrtSoftShadows* soft, int max_level, float* rgb, CgrtRenderStats* stats) {
    if (!scenes || nscenes <= 0 || nscenes > 64 || !cam || !rgb) return fail(CGRT_E_ARG, "NULL argument or bad replica count");
    for (int i = 0; i < nscenes; i++) {
        if (!scenes[i]) return fail(CGRT_E_ARG, "NULL scene");
        for (int k = 0; k < i; k++)
            if (scenes[k] == scenes[i]) return fail(CGRT_E_ARG, "the same replica twice: create one scene per rank (the same device may repeat)");
    }
    if (W <= 0 || H <= 0) return fail(CGRT_E_ARG, "bad frame size");
    // every replica renders its super-tiles (shading included: a pixel's secondary rays stay on the device that owns it) into a
    // frame of its own on a host thread of its own; the owned pixels are then merged into the caller's frame
    std::vector<CgrtRenderStats> st(nscenes);
    std::vector<int> status(nscenes, CGRT_OK);
    std::vector<std::string> errs(nscenes);
    auto work = [&](int i) {
        // render_impl downloads the replica's frame into its scene's pinned staging and copies ONLY the super-tiles rank i owns
        // into the caller's frame: disjoint regions, so the replicas' threads write rgb concurrently without a merge pass
        status[i] = render_impl(scenes[i], cam, W, H, lights, nlights, soft, max_level, i, nscenes, rgb, &st[i]);
        if (status[i]) errs[i] = g_err;  // (thread-local: carried over to the caller's thread below)
    };
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < nscenes; i++) pool.emplace_back(work, i);
        work(0);
        for (std::thread& t : pool) t.join();
    }
    for (int i = 0; i < nscenes; i++)
        if (status[i]) return fail(status[i], errs[i]);
    CgrtRenderStats tot{};
    for (int i = 0; i < nscenes; i++) {
        tot.primary_rays += st[i].primary_rays;
        tot.shadow_rays += st[i].shadow_rays;
        tot.reflection_rays += st[i].reflection_rays;
        tot.soft_shadow_rays += st[i].soft_shadow_rays;
        tot.levels = std::max(tot.levels, st[i].levels);
        tot.device_ms = std::max(tot.device_ms, st[i].device_ms);
    }
    if (stats) *stats = tot;
    return CGRT_OK;
}

// ------------------------------------------------------------------------------------------------
// element-wise primitives
#define PRIM_PROLOGUE(device)           \
    int rc_ = select_device(device);    \
    if (rc_) return rc_;                \
    if (n == 0) return CGRT_OK;

int cgrt_ray_triangle_batch(int device, const float* tri, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit, float* normals) {
    if (n && (!tri || !rays || !t_out || !hit)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, r, t, h, nn;
    HIP_TRY(a.alloc(n * 72));
    HIP_TRY(r.alloc(n * 28));
    HIP_TRY(t.alloc(n * 4));
    HIP_TRY(h.alloc(n));
    HIP_TRY(hipMemcpy(a.p, tri, n * 72, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r.p, rays, n * 28, hipMemcpyHostToDevice));
    if (normals) {
        HIP_TRY(nn.alloc(n * 12));
        HIP_TRY(hipMemcpy(nn.p, normals, n * 12, hipMemcpyHostToDevice));
    }
    HIP_TRY(launch_ray_triangle(a.as<float>(), r.as<float>(), n, t.as<float>(), h.as<uint8_t>(), normals ? nn.as<float>() : nullptr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(t_out, t.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hit, h.p, n, hipMemcpyDeviceToHost));
    if (normals) HIP_TRY(hipMemcpy(normals, nn.p, n * 12, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_ray_plane_batch(int device, const float* plane, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit) {
    if (n && (!plane || !rays || !t_out || !hit)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, r, t, h;
    HIP_TRY(a.alloc(n * 16));
    HIP_TRY(r.alloc(n * 28));
    HIP_TRY(t.alloc(n * 4));
    HIP_TRY(h.alloc(n));
    HIP_TRY(hipMemcpy(a.p, plane, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r.p, rays, n * 28, hipMemcpyHostToDevice));
    HIP_TRY(launch_ray_plane(a.as<float>(), r.as<float>(), n, t.as<float>(), h.as<uint8_t>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(t_out, t.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hit, h.p, n, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_ray_box_batch(int device, const float* box, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit, uint8_t* inside) {
    if (n && (!box || !rays || !t_out || !hit)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, r, t, h, in;
    HIP_TRY(a.alloc(n * 24));
    HIP_TRY(r.alloc(n * 28));
    HIP_TRY(t.alloc(n * 4));
    HIP_TRY(h.alloc(n));
    HIP_TRY(in.alloc(n));
    HIP_TRY(hipMemcpy(a.p, box, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r.p, rays, n * 28, hipMemcpyHostToDevice));
    HIP_TRY(launch_ray_box(a.as<float>(), r.as<float>(), n, t.as<float>(), h.as<uint8_t>(), in.as<uint8_t>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(t_out, t.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hit, h.p, n, hipMemcpyDeviceToHost));
    if (inside) HIP_TRY(hipMemcpy(inside, in.p, n, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_ray_sphere_batch(int device, const float* sphere, const CgrtRay* rays, uint64_t n, float* t_out, uint8_t* hit, float* normals) {
    if (n && (!sphere || !rays || !t_out || !hit)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, r, t, h, nn;
    HIP_TRY(a.alloc(n * 16));
    HIP_TRY(r.alloc(n * 28));
    HIP_TRY(t.alloc(n * 4));
    HIP_TRY(h.alloc(n));
    HIP_TRY(hipMemcpy(a.p, sphere, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r.p, rays, n * 28, hipMemcpyHostToDevice));
    if (normals) {
        HIP_TRY(nn.alloc(n * 12));
        HIP_TRY(hipMemcpy(nn.p, normals, n * 12, hipMemcpyHostToDevice));
    }
    HIP_TRY(launch_ray_sphere(a.as<float>(), r.as<float>(), n, t.as<float>(), h.as<uint8_t>(), normals ? nn.as<float>() : nullptr, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(t_out, t.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hit, h.p, n, hipMemcpyDeviceToHost));
    if (normals) HIP_TRY(hipMemcpy(normals, nn.p, n * 12, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_debug_gather_calibration(int device, uint64_t nrecords, int repeats) {
    // nrecords x 64 B of zeros, each record read exactly once per launch in a scattered order (see k_gather_calib)
    int n_ = 1;
    (void)n_;
    int rc_ = select_device(device);
    if (rc_) return rc_;
    if (nrecords < 1024 || repeats < 1) return fail(CGRT_E_ARG, "nrecords >= 1024, repeats >= 1");
    DevBuf table, sink;
    HIP_TRY(table.alloc((size_t)nrecords * 64));
    HIP_TRY(sink.alloc(16));
    HIP_TRY(hipMemset(table.p, 0, (size_t)nrecords * 64));
    unsigned long long mult = 2654435761ull;
    auto gcd = [](unsigned long long a, unsigned long long b) {
        while (b) {
            const unsigned long long t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    while (gcd(mult, nrecords) != 1) mult += 2;
    for (int r = 0; r < repeats; r++) HIP_TRY(launch_gather_calib(table.p, nrecords, mult, 12345ull + 7919ull * r, sink.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return CGRT_OK;
}

int cgrt_debug_fastdiv_check(int device, const float* a, const float* d, uint64_t n, uint64_t* mismatches, float* first_bad) {
    if (n && (!a || !d || !mismatches || !first_bad)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf da, dd, dm, db;
    HIP_TRY(da.alloc(n * 4));
    HIP_TRY(dd.alloc(n * 4));
    HIP_TRY(dm.alloc(8));
    HIP_TRY(db.alloc(16));
    HIP_TRY(hipMemcpy(da.p, a, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd.p, d, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dm.p, 0, 8));
    HIP_TRY(hipMemset(db.p, 0, 16));
    HIP_TRY(launch_fastdiv_check(da.as<float>(), dd.as<float>(), n, dm.as<unsigned long long>(), db.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(mismatches, dm.p, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(first_bad, db.p, 16, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_triangle_plane_batch(int device, const float* tri, uint64_t n, float* plane) {
    if (n && (!tri || !plane)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, p;
    HIP_TRY(a.alloc(n * 36));
    HIP_TRY(p.alloc(n * 16));
    HIP_TRY(hipMemcpy(a.p, tri, n * 36, hipMemcpyHostToDevice));
    HIP_TRY(launch_triangle_plane(a.as<float>(), n, p.as<float>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(plane, p.p, n * 16, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

int cgrt_point_in_triangle_batch(int device, const float* in, uint64_t n, uint8_t* out) {
    if (n && (!in || !out)) return fail(CGRT_E_ARG, "NULL argument");
    PRIM_PROLOGUE(device)
    DevBuf a, o;
    HIP_TRY(a.alloc(n * 60));
    HIP_TRY(o.alloc(n));
    HIP_TRY(hipMemcpy(a.p, in, n * 60, hipMemcpyHostToDevice));
    HIP_TRY(launch_point_in_triangle(a.as<float>(), n, o.as<uint8_t>(), nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, o.p, n, hipMemcpyDeviceToHost));
    return CGRT_OK;
}

}  // extern "C"
