// variant_kernels.hip -- variants of the primary-frame kernel that are NOT the shipped configuration:
//   * k_trace_primary_stamped: the shipped walk with per-wave s_memtime stamps and wave-level work counters
//     (cgrt_debug_wave_times; tools/diag_wave_times.py) -- a diagnostic launch, never timed;
//   * k_trace_primary_persistent (cgrt_set_primary_mode(1)): persistent waves pulling tiles from per-XCD queues with
//     lane refill, on the older "while-while" rounds of the exact walk (walk_round below).  Bit-identical results
//     (tests), measured slower on the bench frame (profiles/r1_exp_persistent_kernel.txt); kept selectable.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "walk_fast.h"

namespace cgrt {

// lane 0 of every wave writes {s_memtime start, end, s_memrealtime start, end}, the wave's summed counters
// {inner, leaf, tri, sub, wave-level inner, sub, tri bodies}, the per-lane maxima {inner, leaf, tri, sub} and its active lane
// count into `stamps` (16 x nwaves u64).
template <bool FAST>
__global__ CGRT_LB void k_trace_primary_stamped(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits,
                                                unsigned long long* __restrict__ stamps) {
    extern __shared__ uint32_t s_lds[];  // CGRT_LDS_WORDS(blockDim.x)
    const int lane = threadIdx.x & 63;
    const uint32_t wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int x = 0, y = 0;
    const bool active = tile_pixel(F, lane, x, y);
    LaneCounters cnt;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    if (active) primary_ray(C, F.W, F.H, x, y, o, d);
    float t = 3.402823466e+38f;
    uint32_t hit_rec = REF_NONE;
    walk_tree<true, FAST>(S, active, o, d, t, hit_rec, CGRT_WAVE_STACK(s_lds), CGRT_WAVE_MAP(s_lds), cnt);
    if (active) finish_ray(S, o, d, t, hit_rec, hits + ((size_t)y * F.W + x), nullptr);
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long nactive = __popcll(__ballot(active));
    uint32_t v[11] = {cnt.inner, cnt.leaf, cnt.tri, cnt.sub, cnt.w_inner, cnt.w_sub, cnt.w_tri, cnt.inner, cnt.leaf, cnt.tri, cnt.sub};
    for (int k = 0; k < 11; k++)
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t other = __shfl_down(v[k], off, 64);
            v[k] = k < 7 ? v[k] + other : (v[k] > other ? v[k] : other);  // sums, then per-lane maxima
        }
    if (lane == 0) {
        unsigned long long* p = stamps + 16ull * wave_global;
        p[0] = st0;
        p[1] = st1;
        p[2] = rt0;
        p[3] = rt1;
        for (int k = 0; k < 11; k++) p[4 + k] = v[k];
        p[15] = nactive;
    }
}

// Topology phase of a round: intersectNonLeaf steps until the lane stands on a leaf (returns false, W.cur = the leaf)
// or has nothing left (returns true).
template <bool COUNT>
__device__ __forceinline__ bool walk_topology(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    const F3 o = W.o, d = W.d;
    uint32_t cur = W.cur;
    int sp = W.sp;
    // ---- topology phase: intersectNonLeaf steps until a leaf is reached ----
    for (;;) {
        if (cur == REF_NONE) {
            bool found = false;
            while (sp > 0) {
                sp -= 2;
                const float ts = __uint_as_float(stk[(sp + 1) * CGRT_STRIDE]);
                const uint32_t r = stk[sp * CGRT_STRIDE];
                if (!(W.t < ts)) {  // bvh.cpp:582: skip the deferred child iff ray.t < tSecond
                    cur = r;
                    found = true;
                    break;
                }
            }
            if (!found) {
                W.cur = REF_NONE;
                W.sp = 0;
                return true;
            }
        }
        if (cur & REF_LEAF) break;
        topo_step<COUNT>(S, W, o, d, cur, sp, stk, cnt);
    }
    // the lane stands on a leaf
    W.cur = cur;
    W.sp = sp;
    return false;
}

template <bool COUNT>
__device__ __forceinline__ bool walk_round(const SceneDev& S, Walk& W, uint32_t* __restrict__ stk, LaneCounters& cnt) {
    if (walk_topology<COUNT>(S, W, stk, cnt)) return true;
    // ---- leaf phase ----
    if (COUNT) cnt.leaf++;
    scan_leaf<COUNT>(S, leaf_rec_of(S, W.cur), W.o, W.d, W.P, W.t, W.hit_rec, stk, W.sp, cnt);
    W.cur = REF_NONE;
    return false;
}
// ---------------------------------------------------------------------------------------------
// Persistent primary-frame kernel: a fixed grid of waves (4 workgroups per CU) pulls 8x8-pixel tiles
// from eight queues (one per blockIdx % 8 residue, i.e. per XCD under the observed placement; a wave
// drains its home queue first, then steals from the others) and keeps its 64 lanes busy: between two
// rounds of walk_round(), when at least CGRT_REFILL_MIN_IDLE lanes have finished their ray, the idle
// lanes are handed the next pixels of the wave's current tile (ballot of the idle lanes, rank by
// popcount of the lower lanes).  Rays that fail the root gate are written out at once and their lane is
// refilled in the same pass, so after a refill the wave's active lanes all carry rays that actually
// traverse.  Tiles are handed out in the same super-tile order as the plain kernel.
// Every wave leaves when all queues are exhausted and its lanes are idle; no wave waits on another.
// The queue block (8 heads + 1 exit counter, one 128-byte line each) is reset by the last wave to leave.
#ifndef CGRT_REFILL_MIN_IDLE
#define CGRT_REFILL_MIN_IDLE 16
#endif
#ifndef CGRT_QUEUE_CHUNK
#define CGRT_QUEUE_CHUNK 1  // tiles taken per atomic; larger chunks were measured much slower (4: 2.4x, 16: 7x): hard tiles cluster
#endif
// Queue block layout: head q at word 32*q (each head on its own 128-byte line: atomics on one line serialise
// at the memory side), exit counter at word 32*8.
#define CGRT_QUEUE_WORDS (32 * 9)

__device__ __forceinline__ uint32_t queue_units(const FrameDev& F, uint32_t q) {  // tiles in queue q
    return q < F.nst_rank ? ((F.nst_rank - q + 7u) / 8u) * 64u : 0u;
}

template <bool COUNT>
__global__ __launch_bounds__(CGRT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_trace_primary_persistent(SceneDev S, CameraDev C, FrameDev F, CgrtHitDev* __restrict__ hits,
                                                   float* __restrict__ normals, unsigned long long* counters,
                                                   unsigned int* __restrict__ queue, unsigned int total_waves) {
    __shared__ uint32_t s_stk[CGRT_STACK_SLOTS * CGRT_BLOCK];
    uint32_t* stk = CGRT_WAVE_STACK(s_stk) + (threadIdx.x & 63u);
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t home = blockIdx.x & 7u;
    LaneCounters cnt;
    Walk W;
    bool active = false;
    size_t pix = 0;
    // wave-uniform refill state
    uint32_t tile_q = 0, tile_k = 0, next_pix = 64, tried = 0, chunk_left = 0;
    bool exhausted = false;
    uint32_t nrays = 0;
    for (;;) {
        unsigned long long act = __ballot(active);
        if (!exhausted && (64 - __popcll(act)) >= CGRT_REFILL_MIN_IDLE) {
            unsigned long long idle = ~act;
            while (idle != 0ull) {
                if (next_pix >= 64u) {  // next tile: rest of the chunk in hand, else home queue first, then the others in order
                    bool got = false;
                    if (chunk_left > 0u) {
                        tile_k += 1u;
                        chunk_left -= 1u;
                        next_pix = 0;
                        got = true;
                    }
                    while (!got && tried < 8u) {
                        const uint32_t q = (home + tried) & 7u;
                        uint32_t k = 0;
                        if (lane == 0) k = atomicAdd(queue + 32u * q, (unsigned)CGRT_QUEUE_CHUNK);
                        k = __builtin_amdgcn_readfirstlane(k);
                        const uint32_t nq = queue_units(F, q);
                        if (k < nq) {
                            tile_q = q;
                            tile_k = k;
                            chunk_left = min((uint32_t)CGRT_QUEUE_CHUNK, nq - k) - 1u;
                            next_pix = 0;
                            got = true;
                            break;
                        }
                        tried++;
                    }
                    if (!got) {
                        exhausted = true;
                        break;
                    }
                }
                const uint32_t nidle = (uint32_t)__popcll(idle);
                const uint32_t take = min(64u - next_pix, nidle);
                const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                const bool mine = ((idle >> lane) & 1ull) && rank < take;
                if (mine) {
                    const uint32_t p = next_pix + rank;
                    const uint32_t s_loc = tile_q + 8u * (tile_k >> 6);  // rank-local super-tile
                    const uint32_t st = (uint32_t)F.rank + (uint32_t)F.nranks * s_loc;
                    const int stx = (int)(st % (uint32_t)F.st_x), sty = (int)(st / (uint32_t)F.st_x);
                    const int idx = (int)(tile_k & 63u);
                    const int x = F.x0 + (stx * ST_TILES + (idx & 7)) * 8 + (int)(p & 7u);
                    const int y = F.y0 + (sty * ST_TILES + (idx >> 3)) * 8 + (int)(p >> 3);
                    if (x < F.x1 && y < F.y1) {
                        if (COUNT) nrays++;
                        pix = (size_t)y * F.W + x;
                        primary_ray(C, F.W, F.H, x, y, W.o, W.d);
                        W.t = 3.402823466e+38f;  // std::numeric_limits<float>::max(), trackball.cpp:101
                        if (walk_begin(S, W)) {
                            active = true;
                            if (COUNT) cnt.entered++;
                        } else
                            finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
                    }
                }
                next_pix += take;
                idle = ~__ballot(active);  // lanes whose ray died at the root gate are idle again
                if (__popcll(idle) < CGRT_REFILL_MIN_IDLE && next_pix < 64u) break;
            }
            act = __ballot(active);
        }
        if (act == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (active) {
            if (walk_round<COUNT>(S, W, stk, cnt)) {
                finish_ray(S, W.o, W.d, W.t, W.hit_rec, hits + pix, normals ? normals + 3 * pix : nullptr);
                active = false;
            }
        }
    }
    if (COUNT) {
        // all eight counters of CgrtCounters (this kernel walks exactly: no certificates, no fallbacks)
        unsigned long long v[8] = {nrays, cnt.inner, cnt.leaf, cnt.tri, cnt.sub, 0ull, 0ull, cnt.entered};
        for (int k = 0; k < 8; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(counters + k, x);
        }
    }
    if (lane == 0) {
        if (atomicAdd(queue + 32 * 8, 1u) == total_waves - 1u) {  // last wave out: leave the queue block clean
            for (int q = 0; q < 9; q++) queue[32 * q] = 0u;
        }
    }
}

hipError_t launch_trace_primary_persistent(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits, float* normals,
                                           unsigned long long* counters, unsigned int* queue, unsigned blocks, hipStream_t stream) {
    if (F.nst_rank == 0) return hipSuccess;
    const unsigned long long tiles = (unsigned long long)((F.nst_rank + 7u) / 8u) * 8u * 64u;
    unsigned b = (unsigned)std::min<unsigned long long>(blocks, (tiles + 3) / 4);
    b = std::max(8u, (b + 7u) & ~7u);
    const unsigned waves = b * (CGRT_BLOCK / 64);
    // the queue block is reset on the launch's own stream: a launch that was aborted, or more launches in flight than
    // the caller's ring has blocks, must not leave stale heads behind
    hipError_t e = hipMemsetAsync(queue, 0, CGRT_QUEUE_BLOCK_WORDS * sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    if (counters)
        hipLaunchKernelGGL(k_trace_primary_persistent<true>, dim3(b), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters, queue, waves);
    else
        hipLaunchKernelGGL(k_trace_primary_persistent<false>, dim3(b), dim3(CGRT_BLOCK), 0, stream, S, C, F, hits, normals, counters, queue, waves);
    return hipGetLastError();
}
hipError_t launch_trace_primary_stamped(const SceneDev& S, const CameraDev& C, const FrameDev& F, CgrtHitDev* hits,
                                        unsigned long long* stamps, hipStream_t stream) {
    if (F.nblocks == 0) return hipSuccess;
    if (S.fast_root != REF_NONE)
        hipLaunchKernelGGL(k_trace_primary_stamped<true>, dim3(F.nblocks), dim3((unsigned)F.block), sizeof(uint32_t) * (size_t)CGRT_LDS_WORDS(F.block), stream, S, C, F, hits, stamps);
    else
        hipLaunchKernelGGL(k_trace_primary_stamped<false>, dim3(F.nblocks), dim3((unsigned)F.block), sizeof(uint32_t) * (size_t)CGRT_LDS_WORDS(F.block), stream, S, C, F, hits, stamps);
    return hipGetLastError();
}

}  // namespace cgrt
