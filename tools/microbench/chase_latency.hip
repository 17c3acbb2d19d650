// chase_latency.hip -- what does ONE dependent 128-byte record fetch cost a lone lane on gfx950, by where the record sits?
// A traversal step is "load the record the previous step named, compute, name the next": the latency of that load is the
// floor under every hard ray (DESIGN.md "Latency shape").  One wave, lane 0 only, follows a random permutation cycle through a
// table of 128-byte records (a 16-byte load of the record's first quarter, whose first word names the next record), `steps`
// steps, timed with s_memtime:
//   pass 0  first touch in this launch          (cold: HBM or Infinity Cache, TLB misses included)
//   pass 1  the same records again              (L2 of this XCD, L1 if the chain fits)
// and the whole launch is repeated: launch k >= 1's pass 0 tells whether anything survives a kernel boundary in the L2.
// Table sizes: 8 MiB (fits one L2... 4 MiB per XCD), 128 MiB (the scene's size: fits the 256 MiB Infinity Cache), 1 GiB (HBM).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/chase_latency.hip -o cg-raytracer_amd/lib/chase_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__global__ __launch_bounds__(64) void k_chase(const uint4* __restrict__ table, uint32_t start, int steps, int passes, unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    for (int p = 0; p < passes; p++) {
        uint32_t cur = start;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < steps; i++) {
            const uint4 v = table[(size_t)cur * 8];  // first quarter of a 128-byte record
            cur = v.x;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[2 * p] = t1 - t0;
        out[2 * p + 1] = cur;
    }
}

// `width` independent chains per lane-0 step: memory-level parallelism of one lane (loads issued back to back, one wait)
template <int WIDTH>
__global__ __launch_bounds__(64) void k_chase_mlp(const uint4* __restrict__ table, uint32_t start, uint32_t stride, uint32_t nrec, int steps, unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    uint32_t cur[WIDTH];
    for (int w = 0; w < WIDTH; w++) cur[w] = (start + (uint32_t)w * stride) % nrec;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; i++) {
        uint4 v[WIDTH];
#pragma unroll
        for (int w = 0; w < WIDTH; w++) v[w] = table[(size_t)cur[w] * 8];
#pragma unroll
        for (int w = 0; w < WIDTH; w++) cur[w] = v[w].x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0;
    uint32_t s = 0;
    for (int w = 0; w < WIDTH; w++) s += cur[w];
    out[1] = s;
}

int main() {
    const int steps = 256;
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, 64));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));  // s_memtime ticks at the constant wall clock (100 MHz on MI300-class)
    printf("s_memtime clock: %d kHz (ns per tick %.2f)\n", clk_khz, 1e6 / clk_khz);
    const double ns = 1e6 / clk_khz;
    for (size_t mib : {8, 128, 1024}) {
        const size_t nrec = mib * 1024 * 1024 / 128;
        std::vector<uint32_t> perm(nrec);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(12345);
        for (size_t i = nrec - 1; i > 0; i--) std::swap(perm[i], perm[rng() % (i + 1)]);  // a random cycle: perm[k] -> perm[k+1]
        std::vector<uint32_t> host(nrec * 32, 0u);
        for (size_t k = 0; k < nrec; k++) host[(size_t)perm[k] * 32] = perm[(k + 1) % nrec];
        uint4* d_table;
        CK(hipMalloc(&d_table, nrec * 128));
        CK(hipMemcpy(d_table, host.data(), nrec * 128, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        printf("table %zu MiB (%zu records of 128 B), %d dependent steps, lane 0 of one wave:\n", mib, nrec, steps);
        for (int launch = 0; launch < 3; launch++) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d_table, perm[0], steps, 3, d_out);
            CK(hipDeviceSynchronize());
            unsigned long long h[6];
            CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
            printf("  launch %d (same chain): first touch %.0f ns/step, again %.0f ns/step, third time %.0f ns/step\n", launch, h[0] * ns / steps, h[2] * ns / steps,
                   h[4] * ns / steps);
        }
        // memory-level parallelism of one lane on a fresh part of the table (cold) -- 1, 2, 4 loads in flight
        hipLaunchKernelGGL(k_chase_mlp<1>, dim3(1), dim3(64), 0, 0, d_table, perm[nrec / 2], 7919u, (uint32_t)nrec, steps, d_out);
        CK(hipDeviceSynchronize());
        unsigned long long h1[2], h2[2], h4[2];
        CK(hipMemcpy(h1, d_out, 16, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_chase_mlp<2>, dim3(1), dim3(64), 0, 0, d_table, perm[nrec / 3], 7919u, (uint32_t)nrec, steps, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h2, d_out, 16, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_chase_mlp<4>, dim3(1), dim3(64), 0, 0, d_table, perm[nrec / 5], 7919u, (uint32_t)nrec, steps, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h4, d_out, 16, hipMemcpyDeviceToHost));
        printf("  cold, k loads in flight per step: k=1 %.0f ns/step, k=2 %.0f ns/step, k=4 %.0f ns/step\n", h1[0] * ns / steps, h2[0] * ns / steps, h4[0] * ns / steps);
        CK(hipFree(d_table));
    }
    return 0;
}
