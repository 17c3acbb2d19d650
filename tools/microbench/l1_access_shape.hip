// l1_access_shape.hip -- what does the vector L1 of a gfx950 CU charge for: bytes, or distinct lines per instruction?
// Every lane issues eight 16-byte loads per iteration in both modes, so bytes and instruction counts are equal:
//   mode 0 "lane per record"  : lane l reads the eight quarters of ITS OWN 128-byte record (a different record per lane and
//                               iteration): every instruction touches 64 lines -- the traversal kernels' pattern;
//   mode 1 "eight lanes per record": lanes 8g .. 8g+7 read the eight quarters of one record side by side: every instruction
//                               touches 8 lines (the cooperative fetch of DESIGN.md section 9).
// Records are drawn from a pool of `nrec` 128-byte records (small pool: L1 hits; large pool: L2 / Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/l1_access_shape.hip -o cg-raytracer_amd/lib/l1_access_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_shape(const float4* __restrict__ pool, unsigned nrec_mask, int iters, float* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63u, wave = blockIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        float4 v[8];
        if (MODE == 0) {
            const unsigned r = hash32(wave * 64u + lane + 0x9e3779b9u * (unsigned)it) & nrec_mask;
            const float4* p = pool + (size_t)r * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[k];
        } else if (MODE == 2 || MODE == 3) {
            // the traversal's shape with narrower loads: same number of instructions, half / a quarter of the bytes
            const unsigned r = hash32(wave * 64u + lane + 0x9e3779b9u * (unsigned)it) & nrec_mask;
            const float4* p = pool + (size_t)r * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (MODE == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(p + k);
                    v[k] = make_float4(t.x, 0.0f, 0.0f, t.y);
                } else {
                    const float t = *reinterpret_cast<const float*>(p + k);
                    v[k] = make_float4(t, 0.0f, 0.0f, 0.0f);
                }
            }
        } else if (MODE == 6) {
            // every lane reads the SAME record with vector loads (the top levels of a coherent wave): one line, 1 KB delivered
            const unsigned r = hash32(wave * 64u + 0x9e3779b9u * (unsigned)it) & nrec_mask;
            const float4* p = pool + (size_t)r * 8;
            unsigned zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));  // keeps the address in vector registers: vector loads, not scalar ones
            p += zero;
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[k];
        } else if (MODE == 7) {
            // the same record through the scalar cache (uniform address -> s_load), then used by every lane
            const unsigned r = __builtin_amdgcn_readfirstlane(hash32(wave * 64u + 0x9e3779b9u * (unsigned)it) & nrec_mask);
            const float4* p = pool + (size_t)r * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = p[k];
        } else if (MODE == 4 || MODE == 5) {
            // controls: 4 = the whole wave reads 256 contiguous bytes (4 B per lane, two lines) of a random record pair;
            //           5 = eight lanes per record like mode 1, but 4 B per lane (eight lines, 32 B each)
            const unsigned g = lane >> 3, q = lane & 7u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float t;
                if (MODE == 4) {
                    const unsigned r = hash32(wave * 64u + (unsigned)k + 0x9e3779b9u * (unsigned)it) & nrec_mask & ~1u;
                    t = reinterpret_cast<const float*>(pool + (size_t)r * 8)[lane];
                } else {
                    const unsigned r = hash32(wave * 64u + (g * 8u + (unsigned)k) + 0x9e3779b9u * (unsigned)it) & nrec_mask;
                    t = reinterpret_cast<const float*>(pool + (size_t)r * 8)[q];
                }
                v[k] = make_float4(t, 0.0f, 0.0f, 0.0f);
            }
        } else {
            const unsigned g = lane >> 3, q = lane & 7u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned r = hash32(wave * 64u + (g * 8u + (unsigned)k) + 0x9e3779b9u * (unsigned)it) & nrec_mask;
                v[k] = pool[(size_t)r * 8 + q];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k].x + v[k].w;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 12;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int ncu = prop.multiProcessorCount;
    const int blocks = ncu * waves_per_cu, iters = 2000;
    float* out; hipMalloc(&out, (size_t)blocks * 64 * sizeof(float));
    printf("gfx950 L1 access shape: %d CUs x %d waves, %d iterations x 8 x 16 B per lane\n", ncu, waves_per_cu, iters);
    for (unsigned nrec : {64u, 32768u, 1048576u}) {  // 8 KB, 4 MB, 128 MB pools
        float4* pool; hipMalloc(&pool, (size_t)nrec * 128);
        hipMemset(pool, 0, (size_t)nrec * 128);
        for (int mode = 0; mode < 8; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, nullptr);
                if (mode == 0) hipLaunchKernelGGL(k_shape<0>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 1) hipLaunchKernelGGL(k_shape<1>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 2) hipLaunchKernelGGL(k_shape<2>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 3) hipLaunchKernelGGL(k_shape<3>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 4) hipLaunchKernelGGL(k_shape<4>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 5) hipLaunchKernelGGL(k_shape<5>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else if (mode == 6) hipLaunchKernelGGL(k_shape<6>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                else hipLaunchKernelGGL(k_shape<7>, dim3(blocks), dim3(64), 0, nullptr, pool, nrec - 1, iters, out);
                hipEventRecord(e1, nullptr);
                hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_cu = (double)waves_per_cu * iters * 8;  // wave-level load instructions per CU
            const double bytes = (double)blocks * 64 * iters * (mode == 2 ? 64.0 : ((mode >= 3 && mode <= 5) ? 32.0 : 128.0));
            printf("pool %8u records (%7.0f KB)  %s : %8.3f ms  %7.1f ns per wave-level load per CU  %8.1f GB/s aggregate\n", nrec, nrec * 128 / 1024.0,
                   mode == 0 ? "lane per record, 16 B  " : (mode == 1 ? "8 lanes per record,16 B" : (mode == 2 ? "lane per record, 8 B   " : (mode == 3 ? "lane per record, 4 B   " : (mode == 4 ? "wave reads 256 B, 4 B  " : (mode == 5 ? "8 lanes per record, 4 B" : (mode == 6 ? "all lanes 1 record,16 B" : "1 record, scalar loads ")))))), ms, ms * 1e6 / insts_per_cu, bytes / ms / 1e6);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        hipFree(pool);
    }
    hipFree(out);
    return 0;
}
