// ubench_gather.hip -- how fast can a CU gather random 64-byte records (BVH nodes)?
// Compares per-lane fetches (4 x dwordx4 per lane, every lane its own line) with quad-cooperative
// fetches (4 lanes share one 64-byte line per instruction) through registers or LDS-DMA.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o gpurun_out/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x;
}

// mode 0: each lane loads its own 64-B record with 4 dwordx4
// mode 1: quad-cooperative through registers (no redistribution; just the access pattern)
// mode 2: quad-cooperative LDS-DMA + 4 ds_read_b128 per lane (own record re-assembled)
// mode 3: each lane loads 16 B only (1 dwordx4) of its own record
// mode 4: each lane loads 32 B (2 dwordx4)
template <int MODE>
__global__ __launch_bounds__(64) void gather_kernel(const float4 *__restrict__ table, uint32_t nrec, int iters, float *out) {
    __shared__ float4 lds[4 * 64];
    const uint32_t lane = threadIdx.x;
    uint32_t h = mix32(blockIdx.x * 64u + lane + 1u);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        h = mix32(h + 0x9E3779B9u);
        const uint32_t rec = h % nrec;
        if (MODE == 0) {
            const float4 *p = table + (size_t)rec * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 3) {
            const float4 *p = table + (size_t)rec * 4;
            float4 a = p[0];
            acc += a.x;
        } else if (MODE == 4) {
            const float4 *p = table + (size_t)rec * 4;
            float4 a = p[0], b = p[1];
            acc += a.x + b.y;
        } else if (MODE == 1) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rk = __shfl(rec, (lane & ~3u) + k);
                v[k] = table[(size_t)rk * 4 + (lane & 3u)];
            }
            acc += v[0].x + v[1].y + v[2].z + v[3].w;
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rk = __shfl(rec, (lane & ~3u) + k);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + (size_t)rk * 4 + (lane & 3u)),
                                                 (__attribute__((address_space(3))) void *)(lds + k * 64), 16, 0, 0);
            }
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4 *q = lds + (lane & 3u) * 64 + (lane & ~3u);
            float4 a = q[0], b = q[1], c = q[2], d = q[3];
            acc += a.x + b.y + c.z + d.w;
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const char *name, const float4 *table, uint32_t nrec, int iters, float *out, double bytes_per_lane_iter) {
    const int grid = 256 * 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(64), 0, 0, table, nrec, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(64), 0, 0, table, nrec, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)grid * 64 * iters;
    printf("%-44s table %7.1f MB  %8.3f ms  %7.2f Grec/s  %7.2f TB/s  %6.1f B/clk/CU(2.4GHz)\n", name, nrec * 64.0 / 1e6, ms,
           recs / ms / 1e6, recs * bytes_per_lane_iter / ms / 1e9, recs * bytes_per_lane_iter / (ms * 1e-3) / 256 / 2.4e9);
    fflush(stdout);
}

int main() {
    const uint32_t sizes[] = {128u, 256u, 2048u, 16384u, 300000u};   // 8 KB, 16 KB (L1), 128 KB, 1 MB (L2), 19 MB
    float *out; CK(hipMalloc(&out, 4));
    for (uint32_t nrec : sizes) {
        float4 *table; CK(hipMalloc(&table, (size_t)nrec * 64));
        CK(hipMemset(table, 0, (size_t)nrec * 64));
        const int iters = 256;
        run<0>("per-lane 64 B (4 x dwordx4)", table, nrec, iters, out, 64);
        run<4>("per-lane 32 B (2 x dwordx4)", table, nrec, iters, out, 32);
        run<3>("per-lane 16 B (1 x dwordx4)", table, nrec, iters, out, 16);
        run<1>("quad-coop 64 B, registers", table, nrec, iters, out, 64);
        run<2>("quad-coop 64 B, LDS-DMA + 4 ds_read_b128", table, nrec, iters, out, 64);
        CK(hipFree(table));
    }
    return 0;
}
