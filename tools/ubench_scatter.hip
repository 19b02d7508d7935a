// ubench_scatter.hip -- how fast can the chip write scattered 64/128-byte records (path-pool headers and
// frames)?  Compares per-lane stores (every lane writes its own record with 4 or 8 dwordx4 stores, so one
// store instruction touches 64 different lines) with cooperative stores (4 or 8 neighbouring lanes write one
// whole record per instruction, so an instruction carries 16 or 8 complete lines).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o tools/bin/ubench_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x;
}

// QUADS = float4 per record (4 = 64 B, 8 = 128 B); COOP = lanes per record in one instruction (1 or QUADS)
template <int QUADS, bool COOP, bool READ>
__global__ __launch_bounds__(256) void scatter_kernel(float4 *__restrict__ table, uint32_t nrec, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rec = mix32(i + 1u) % nrec;            // record of this lane (i < n always: n is a multiple of 256)
    float acc = 0.f;
    if (!COOP) {
        float4 *p = table + (size_t)rec * QUADS;
#pragma unroll
        for (int k = 0; k < QUADS; ++k) {
            if (READ) { float4 v = p[k]; acc += v.x + v.w; }
            else p[k] = make_float4((float)i, (float)k, 1.f, 2.f);
        }
    } else {
        constexpr int PER = 64 / QUADS;                    // records per instruction
#pragma unroll
        for (int j = 0; j < QUADS; ++j) {
            const uint32_t src = j * PER + lane / QUADS;   // lane whose record this lane helps with
            const uint32_t r = __shfl(rec, src);
            float4 *p = table + (size_t)r * QUADS + (lane % QUADS);
            if (READ) { float4 v = *p; acc += v.x + v.w; }
            else *p = make_float4((float)i, (float)j, 1.f, 2.f);
        }
    }
    if (READ && acc == 123.456f) out[0] = acc;
}

template <int QUADS, bool COOP, bool READ>
static void run(const char *name, float4 *table, uint32_t nrec, uint32_t n, float *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((scatter_kernel<QUADS, COOP, READ>), dim3(n / 256), dim3(256), 0, 0, table, nrec, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((scatter_kernel<QUADS, COOP, READ>), dim3(n / 256), dim3(256), 0, 0, table, nrec, n, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-52s %8.1f us  %7.2f Grec/s  %7.2f TB/s  %7.1f G lane-accesses/s\n", name, ms * 1e3, n / ms / 1e6,
           (double)n * QUADS * 16 / ms / 1e9, (double)n * QUADS / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    // default 2 M records = 256 MB (fits the 256 MB Infinity Cache); pass 24 for 16 M records = 2 GB (HBM)
    const int lg = argc > 1 ? atoi(argv[1]) : 21;
    const uint32_t nrec = 1u << lg, n = 1u << lg;         // one access per record slot (with collisions)
    printf("%u records of 128 B\n", nrec);
    float *out; CK(hipMalloc(&out, 4));
    float4 *table; CK(hipMalloc(&table, (size_t)nrec * 128));
    CK(hipMemset(table, 0, (size_t)nrec * 128));
    run<4, false, false>("write 64 B, per lane (4 x dwordx4)", table, nrec, n, out);
    run<4, true, false>("write 64 B, quad-cooperative", table, nrec, n, out);
    run<8, false, false>("write 128 B, per lane (8 x dwordx4)", table, nrec, n, out);
    run<8, true, false>("write 128 B, octet-cooperative", table, nrec, n, out);
    run<4, false, true>("read 64 B, per lane", table, nrec, n, out);
    run<4, true, true>("read 64 B, quad-cooperative", table, nrec, n, out);
    run<8, false, true>("read 128 B, per lane", table, nrec, n, out);
    run<8, true, true>("read 128 B, octet-cooperative", table, nrec, n, out);
    return 0;
}
