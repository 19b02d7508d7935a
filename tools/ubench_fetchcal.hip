// ubench_fetchcal.hip -- calibrates rocprofv3's FETCH_SIZE for the extend kernel's access pattern.
//
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read and
// is "uncalibrated for other access widths: calibrate on a known byte count in your own access pattern".  The extend
// kernel gathers 64-byte node records (4 x dwordx4 per lane) and 48-byte triangle records (3 x dwordx4 per lane) at
// random; this program reads tables far larger than the 256 MiB Infinity Cache exactly once in those patterns, so the
// bytes that must come from HBM are known, and prints them per kernel.  Run it under
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./ubench_fetchcal
// and compare (tools/fetchcal_summary.py): the ratio known bytes / (FETCH_SIZE * 1024) is the correction factor.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fetchcal.hip -o tools/bin/ubench_fetchcal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// record i of the pass -> a permutation of [0, nrec) for nrec a power of two (odd multiplier, xor with a shifted copy)
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) {
    uint32_t x = (i * 0x9E3779B1u) & mask;
    x ^= x >> 15; x = (x * 0x85EBCA6Bu) & mask;     // odd multiplier: still a bijection modulo 2^k; xorshift by >= k/2 is one too
    return x;
}

// each lane reads ONE 64-byte record, chosen at random without repetition, with 4 x dwordx4 (the node fetch)
__global__ __launch_bounds__(64) void gather64_once(const float4 *__restrict__ table, uint32_t nrec, float *out) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= nrec) return;
    const float4 *p = table + (size_t)perm(i, nrec - 1) * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 123.456f) out[0] = s;
}
// each lane reads ONE 48-byte record with 3 x dwordx4 (the triangle fetch)
__global__ __launch_bounds__(64) void gather48_once(const float4 *__restrict__ table, uint32_t nrec, float *out) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= nrec) return;
    const float4 *p = table + (size_t)perm(i, nrec - 1) * 3;
    const float4 a = p[0], b = p[1], c = p[2];
    const float s = a.x + b.y + c.z;
    if (s == 123.456f) out[0] = s;
}
// each lane reads ONE whole 128-byte line with 8 x dwordx4 (both halves of an L2 line in one go: separates "one request
// per line, tallied at 64 B" from "64-byte requests")
__global__ __launch_bounds__(64) void gather128_once(const float4 *__restrict__ table, uint32_t nrec, float *out) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= nrec) return;
    const float4 *p = table + (size_t)perm(i, nrec - 1) * 8;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float4 v = p[k]; s += v.x + v.w; }
    if (s == 123.456f) out[0] = s;
}
// control: the same bytes as one wide coalesced stream (16 B per lane, consecutive lanes consecutive addresses)
__global__ __launch_bounds__(256) void stream16(const float4 *__restrict__ table, size_t n16, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = table[i]; s += v.x + v.w; }
    if (s == 123.456f) out[0] = s;
}
// 256 MiB of writes between the passes, so that nothing of the table survives in the Infinity Cache
__global__ __launch_bounds__(256) void flush(float4 *__restrict__ buf, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) buf[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const uint32_t nrec = 1u << 25;                         // 32 Mi records: 2 GiB of 64-B records, 1.5 GiB of 48-B records
    float4 *table, *scrub; float *out;
    CK(hipMalloc(&table, (size_t)nrec * 64));
    CK(hipMalloc(&scrub, (size_t)1 << 30));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(table, 0, (size_t)nrec * 64));
    auto scrub_caches = [&]() { hipLaunchKernelGGL(flush, dim3(4096), dim3(256), 0, 0, scrub, ((size_t)1 << 30) / 16); CK(hipDeviceSynchronize()); };
    for (int rep = 0; rep < 2; ++rep) {
        scrub_caches();
        hipLaunchKernelGGL(gather64_once, dim3(nrec / 64), dim3(64), 0, 0, table, nrec, out);
        CK(hipDeviceSynchronize());
        scrub_caches();
        hipLaunchKernelGGL(gather48_once, dim3(nrec / 64), dim3(64), 0, 0, table, nrec, out);
        CK(hipDeviceSynchronize());
        scrub_caches();
        hipLaunchKernelGGL(gather128_once, dim3(nrec / 2 / 64), dim3(64), 0, 0, table, nrec / 2, out);
        CK(hipDeviceSynchronize());
        scrub_caches();
        hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, table, (size_t)nrec * 4, out);
        CK(hipDeviceSynchronize());
    }
    printf("known_bytes gather64_once %llu\n", (unsigned long long)nrec * 64ull);
    printf("known_bytes gather48_once %llu\n", (unsigned long long)nrec * 48ull);
    printf("known_bytes gather128_once %llu\n", (unsigned long long)(nrec / 2) * 128ull);
    printf("known_bytes stream16 %llu\n", (unsigned long long)nrec * 64ull);
    return 0;
}
