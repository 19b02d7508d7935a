// Random 64-byte sector reads (four 16-byte loads per thread, as the shade kernels read a frame's first sector) over
// tables of growing size: does the rate fall once the table outgrows the TLB reach?  The path pool's frames are read
// in slot order = random order over up to 45 GB.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_tlb tools/ubench_tlb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
// window = 0: every lane anywhere in the table; else the 64 lanes of a wave read random sectors of one window of that
// many sectors, placed anywhere (a wave whose paths have neighbouring slots)
__global__ __launch_bounds__(256) void probe(const float4 *t, uint64_t sectors, uint32_t per_thread, float *sink, uint64_t window) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, wave = tid >> 6;
    float acc = 0.0f;
    for (uint32_t i = 0; i < per_thread; ++i) {
        uint64_t r = ((uint64_t)mix32(tid * 977u + i) << 20) ^ mix32(tid + i * 7919u);
        if (window) {
            const uint64_t wb = (((uint64_t)mix32(wave * 31u + i * 131u) << 20) ^ mix32(wave + i * 977u)) % (sectors - window);
            r = wb + r % window;
        }
        const float4 *p = t + (r % sectors) * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
int main() {
    float *sink; hipMalloc(&sink, 4);
    for (uint64_t gib : {1ull, 4ull, 16ull, 96ull}) {
        float4 *t = nullptr;
        if (hipMalloc(&t, gib << 30) != hipSuccess) { printf("table %llu GiB: no memory\n", (unsigned long long)gib); continue; }
        hipMemset(t, 0, gib << 30);
        const uint64_t sectors = (gib << 30) / 64;
        const uint32_t wgs = 256 * 8 * 4, per_thread = 64;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (uint64_t window : {0ull, 1ull << 8, 1ull << 10, 1ull << 12, 1ull << 14, 1ull << 18}) {      // sectors: 16 KB .. 16 MB
            hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, t, sectors, per_thread, sink, window);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, t, sectors, per_thread, sink, window);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double n = (double)wgs * 256 * per_thread;
            printf("table %3llu GiB, wave window %8llu KB: %.0f M random 64-byte sectors in %.3f ms = %.0f GB/s, %.1f G sectors/s\n", (unsigned long long)gib,
                   (unsigned long long)(window * 64 / 1024), n / 1e6, ms, n * 64 / ms / 1e6, n / ms / 1e6);
        }
        hipFree(t);
    }
    return 0;
}
