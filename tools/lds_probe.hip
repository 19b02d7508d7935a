// lds_probe.hip -- how much LDS may a 256-thread workgroup with 32 KB of static LDS take and still run 3 (or 4) per CU?
// build: hipcc --offload-arch=gfx950 -O2 tools/lds_probe.hip -o tools/bin/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int STATIC_WORDS>
__global__ __launch_bounds__(256) void k(float *out) {
    __shared__ float s[STATIC_WORDS];
    extern __shared__ float d[];
    s[threadIdx.x] = out[threadIdx.x]; d[threadIdx.x] = s[threadIdx.x];
    __syncthreads();
    out[threadIdx.x] = s[(threadIdx.x + 1) & 255] + d[(threadIdx.x + 3) & 255];
}
int main() {
    for (int dyn = 16384; dyn <= 24576; dyn += 256) {
        int n = 0;
        hipFuncSetAttribute((const void *)k<8192>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k<8192>, 256, dyn);
        printf("static 32768 + dynamic %5d = %5d B -> %d blocks/CU\n", dyn, 32768 + dyn, n);
    }
    for (int dyn = 6144; dyn <= 9216; dyn += 256) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k<8192>, 256, dyn);
        printf("static 32768 + dynamic %5d = %5d B -> %d blocks/CU\n", dyn, 32768 + dyn, n);
    }
    return 0;
}
