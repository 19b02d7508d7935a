// ubench_pk.hip -- issue rate of v_pk_mul_f32 / v_pk_add_f32 vs their scalar forms on gfx950 (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float s) {
    f2 a = {1.0f + threadIdx.x, 2.0f}, b = {0.5f, 0.25f}, c = {3.f, 4.f}, d = {5.f, 6.f};
    f2 m = {s, s};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {          // 4 packed ops = 8 flops-lanes
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else if (MODE == 1) {   // 8 scalar ops
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y) : "v"(s));
            } else if (MODE == 2) {
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            } else {                  // 4 scalar ops (same instruction count as MODE 0)
                asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                             : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(s));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
}

template <int MODE> static void run(const char *name, float *out, int ninstr_per_u) {
    const int grid = 256 * 32, iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)grid * iters * 16 * ninstr_per_u;
    printf("%-28s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / winstr);
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 32 * 64 * 4));
    run<0>("4 x v_pk_mul_f32", out, 4);
    run<1>("8 x v_mul_f32", out, 8);
    run<2>("4 x v_pk_add_f32", out, 4);
    run<3>("4 x v_mul_f32", out, 4);
    return 0;
}
