// ubench_atomic.hip -- cost of "one atomic per wave" queue counters: N waves, each wave's leader lane does K
// atomicAdds (u64) on counters that are the same word, neighbouring words of one line, or words 4 KiB apart.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o tools/bin/ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// stride: distance in u64 words between the K counters a wave touches; RET: use the returned value
template <bool RET>
__global__ __launch_bounds__(64) void atomic_kernel(unsigned long long *c, int K, int stride, int spread, unsigned long long *sink,
                                                    int shard_words = 4096) {
    unsigned long long acc = 0;
    if (threadIdx.x == 0) {
        // spread > 1: waves are dealt over `spread` replicas of the counter set (sharded counters)
        unsigned long long *base = c + (size_t)(blockIdx.x % spread) * shard_words;
        for (int k = 0; k < K; ++k) {
            if (RET) acc += atomicAdd(base + (size_t)k * stride, 1ull);
            else atomicAdd(base + (size_t)k * stride, 1ull);
        }
        if (RET && acc == 0xFFFFFFFFFFFFull) sink[0] = acc;
    }
}

__global__ __launch_bounds__(64) void atomic32_kernel(uint32_t *c, int spread, int shard_words, uint32_t *sink) {
    if (threadIdx.x == 0) {
        uint32_t v = atomicAdd(c + (size_t)(blockIdx.x % spread) * shard_words, 1u);
        if (v == 0xFFFFFFFFu) sink[0] = v;
    }
}

static void run32(const char *name, uint32_t *c, int waves, int spread, int shard_words) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(atomic32_kernel, dim3(waves), dim3(64), 0, 0, c, spread, shard_words, c + 100000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(atomic32_kernel, dim3(waves), dim3(64), 0, 0, c, spread, shard_words, c + 100000);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-64s %8.1f us  %6.2f ns per atomic\n", name, ms * 1e3, ms * 1e6 / (double)waves);
    fflush(stdout);
}

template <bool RET>
static void run(const char *name, unsigned long long *c, int waves, int K, int stride, int spread, unsigned long long *sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(atomic_kernel<RET>, dim3(waves), dim3(64), 0, 0, c, K, stride, spread, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(atomic_kernel<RET>, dim3(waves), dim3(64), 0, 0, c, K, stride, spread, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-64s %8.1f us  %6.2f ns per atomic\n", name, ms * 1e3, ms * 1e6 / ((double)waves * K));
    fflush(stdout);
}

int main() {
    unsigned long long *c, *sink;
    CK(hipMalloc(&c, 4096 * 8 * 64)); CK(hipMemset(c, 0, 4096 * 8 * 64));
    CK(hipMalloc(&sink, 8));
    const int waves = 32768;
    run<true>("empty launch (K=0)", c, waves, 0, 0, 1, sink);
    run<true>("1 atomic/wave, same word, returned", c, waves, 1, 0, 1, sink);
    run<false>("1 atomic/wave, same word, not returned", c, waves, 1, 0, 1, sink);
    run<true>("4 atomics/wave, 4 neighbouring words, returned", c, waves, 4, 1, 1, sink);
    run<false>("4 atomics/wave, 4 neighbouring words, not returned", c, waves, 4, 1, 1, sink);
    run<true>("4 atomics/wave, words 4 KiB apart, returned", c, waves, 4, 512, 1, sink);
    run<false>("4 atomics/wave, words 4 KiB apart, not returned", c, waves, 4, 512, 1, sink);
    run<true>("4 atomics/wave, words 256 B apart, returned", c, waves, 4, 32, 1, sink);
    run<true>("1 atomic/wave, 8 sharded counters, returned", c, waves, 1, 0, 8, sink);
    run<true>("1 atomic/wave, 64 sharded counters, returned", c, waves, 1, 0, 64, sink);
    run<true>("4 atomics/wave, neighbouring, 8 shards, returned", c, waves, 4, 1, 8, sink);
    run<true>("4 atomics/wave, neighbouring, 64 shards, returned", c, waves, 4, 1, 64, sink);
    run32("u32: 1 atomic/wave, same word, returned", (uint32_t *)c, waves, 1, 0);
    run32("u32: 16 shards 128 B apart", (uint32_t *)c, waves, 16, 32);
    run32("u32: 16 shards 256 B apart", (uint32_t *)c, waves, 16, 64);
    run32("u32: 16 shards 4352 B apart", (uint32_t *)c, waves, 16, 1088);
    run32("u32: 16 shards 32 KiB apart", (uint32_t *)c, waves, 16, 8192);
    run32("u32: 64 shards 128 B apart", (uint32_t *)c, waves, 64, 32);
    run32("u32: 64 shards 4352 B apart", (uint32_t *)c, waves, 64, 1088);
    return 0;
}
