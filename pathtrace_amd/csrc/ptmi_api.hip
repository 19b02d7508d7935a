// ptmi_api.hip -- C ABI (include/ptmi.h): context, BVH build/upload, batched ray queries.
//
// Kernels here: `extend` (closest hit, replaces LinearBVH.CheckIntersection, linear_bvh.go:116)
// and `connect` (any hit, replaces QuickCheckIntersection, linear_bvh.go:202) over caller-provided
// ray arrays.  One ray per lane, one wave per workgroup, traversal stack in LDS; the extend kernel is
// persistent (lanes are re-filled with fresh rays) and schedules inner-node / triangle steps by vote.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "ptmi_ctx.hpp"

using namespace ptmi;

static thread_local std::string g_last_error;

namespace ptmi {
int set_err(ptmi_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return code;
}
int hip_err(ptmi_ctx *ctx, hipError_t e, const char *what) {
    return set_err(ctx, PTMI_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
int ensure_stage(ptmi_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->stage_bytes) return PTMI_OK;
    if (ctx->d_stage) { (void)hipFree(ctx->d_stage); ctx->d_stage = nullptr; ctx->stage_bytes = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    PTMI_HIP(ctx, hipMalloc(&ctx->d_stage, want));
    ctx->stage_bytes = want;
    return PTMI_OK;
}
}  // namespace ptmi

// ------------------------------------------------------------------------------------ kernels

constexpr int WAVE = 64;
constexpr uint64_t PTMI_MAX_RAYS_PER_CALL = 0x3FFFFFFFull * WAVE;   // the one-shot kernels' grid is n / 64 workgroups

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const TravCounters &c, unsigned long long *counters, int which_query,
                                               uint32_t active) {
    if (!COUNT) return;
    uint32_t vi = c.inner, vt = c.tri, vl = c.leaf, va = active;
    for (int off = 32; off > 0; off >>= 1) {
        vi += __shfl_xor(vi, off); vt += __shfl_xor(vt, off); vl += __shfl_xor(vl, off); va += __shfl_xor(va, off);
    }
    if ((threadIdx.x & 63) == 0) {
        counters = counter_shard(counters);
        atomicAdd(&counters[CNT_INNER], (unsigned long long)vi);
        atomicAdd(&counters[CNT_TRI], (unsigned long long)vt);
        atomicAdd(&counters[CNT_LEAF], (unsigned long long)vl);
        atomicAdd(&counters[which_query], (unsigned long long)va);
        if (which_query == CNT_ANY) {
            atomicAdd(&counters[CNT_INNER_ANY], (unsigned long long)vi);
            atomicAdd(&counters[CNT_TRI_ANY], (unsigned long long)vt);
        }
    }
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void extend_kernel(DevBVH bvh, const float *__restrict__ o,
                                                      const float *__restrict__ d, const float *__restrict__ tmax,
                                                      uint64_t n, int32_t *__restrict__ prim, float *__restrict__ t,
                                                      unsigned long long *counters) {
    __shared__ uint32_t smem[STACK * WAVE];
    const uint64_t r = (uint64_t)blockIdx.x * WAVE + threadIdx.x;
    TravCounters cnt;
    const bool live = r < n;
    if (live) {
        const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2];
        const float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
        int p; float tt;
        closest_hit<WAVE, COUNT>(bvh, smem + threadIdx.x, ox, oy, oz, dx, dy, dz, tmax[r], p, tt, cnt);
        prim[r] = p; t[r] = tt;
    }
    flush_counters<COUNT>(cnt, counters, CNT_CLOSEST, live ? 1u : 0u);
}

// The production extend kernel: persistent waves walking vote-scheduled single steps (vote_walk, trav_device.hpp).
template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void extend_vote_kernel(DevBVH bvh, const float *__restrict__ o,
                                                           const float *__restrict__ d,
                                                           const float *__restrict__ tmax, uint64_t n,
                                                           int32_t *__restrict__ prim, float *__restrict__ t,
                                                           unsigned long long *counters,
                                                           unsigned long long *cursor, int refill_min, int tri_bias,
                                                           uint32_t chunk_rays) {
    __shared__ uint32_t smem[STACK * WAVE];
    TravCounters cnt;
    const uint32_t nrays = vote_walk<WAVE, false, COUNT, uint64_t>(
        bvh, smem + threadIdx.x, n, cursor, refill_min, tri_bias, chunk_rays, cnt,
        [&](uint64_t idx, NextRay &nx, uint64_t &tag) {
            tag = idx;
            nx.ox = o[3 * idx]; nx.oy = o[3 * idx + 1]; nx.oz = o[3 * idx + 2];
            nx.dx = d[3 * idx]; nx.dy = d[3 * idx + 1]; nx.dz = d[3 * idx + 2];
            nx.tmax = tmax[idx];
        },
        [&](uint64_t ray, const RayTrav &r) { prim[ray] = r.best; t[ray] = r.best >= 0 ? r.best_t : 0.0f; });
    flush_counters<COUNT>(cnt, counters, CNT_CLOSEST, nrays);
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void connect_kernel(DevBVH bvh, const float *__restrict__ o,
                                                       const float *__restrict__ d, const float *__restrict__ tmax,
                                                       uint64_t n, uint8_t *__restrict__ occluded,
                                                       unsigned long long *counters) {
    __shared__ uint32_t smem[STACK * WAVE];
    const uint64_t r = (uint64_t)blockIdx.x * WAVE + threadIdx.x;
    TravCounters cnt;
    const bool live = r < n;
    if (live) {
        const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2];
        const float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
        occluded[r] = any_hit<WAVE, COUNT, !COUNT>(bvh, smem + threadIdx.x, ox, oy, oz, dx, dy, dz, tmax[r], cnt) ? 1 : 0;
    }
    flush_counters<COUNT>(cnt, counters, CNT_ANY, live ? 1u : 0u);
}

// The any-hit batch kernel, scheduled like extend_vote_kernel.  The counting variant walks a ray's nodes in the order
// of QuickCheckIntersection, so its work counters equal the reference walk's (test); the plain variant takes the far
// child first (vote_inner_step_any: the occlusion bit does not depend on the order, and occluders are found sooner).
template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void connect_vote_kernel(DevBVH bvh, const float *__restrict__ o,
                                                            const float *__restrict__ d,
                                                            const float *__restrict__ tmax, uint64_t n,
                                                            uint8_t *__restrict__ occluded,
                                                            unsigned long long *counters, unsigned long long *cursor,
                                                            int refill_min, int tri_bias, uint32_t chunk_rays) {
    __shared__ uint32_t smem[STACK * WAVE];
    TravCounters cnt;
    const uint32_t nrays = vote_walk<WAVE, true, COUNT, uint64_t, false, !COUNT>(
        bvh, smem + threadIdx.x, n, cursor, refill_min, tri_bias, chunk_rays, cnt,
        [&](uint64_t idx, NextRay &nx, uint64_t &tag) {
            tag = idx;
            nx.ox = o[3 * idx]; nx.oy = o[3 * idx + 1]; nx.oz = o[3 * idx + 2];
            nx.dx = d[3 * idx]; nx.dy = d[3 * idx + 1]; nx.dz = d[3 * idx + 2];
            nx.tmax = tmax[idx];
        },
        [&](uint64_t ray, const RayTrav &r) { occluded[ray] = r.best ? 1 : 0; });
    flush_counters<COUNT>(cnt, counters, CNT_ANY, nrays);
}

// Where a launch keeps its device-side scratch: the work counters (CNT_SHARDS lines) and this launch's ray cursor.
struct LaunchScratch { unsigned long long *counters, *cursor; int line; };
// The next line of the context's cursor ring.  A line whose previous launch (64 launches ago) has not finished yet is
// waited for: the ring then never hands a live kernel's cursor to a second launch.  Streams under capture record and
// wait for nothing (an event recorded into a graph cannot be waited for from outside it; the capturing caller orders
// its launches itself, as ptmi_closest_hit_dev_noalloc's contract says).
static int ctx_scratch(ptmi_ctx *ctx, LaunchScratch &out) {
    const int line = (int)(ctx->cursor_seq++ % CNT_CURSOR_LINES);
    if (ctx->cursor_pending[line]) {
        PTMI_HIP(ctx, hipEventSynchronize(ctx->cursor_ev[line]));
        ctx->cursor_pending[line] = false;
    }
    out = LaunchScratch{ctx->d_counters, ctx->d_counters + CNT_CURSOR + (size_t)line * CNT_WORDS, line};
    return PTMI_OK;
}
static int scratch_launched(ptmi_ctx *ctx, const LaunchScratch &sc, hipStream_t s) {
    if (sc.line < 0) return PTMI_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return PTMI_OK; }
    if (cap != hipStreamCaptureStatusNone) return PTMI_OK;
    if (!ctx->cursor_ev[sc.line]) PTMI_HIP(ctx, hipEventCreateWithFlags(&ctx->cursor_ev[sc.line], hipEventDisableTiming));
    PTMI_HIP(ctx, hipEventRecord(ctx->cursor_ev[sc.line], s));
    ctx->cursor_pending[sc.line] = true;
    return PTMI_OK;
}

template <int STACK>
static int launch_extend(ptmi_ctx *ctx, const LaunchScratch &sc, const float *o, const float *d, const float *tmax, uint64_t n,
                         int32_t *prim, float *t, hipStream_t s) {
    dim3 grid((unsigned)((n + WAVE - 1) / WAVE)), block(WAVE);
    if (ctx->extend_variant == 0) {      // one-shot kernel (A/B reference, PTMI_EXTEND_VARIANT=0)
        if (ctx->counting)
            hipLaunchKernelGGL((extend_kernel<STACK, true>), grid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, prim, t, sc.counters);
        else
            hipLaunchKernelGGL((extend_kernel<STACK, false>), grid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, prim, t, sc.counters);
        return PTMI_OK;
    }
    // persistent vote-scheduled kernel: one resident round of waves (the LDS stack bounds how many fit)
    const void *fn = ctx->counting ? (const void *)extend_vote_kernel<STACK, true> : (const void *)extend_vote_kernel<STACK, false>;
    const int per_cu = std::min(ctx->waves_per_cu(fn, WAVE), ctx->persist_waves_per_cu);
    PTMI_HIP(ctx, hipMemsetAsync(sc.cursor, 0, sizeof(unsigned long long), s));
    dim3 pgrid((unsigned)std::min<uint64_t>(grid.x, (uint64_t)ctx->num_cus * per_cu));
    if (ctx->counting)
        hipLaunchKernelGGL((extend_vote_kernel<STACK, true>), pgrid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, prim, t,
                           sc.counters, sc.cursor, ctx->vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->vote_chunk);
    else
        hipLaunchKernelGGL((extend_vote_kernel<STACK, false>), pgrid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, prim, t,
                           sc.counters, sc.cursor, ctx->vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->vote_chunk);
    return PTMI_OK;
}
template <int STACK>
static int launch_connect(ptmi_ctx *ctx, const LaunchScratch &sc, const float *o, const float *d, const float *tmax, uint64_t n,
                          uint8_t *occ, hipStream_t s) {
    dim3 grid((unsigned)((n + WAVE - 1) / WAVE)), block(WAVE);
    if (ctx->extend_variant == 0) {      // one-shot kernel (A/B reference, PTMI_EXTEND_VARIANT=0)
        if (ctx->counting)
            hipLaunchKernelGGL((connect_kernel<STACK, true>), grid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, occ, sc.counters);
        else
            hipLaunchKernelGGL((connect_kernel<STACK, false>), grid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, occ, sc.counters);
        return PTMI_OK;
    }
    const void *fn = ctx->counting ? (const void *)connect_vote_kernel<STACK, true> : (const void *)connect_vote_kernel<STACK, false>;
    const int per_cu = std::min(ctx->waves_per_cu(fn, WAVE), ctx->persist_waves_per_cu);
    PTMI_HIP(ctx, hipMemsetAsync(sc.cursor, 0, sizeof(unsigned long long), s));
    dim3 pgrid((unsigned)std::min<uint64_t>(grid.x, (uint64_t)ctx->num_cus * per_cu));
    if (ctx->counting)
        hipLaunchKernelGGL((connect_vote_kernel<STACK, true>), pgrid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, occ,
                           sc.counters, sc.cursor, ctx->vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->vote_chunk);
    else
        hipLaunchKernelGGL((connect_vote_kernel<STACK, false>), pgrid, block, 0, s, ctx->dev_bvh(), o, d, tmax, n, occ,
                           sc.counters, sc.cursor, ctx->vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->vote_chunk);
    return PTMI_OK;
}

// Resident waves per CU of the persistent batch kernels of one stack class, asked for once per context when a tree is
// uploaded (the class follows from the tree's depth): a query call can load a code object, which a caller under
// stream capture -- ptmi_closest_hit_dev_noalloc's audience -- must not trigger, so no batch call makes one.
template <int STACK>
static void resolve_occupancy_t(ptmi_ctx *ctx) {
    for (const void *fn : {(const void *)extend_vote_kernel<STACK, false>, (const void *)extend_vote_kernel<STACK, true>,
                           (const void *)connect_vote_kernel<STACK, false>, (const void *)connect_vote_kernel<STACK, true>})
        (void)ctx->waves_per_cu(fn, WAVE);
}
static void resolve_occupancy(ptmi_ctx *ctx) {
    const int dep = ctx->stack_entries();
    if (dep <= 16) resolve_occupancy_t<16>(ctx);
    else if (dep <= 20) resolve_occupancy_t<20>(ctx);
    else if (dep <= 24) resolve_occupancy_t<24>(ctx);
    else if (dep <= 28) resolve_occupancy_t<28>(ctx);
    else if (dep <= 32) resolve_occupancy_t<32>(ctx);
    else if (dep <= 48) resolve_occupancy_t<48>(ctx);
    else resolve_occupancy_t<64>(ctx);
}

// ------------------------------------------------------------------------------------ allocation-free scratch
// ptmi_closest_hit_dev_noalloc (no counterpart in the reference) serves callers that cannot allocate: code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc.  The persistent extend kernel needs one piece of
// device memory beyond its inputs and outputs -- the ray cursor its waves draw chunks from, which the ordinary entry
// points keep in the hipMalloc'd counter block of the context.  Here it lives in this file-scope array: one 128-byte
// line for the cursor (atomics on one line serialise, so it shares its line with nothing), padded to the array's
// 256-byte alignment.  The size does not depend on the number of rays.  In host code `scratch` is only a stand-in for the
// symbol: ptmi_create asks the runtime for the array's address on the context's device and every use goes through
// that pointer (ctx->noalloc_scratch).  All calls on one GPU share the array: calls that overlap on one GPU race.
constexpr size_t NOALLOC_SCRATCH_WORDS = 2 * CNT_WORDS;          // 256 bytes
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_SCRATCH_WORDS * sizeof(unsigned long long);
static __device__ __align__(256) unsigned long long scratch[NOALLOC_SCRATCH_WORDS];

// ------------------------------------------------------------------------------------ context

extern "C" const char *ptmi_version(void) { return "ptmi 0.3 (gfx950)"; }

extern "C" const char *ptmi_last_error(const ptmi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int ptmi_create(int device, ptmi_ctx **out) {
    if (!out) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_create: out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return set_err(nullptr, PTMI_ERR_NO_DEVICE,
                       std::string("ptmi_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= ndev) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_create: device index out of range");
    ptmi_ctx *ctx = new (std::nothrow) ptmi_ctx();
    if (!ctx) return set_err(nullptr, PTMI_ERR_OOM, "ptmi_create: out of host memory");
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->num_cus = prop.multiProcessorCount;
        // tuning / A-B knobs (defaults are the measured best)
        if (const char *v = getenv("PTMI_EXTEND_VARIANT")) ctx->extend_variant = atoi(v);
        if (const char *v = getenv("PTMI_PERSIST_WAVES_PER_CU")) ctx->persist_waves_per_cu = std::max(1, atoi(v));
        if (const char *v = getenv("PTMI_VOTE_REFILL_MIN")) ctx->vote_refill_min = std::max(1, std::min(64, atoi(v)));
        if (const char *v = getenv("PTMI_VOTE_TRI_BIAS")) ctx->vote_tri_bias = atoi(v);
        if (const char *v = getenv("PTMI_POOL_PATHS")) ctx->pool_paths = std::max(1 << 16, std::min(1 << 28, atoi(v)));
        if (const char *v = getenv("PTMI_POOL_GIB")) ctx->pool_bytes_max = (uint64_t)std::max(1, atoi(v)) << 30;
        if (const char *v = getenv("PTMI_RENDER_LANES")) ctx->render_lanes = atoi(v);
        if (const char *v = getenv("PTMI_SHADE_SPLIT")) ctx->shade_split = atoi(v);
        if (const char *v = getenv("PTMI_RENDER_BATCH_ITERS")) ctx->render_batch_iters = std::max(1, atoi(v));
        if (const char *v = getenv("PTMI_CONNECT_VOTE")) ctx->connect_vote = atoi(v);
        if (const char *v = getenv("PTMI_SEGMENT_SHORTCUT")) ctx->segment_shortcut = atoi(v);
        if (const char *v = getenv("PTMI_RENDER_VOTE_MIN")) ctx->render_vote_min = atoll(v);
        if (const char *v = getenv("PTMI_RENDER_VOTE_REFILL_MIN")) ctx->render_vote_refill_min = std::max(1, std::min(64, atoi(v)));
        if (const char *v = getenv("PTMI_RENDER_VOTE_CHUNK")) ctx->render_vote_chunk = std::max(64, atoi(v) / 64 * 64);
        if (const char *v = getenv("PTMI_EXTEND_VOTE")) ctx->extend_vote = atoi(v);
        if (const char *v = getenv("PTMI_IMPLICIT_FRESH")) ctx->implicit_fresh = atoi(v);
        if (const char *v = getenv("PTMI_SHRINK_GRIDS")) ctx->render_shrink_grids = atoi(v);
        if (const char *v = getenv("PTMI_CHUNK_MSAMPLES")) ctx->chunk_msamples = std::max(1, std::min(1024, atoi(v)));
        if (const char *v = getenv("PTMI_VOTE_CHUNK")) ctx->vote_chunk = std::max(64, atoi(v) / 64 * 64);
    }
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipMalloc((void **)&ctx->d_counters, CNT_TOTAL_WORDS * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(ctx->d_counters, 0, CNT_TOTAL_WORDS * sizeof(unsigned long long))) != hipSuccess) {
        set_err(nullptr, PTMI_ERR_HIP, std::string("ptmi_create: ") + hipGetErrorString(e));
        delete ctx;
        return PTMI_ERR_HIP;
    }
    {   // address of the allocation-free entry point's scratch on THIS device (the device is current: hipSetDevice above)
        unsigned long long *ptr = nullptr;
        if ((e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch))) != hipSuccess || !ptr ||
            (e = hipMemset(ptr, 0, NOALLOC_SCRATCH_BYTES)) != hipSuccess) {
            set_err(nullptr, PTMI_ERR_HIP, std::string("ptmi_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
            ptmi_destroy(ctx);
            return PTMI_ERR_HIP;
        }
        ctx->noalloc_scratch = ptr;
    }
    *out = ctx;
    return PTMI_OK;
}

static void free_bvh_device(ptmi_ctx *ctx) {
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_tris) (void)hipFree(ctx->d_tris);
    ctx->d_nodes = ctx->d_tris = nullptr;
    ctx->have_bvh = false;
}

namespace ptmi { void free_scene(ptmi_ctx *ctx); void invalidate_scene(ptmi_ctx *ctx); }

extern "C" void ptmi_destroy(ptmi_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)ptmi_comm_destroy(ctx);
    free_bvh_device(ctx);
    ptmi::free_scene(ctx);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (auto &sp : ctx->prof_open) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : ctx->prof_free) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->cursor_ev) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

// ------------------------------------------------------------------------------------ BVH

// Packs `lin` and replaces the context's tree with it.  The context only changes once packing has succeeded; from
// then on the old tree is gone, and so is the scene that was uploaded for the old mesh (its per-triangle records are
// sized for that mesh): render calls return PTMI_ERR_NO_SCENE until ptmi_upload_scene is called again.
static int upload_packed(ptmi_ctx *ctx, LinearBVH &lin, const float *verts, uint64_t nverts, const uint32_t *tri_idx, uint64_t ntris) {
    const char *msg = "";
    PackedBVH packed;
    int rc = pack_bvh(lin, verts, tri_idx, packed, &msg);
    if (rc) return set_err(ctx, rc, std::string("pack_bvh: ") + msg);
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    PTMI_HIP(ctx, hipDeviceSynchronize());          // no kernel of this context may still be walking the old tree
    free_bvh_device(ctx);
    ptmi::invalidate_scene(ctx);
    ctx->lin.nodes.swap(lin.nodes);
    ctx->lin.tri_ids.swap(lin.tri_ids);
    size_t nb = packed.nodes.size() * sizeof(PackedNode), tb = packed.tris.size() * sizeof(PackedTri);
    PTMI_HIP(ctx, hipMalloc((void **)&ctx->d_nodes, nb ? nb : 64));
    PTMI_HIP(ctx, hipMalloc((void **)&ctx->d_tris, tb ? tb : 48));
    if (nb) PTMI_HIP(ctx, hipMemcpy(ctx->d_nodes, packed.nodes.data(), nb, hipMemcpyHostToDevice));
    if (tb) PTMI_HIP(ctx, hipMemcpy(ctx->d_tris, packed.tris.data(), tb, hipMemcpyHostToDevice));
    ctx->root_ref = packed.root_ref;
    ctx->bounds_finite = packed.bounds_finite ? 1u : 0u;
    ctx->dummy_tri = packed.dummy_tri;
    {   // tight box: union of the non-empty child boxes (an inverted box holds no triangle)
        float lo[3] = {PT_INF, PT_INF, PT_INF}, hi[3] = {-PT_INF, -PT_INF, -PT_INF};
        bool any = false;
        for (const PackedNode &pn : packed.nodes) {
            const float *ax[3] = {pn.x, pn.y, pn.z};
            for (int c = 0; c < 2; ++c) {
                if (!(ax[0][2 * c] <= ax[0][2 * c + 1] && ax[1][2 * c] <= ax[1][2 * c + 1] && ax[2][2 * c] <= ax[2][2 * c + 1])) continue;
                for (int x = 0; x < 3; ++x) { lo[x] = std::min(lo[x], ax[x][2 * c]); hi[x] = std::max(hi[x], ax[x][2 * c + 1]); }
                any = true;
            }
        }
        float mag = 0.0f;
        for (int x = 0; x < 3; ++x) { ctx->tight_min[x] = lo[x]; ctx->tight_max[x] = hi[x]; mag = std::max(mag, std::max(std::fabs(lo[x]), std::fabs(hi[x]))); }
        // rounding of (bound - o) * inv against `step`: a few ulps of max(|bound|, |o|) <= 1e5 -> < 0.05; plus 1e-3 relative
        ctx->tight_margin = 0.1f + 1e-3f * mag;
        ctx->tight_valid = (ctx->segment_shortcut && any && packed.bounds_finite && !(packed.root_ref & D_REF_LEAF_BIT) && mag <= SEG_OUTSIDE_MAX_ORIGIN) ? 1u : 0u;
    }
    ctx->max_depth = packed.max_depth;
    ctx->packed_inner = packed.nodes.size();
    ctx->packed_tris = packed.tris.size();
    ctx->stats = linear_bvh_stats(ctx->lin);
    ctx->verts.assign(verts, verts + 3 * nverts);
    ctx->tri_idx.assign(tri_idx, tri_idx + 3 * ntris);
    ctx->have_bvh = true;
    resolve_occupancy(ctx);
    return PTMI_OK;
}

struct ptmi_host_bvh {
    LinearBVH lin;
    std::vector<float> verts;
    std::vector<uint32_t> tri_idx;
    double build_ms = 0.0;
};

extern "C" int ptmi_host_bvh_build(const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx, uint64_t ntris,
                                   const float bounds[6], int threshold, int depth, ptmi_host_bvh **out) {
    if (!out) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_host_bvh_build: out is null");
    *out = nullptr;
    ptmi_host_bvh *h = new (std::nothrow) ptmi_host_bvh();
    if (!h) return set_err(nullptr, PTMI_ERR_OOM, "ptmi_host_bvh_build: out of host memory");
    const char *msg = "";
    auto t0 = std::chrono::steady_clock::now();
    int rc = build_linear_bvh(verts_xyz, nverts, tri_idx, ntris, bounds, threshold, depth, h->lin, &msg);
    if (rc) {
        delete h;
        return set_err(nullptr, rc == -5 ? PTMI_ERR_DEPTH : PTMI_ERR_INVALID, std::string("ptmi_host_bvh_build: ") + msg);
    }
    h->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->verts.assign(verts_xyz, verts_xyz + 3 * nverts);
    h->tri_idx.assign(tri_idx, tri_idx + 3 * ntris);
    *out = h;
    return PTMI_OK;
}

extern "C" void ptmi_host_bvh_free(ptmi_host_bvh *bvh) { delete bvh; }

extern "C" int ptmi_host_bvh_sizes(const ptmi_host_bvh *bvh, uint64_t *n_nodes, uint64_t *n_tri_ids) {
    if (!bvh || !n_nodes || !n_tri_ids) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_host_bvh_sizes: null argument");
    *n_nodes = bvh->lin.nodes.size();
    *n_tri_ids = bvh->lin.tri_ids.size();
    return PTMI_OK;
}

static void export_linear(const LinearBVH &lin, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_offset,
                          uint32_t *tri_count, uint32_t *second_child, uint32_t *tri_ids) {
    for (size_t i = 0; i < lin.nodes.size(); ++i) {
        const LinNode &n = lin.nodes[i];
        for (int a = 0; a < 3; ++a) { node_bounds[6 * i + a] = n.minb[a]; node_bounds[6 * i + 3 + a] = n.maxb[a]; }
        is_leaf[i] = n.is_leaf; tri_offset[i] = n.tri_offset; tri_count[i] = n.tri_count; second_child[i] = n.second_child;
    }
    if (!lin.tri_ids.empty()) std::memcpy(tri_ids, lin.tri_ids.data(), lin.tri_ids.size() * sizeof(uint32_t));
}

extern "C" int ptmi_host_bvh_get(const ptmi_host_bvh *bvh, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_offset,
                                 uint32_t *tri_count, uint32_t *second_child, uint32_t *tri_ids) {
    if (!bvh || !node_bounds || !is_leaf || !tri_offset || !tri_count || !second_child || !tri_ids)
        return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_host_bvh_get: null argument");
    export_linear(bvh->lin, node_bounds, is_leaf, tri_offset, tri_count, second_child, tri_ids);
    return PTMI_OK;
}

static void fill_stats(const LinearBVH &lin, double build_ms, uint64_t packed_inner, PtBvhStats *out) {
    BvhStats s = linear_bvh_stats(lin);
    out->max_depth = s.max_depth; out->max_tris = s.max_tris; out->min_tris = s.min_tris;
    out->total_leafs = s.total_leafs; out->total_nodes = s.total_nodes; out->total_triangles = s.total_triangles;
    out->linear_nodes = (int64_t)lin.nodes.size();
    out->packed_inner_nodes = (int64_t)packed_inner;
    out->build_ms = build_ms;
}

extern "C" int ptmi_host_bvh_stats(const ptmi_host_bvh *bvh, PtBvhStats *out) {
    if (!bvh || !out) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_host_bvh_stats: null argument");
    uint64_t inner = 0;
    for (const LinNode &n : bvh->lin.nodes) inner += n.is_leaf ? 0 : 1;
    fill_stats(bvh->lin, bvh->build_ms, inner, out);
    return PTMI_OK;
}

extern "C" int ptmi_upload_host_bvh(ptmi_ctx *ctx, const ptmi_host_bvh *bvh) {
    if (!ctx || !bvh) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_host_bvh: null argument");
    LinearBVH lin = bvh->lin;
    int rc = upload_packed(ctx, lin, bvh->verts.data(), bvh->verts.size() / 3, bvh->tri_idx.data(), bvh->tri_idx.size() / 3);
    if (rc == PTMI_OK) ctx->build_ms = bvh->build_ms;
    return rc;
}

extern "C" int ptmi_build_bvh(ptmi_ctx *ctx, const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx,
                              uint64_t ntris, const float bounds[6], int threshold, int depth) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_build_bvh: ctx is null");
    const char *msg = "";
    auto t0 = std::chrono::steady_clock::now();
    LinearBVH lin;
    int rc = build_linear_bvh(verts_xyz, nverts, tri_idx, ntris, bounds, threshold, depth, lin, &msg);
    if (rc) return set_err(ctx, rc == -5 ? PTMI_ERR_DEPTH : PTMI_ERR_INVALID, std::string("ptmi_build_bvh: ") + msg);
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = upload_packed(ctx, lin, verts_xyz, nverts, tri_idx, ntris);
    if (rc == PTMI_OK) ctx->build_ms = build_ms;
    return rc;
}

extern "C" int ptmi_upload_bvh(ptmi_ctx *ctx, const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx,
                               uint64_t ntris, uint64_t n_nodes, const float *node_bounds, const uint8_t *is_leaf,
                               const uint32_t *tri_offset, const uint32_t *tri_count, const uint32_t *second_child,
                               uint64_t n_tri_ids, const uint32_t *tri_ids) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_upload_bvh: ctx is null");
    if (!verts_xyz || !tri_idx || !node_bounds || !is_leaf || !tri_offset || !tri_count || !second_child ||
        (!tri_ids && n_tri_ids))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_bvh: null array");
    for (uint64_t i = 0; i < 3 * ntris; ++i)
        if (tri_idx[i] >= nverts) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_bvh: triangle index out of range");
    LinearBVH lin;
    lin.nodes.resize(n_nodes);
    for (uint64_t i = 0; i < n_nodes; ++i) {
        LinNode &n = lin.nodes[i];
        for (int a = 0; a < 3; ++a) { n.minb[a] = node_bounds[6 * i + a]; n.maxb[a] = node_bounds[6 * i + 3 + a]; }
        n.is_leaf = is_leaf[i] ? 1 : 0;
        n.tri_offset = tri_offset[i]; n.tri_count = tri_count[i]; n.second_child = second_child[i];
    }
    lin.tri_ids.assign(tri_ids, tri_ids + n_tri_ids);
    const char *msg = "";
    int rc = validate_linear_bvh(lin, ntris, &msg);
    if (rc) return set_err(ctx, rc == -5 ? PTMI_ERR_DEPTH : PTMI_ERR_INVALID, std::string("ptmi_upload_bvh: ") + msg);
    rc = upload_packed(ctx, lin, verts_xyz, nverts, tri_idx, ntris);
    if (rc == PTMI_OK) ctx->build_ms = 0.0;
    return rc;
}

extern "C" int ptmi_bvh_sizes(ptmi_ctx *ctx, uint64_t *n_nodes, uint64_t *n_tri_ids) {
    if (!ctx || !n_nodes || !n_tri_ids) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_bvh_sizes: null argument");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_bvh_sizes: no BVH");
    *n_nodes = ctx->lin.nodes.size();
    *n_tri_ids = ctx->lin.tri_ids.size();
    return PTMI_OK;
}

extern "C" int ptmi_get_linear_bvh(ptmi_ctx *ctx, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_offset,
                                   uint32_t *tri_count, uint32_t *second_child, uint32_t *tri_ids) {
    if (!ctx || !node_bounds || !is_leaf || !tri_offset || !tri_count || !second_child || !tri_ids)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_get_linear_bvh: null argument");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_get_linear_bvh: no BVH");
    export_linear(ctx->lin, node_bounds, is_leaf, tri_offset, tri_count, second_child, tri_ids);
    return PTMI_OK;
}

extern "C" int ptmi_bvh_stats(ptmi_ctx *ctx, PtBvhStats *out) {
    if (!ctx || !out) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_bvh_stats: null argument");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_bvh_stats: no BVH");
    fill_stats(ctx->lin, ctx->build_ms, ctx->packed_inner, out);
    return PTMI_OK;
}

// ------------------------------------------------------------------------------------ queries

static int dispatch_extend(ptmi_ctx *ctx, const LaunchScratch &sc, const float *o, const float *d, const float *tmax, uint64_t n,
                           int32_t *prim, float *t, hipStream_t s) {
    if (n > PTMI_MAX_RAYS_PER_CALL) return set_err(ctx, PTMI_ERR_INVALID, "too many rays in one call");
    const int dep = ctx->stack_entries();
    int rc;
    if (dep <= 16) rc = launch_extend<16>(ctx, sc, o, d, tmax, n, prim, t, s);
    else if (dep <= 20) rc = launch_extend<20>(ctx, sc, o, d, tmax, n, prim, t, s);
    else if (dep <= 24) rc = launch_extend<24>(ctx, sc, o, d, tmax, n, prim, t, s);
    else if (dep <= 28) rc = launch_extend<28>(ctx, sc, o, d, tmax, n, prim, t, s);      // 1-8 M-triangle trees: 22 waves per CU instead of 20
    else if (dep <= 32) rc = launch_extend<32>(ctx, sc, o, d, tmax, n, prim, t, s);
    else if (dep <= 48) rc = launch_extend<48>(ctx, sc, o, d, tmax, n, prim, t, s);
    else rc = launch_extend<64>(ctx, sc, o, d, tmax, n, prim, t, s);
    if (rc) return rc;
    PTMI_HIP(ctx, hipGetLastError());
    return PTMI_OK;
}
static int dispatch_connect(ptmi_ctx *ctx, const LaunchScratch &sc, const float *o, const float *d, const float *tmax, uint64_t n,
                            uint8_t *occ, hipStream_t s) {
    if (n > PTMI_MAX_RAYS_PER_CALL) return set_err(ctx, PTMI_ERR_INVALID, "too many rays in one call");
    const int dep = ctx->stack_entries();
    int rc;
    if (dep <= 16) rc = launch_connect<16>(ctx, sc, o, d, tmax, n, occ, s);
    else if (dep <= 20) rc = launch_connect<20>(ctx, sc, o, d, tmax, n, occ, s);
    else if (dep <= 24) rc = launch_connect<24>(ctx, sc, o, d, tmax, n, occ, s);
    else if (dep <= 28) rc = launch_connect<28>(ctx, sc, o, d, tmax, n, occ, s);
    else if (dep <= 32) rc = launch_connect<32>(ctx, sc, o, d, tmax, n, occ, s);
    else if (dep <= 48) rc = launch_connect<48>(ctx, sc, o, d, tmax, n, occ, s);
    else rc = launch_connect<64>(ctx, sc, o, d, tmax, n, occ, s);
    if (rc) return rc;
    PTMI_HIP(ctx, hipGetLastError());
    return PTMI_OK;
}

extern "C" int ptmi_closest_hit_dev(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax,
                                    uint64_t n, int32_t *prim_id, float *t, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_closest_hit_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_closest_hit_dev: no BVH uploaded");
    if (n == 0) return PTMI_OK;
    if (!o_xyz || !d_xyz || !tmax || !prim_id || !t) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_closest_hit_dev: null buffer");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    LaunchScratch sc;
    int rc = ctx_scratch(ctx, sc);
    if (rc) return rc;
    if ((rc = dispatch_extend(ctx, sc, o_xyz, d_xyz, tmax, n, prim_id, t, (hipStream_t)stream))) return rc;
    return scratch_launched(ctx, sc, (hipStream_t)stream);
}

// Allocation-free form of ptmi_closest_hit_dev: the same kernels, the ray cursor in the static scratch array.  Calls no
// hipMalloc / hipFree (nor any routine that does): a memset of the scratch and one kernel launch on `stream`.
extern "C" int ptmi_closest_hit_dev_noalloc(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax,
                                            uint64_t n, int32_t *prim_id, float *t, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_closest_hit_dev_noalloc: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_closest_hit_dev_noalloc: no BVH uploaded");
    if (n > (uint64_t)PTMI_NOALLOC_MAX_RAYS)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_closest_hit_dev_noalloc: " + std::to_string(n) + " rays in one call, the maximum is " +
                                              std::to_string((uint64_t)PTMI_NOALLOC_MAX_RAYS) + " (PTMI_NOALLOC_MAX_RAYS)");
    if (n == 0) return PTMI_OK;
    if (!o_xyz || !d_xyz || !tmax || !prim_id || !t) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_closest_hit_dev_noalloc: null buffer");
    if (!ctx->noalloc_scratch) return set_err(ctx, PTMI_ERR_HIP, "ptmi_closest_hit_dev_noalloc: scratch address not resolved");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // the array keeps the last call's contents: the whole of it starts every call at zero (the launchers clear the
    // cursor word again, as they do for the ring)
    PTMI_HIP(ctx, hipMemsetAsync(ctx->noalloc_scratch, 0, NOALLOC_SCRATCH_BYTES, s));
    // work counting selects kernels that add to the context's hipMalloc'd counter block, which is still valid memory
    // here (allocated at ptmi_create); only the cursor moves into the scratch
    const LaunchScratch sc{ctx->d_counters, ctx->noalloc_scratch, -1};
    return dispatch_extend(ctx, sc, o_xyz, d_xyz, tmax, n, prim_id, t, s);
}

extern "C" int ptmi_any_hit_dev(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                                uint8_t *occluded, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_any_hit_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_any_hit_dev: no BVH uploaded");
    if (n == 0) return PTMI_OK;
    if (!o_xyz || !d_xyz || !tmax || !occluded) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_any_hit_dev: null buffer");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    LaunchScratch sc;
    int rc = ctx_scratch(ctx, sc);
    if (rc) return rc;
    if ((rc = dispatch_connect(ctx, sc, o_xyz, d_xyz, tmax, n, occluded, (hipStream_t)stream))) return rc;
    return scratch_launched(ctx, sc, (hipStream_t)stream);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// The host-pointer forms move 40 (29) bytes per ray over the link, more time than the walk itself takes, so the batch
// goes through in chunks on two streams: while one chunk is walked, the next one's rays are on their way in and the
// previous one's answers on their way out (each direction has its own DMA engines).  Every chunk has its own part of
// the staging buffer and its own cursor line, so the two streams never share state.  That is for page-locked caller
// buffers (16.7 M rays: 15 ms = 45 GB/s over the link, against 35 ms unpipelined).
constexpr uint64_t HOST_CHUNK_RAYS = 1ull << 21;

template <typename Launch, typename CopyOut>
static int host_batch(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n, size_t out_bytes,
                      Launch launch, CopyOut copy_out) {
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->stream2) PTMI_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    // out_bytes: what the caller's layout of the staged answers takes in all (its own alignment padding included)
    const size_t b3 = align_up(n * 12, 256), b1 = align_up(n * 4, 256), bo = align_up(out_bytes, 256);
    int rc = ensure_stage(ctx, 2 * b3 + b1 + bo);
    if (rc) return rc;
    char *base = (char *)ctx->d_stage;
    float *dO = (float *)base, *dD = (float *)(base + b3), *dT = (float *)(base + 2 * b3);
    char *dOut = base + 2 * b3 + b1;
    // pageable memory gains nothing from chunks (the runtime stages it synchronously, and many small staged copies are
    // slower than a few large ones: measured -15 %): it goes through as one chunk, as before
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, o_xyz) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    const uint64_t chunk = pinned ? HOST_CHUNK_RAYS : n;
    int k = 0;
    for (uint64_t first = 0; first < n; first += chunk, ++k) {
        const uint64_t m = std::min<uint64_t>(chunk, n - first);
        hipStream_t s = (k & 1) ? ctx->stream2 : ctx->stream;
        PTMI_HIP(ctx, hipMemcpyAsync(dO + 3 * first, o_xyz + 3 * first, m * 12, hipMemcpyHostToDevice, s));
        PTMI_HIP(ctx, hipMemcpyAsync(dD + 3 * first, d_xyz + 3 * first, m * 12, hipMemcpyHostToDevice, s));
        PTMI_HIP(ctx, hipMemcpyAsync(dT + first, tmax + first, m * 4, hipMemcpyHostToDevice, s));
        rc = launch(dO + 3 * first, dD + 3 * first, dT + first, m, dOut, first, s);
        if (rc) break;
        rc = copy_out(dOut, first, m, s);
        if (rc) break;
    }
    hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = hipStreamSynchronize(ctx->stream2);
    if (rc) return rc;
    PTMI_HIP(ctx, e1);
    PTMI_HIP(ctx, e2);
    return PTMI_OK;
}

extern "C" int ptmi_closest_hit(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                                int32_t *prim_id, float *t) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_closest_hit: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_closest_hit: no BVH uploaded");
    if (n == 0) return PTMI_OK;
    if (!o_xyz || !d_xyz || !tmax || !prim_id || !t) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_closest_hit: null buffer");
    const size_t half = align_up(n * 4, 256);          // staged answers: prim ids, then t (each padded to 256 bytes)
    return host_batch(
        ctx, o_xyz, d_xyz, tmax, n, 2 * half,
        [&](const float *dO, const float *dD, const float *dT, uint64_t m, char *dOut, uint64_t first, hipStream_t s) {
            LaunchScratch sc;
            int r = ctx_scratch(ctx, sc);
            if (r) return r;
            if ((r = dispatch_extend(ctx, sc, dO, dD, dT, m, (int32_t *)dOut + first, (float *)(dOut + half) + first, s))) return r;
            return scratch_launched(ctx, sc, s);
        },
        [&](char *dOut, uint64_t first, uint64_t m, hipStream_t s) -> int {
            PTMI_HIP(ctx, hipMemcpyAsync(prim_id + first, (int32_t *)dOut + first, m * 4, hipMemcpyDeviceToHost, s));
            PTMI_HIP(ctx, hipMemcpyAsync(t + first, (float *)(dOut + half) + first, m * 4, hipMemcpyDeviceToHost, s));
            return PTMI_OK;
        });
}

extern "C" int ptmi_any_hit(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                            uint8_t *occluded) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_any_hit: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_any_hit: no BVH uploaded");
    if (n == 0) return PTMI_OK;
    if (!o_xyz || !d_xyz || !tmax || !occluded) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_any_hit: null buffer");
    return host_batch(
        ctx, o_xyz, d_xyz, tmax, n, n,
        [&](const float *dO, const float *dD, const float *dT, uint64_t m, char *dOut, uint64_t first, hipStream_t s) {
            LaunchScratch sc;
            int r = ctx_scratch(ctx, sc);
            if (r) return r;
            if ((r = dispatch_connect(ctx, sc, dO, dD, dT, m, (uint8_t *)dOut + first, s))) return r;
            return scratch_launched(ctx, sc, s);
        },
        [&](char *dOut, uint64_t first, uint64_t m, hipStream_t s) -> int {
            PTMI_HIP(ctx, hipMemcpyAsync(occluded + first, (uint8_t *)dOut + first, m, hipMemcpyDeviceToHost, s));
            return PTMI_OK;
        });
}

// ------------------------------------------------------------------------------------ counters

int ptmi::read_counters(ptmi_ctx *ctx, unsigned long long *h, hipStream_t s) {
    static_assert(CNT_WORDS * sizeof(unsigned long long) == 128, "a counter shard is one 128-byte line");
    std::vector<unsigned long long> all((size_t)CNT_SHARDS * CNT_WORDS);
    PTMI_HIP(ctx, hipMemcpyAsync(all.data(), ctx->d_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    PTMI_HIP(ctx, hipStreamSynchronize(s));
    for (int w = 0; w < CNT_WORDS; ++w) h[w] = 0;
    for (int sh = 0; sh < CNT_SHARDS; ++sh)
        for (int w = 0; w < CNT_WORDS; ++w) {
            if (w == CNT_ERROR) h[w] |= all[(size_t)sh * CNT_WORDS + w];
            else h[w] += all[(size_t)sh * CNT_WORDS + w];
        }
    return PTMI_OK;
}

extern "C" int ptmi_set_counting(ptmi_ctx *ctx, int enabled) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_set_counting: ctx is null");
    ctx->counting = enabled != 0;
    return PTMI_OK;
}

extern "C" int ptmi_reset_counters(ptmi_ctx *ctx) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_reset_counters: ctx is null");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    PTMI_HIP(ctx, hipDeviceSynchronize());
    PTMI_HIP(ctx, hipMemset(ctx->d_counters, 0, CNT_TOTAL_WORDS * sizeof(unsigned long long)));
    return PTMI_OK;
}

extern "C" int ptmi_counters(ptmi_ctx *ctx, PtCounters *out) {
    if (!ctx || !out) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_counters: null argument");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    PTMI_HIP(ctx, hipDeviceSynchronize());
    unsigned long long h[CNT_WORDS];
    int rc = ptmi::read_counters(ctx, h, nullptr);
    if (rc) return rc;
    out->closest_queries = h[CNT_CLOSEST]; out->any_queries = h[CNT_ANY]; out->inner_visits = h[CNT_INNER];
    out->tri_tests = h[CNT_TRI]; out->leaf_visits = h[CNT_LEAF]; out->diffuse_events = h[CNT_DIFFUSE];
    out->samples = h[CNT_SAMPLES];
    out->any_inner_visits = h[CNT_INNER_ANY]; out->any_tri_tests = h[CNT_TRI_ANY];
    out->shade_ret_visits = h[CNT_VISIT_RET]; out->shade_diff_visits = h[CNT_VISIT_DIFF]; out->shade_general_visits = h[CNT_VISIT_GEN];
    out->frames_read = h[CNT_FRAMES]; out->extend_entries = h[CNT_ENTRIES];
    return PTMI_OK;
}
