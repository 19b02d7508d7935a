// ptmi_comm.hip -- multi-GPU behind the C ABI: one RCCL communicator per context, the framebuffer reduce, and a
// frame rendered over all ranks in one call (include/ptmi.h, "multi-GPU").
//
// The reference's 16 tile workers share one address space (main.go:472-491, :935-937: unsynchronised writes to disjoint
// Pixels); across GPUs the same disjoint regions are put together by ONE collective per frame -- ncclReduce(sum) of the
// zero-initialised full-frame accumulators (x + 0 == x, so the reduced frame is the single-GPU frame bit for bit).
// BVH and scene are replicated; there is no other traffic.  RCCL runs over xGMI between the GPUs of a node: a 4K frame
// is 99.5 MB + 33 MB of counts, about a millisecond on a ring, once per frame.
//
// RCCL is bound at run time (dlopen): libptmi.so does not link librccl (573 MB) for single-GPU users, and a process
// that already holds an RCCL (PyTorch's) shares it instead of loading a second copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "ptmi_ctx.hpp"

using namespace ptmi;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error, origin;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    Rccl &r = g_rccl;
    const char *override_path = getenv("PTMI_RCCL_LIB");
    if (override_path && *override_path) {
        r.handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        r.origin = override_path;
        if (!r.handle) { r.error = std::string("dlopen(") + override_path + "): " + dlerror(); return; }
    } else {
        // an RCCL the process already holds (PyTorch links one), else the ROCm installation's
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            if ((r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) { r.origin = std::string(name) + " (already loaded)"; break; }
        }
        if (!r.handle)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                if ((r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) { r.origin = name; break; }
            }
        if (!r.handle) { r.error = std::string("librccl.so not found: ") + dlerror(); return; }
    }
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(r.handle, name);
        if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
}

// 0 when RCCL is usable, else an error code with the reason in the context / thread error text
int need_rccl(ptmi_ctx *ctx) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) return set_err(ctx, PTMI_ERR_COMM, "RCCL unavailable: " + g_rccl.error);
    return PTMI_OK;
}

int nccl_err(ptmi_ctx *ctx, ncclResult_t e, const char *what) {
    return set_err(ctx, PTMI_ERR_COMM, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error"));
}

#define PTMI_NCCL(ctx, call)                                             \
    do {                                                                 \
        ncclResult_t e__ = (call);                                       \
        if (e__ != ncclSuccess) return nccl_err((ctx), e__, #call);      \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PTMI_UNIQUE_ID_BYTES, "PTMI_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

}  // namespace

extern "C" int ptmi_comm_unique_id(uint8_t id[PTMI_UNIQUE_ID_BYTES]) {
    if (!id) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_comm_unique_id: id is null");
    int rc = need_rccl(nullptr);
    if (rc) return rc;
    ncclUniqueId u;
    PTMI_NCCL(nullptr, g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return PTMI_OK;
}

extern "C" int ptmi_comm_destroy(ptmi_ctx *ctx) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_comm_destroy: ctx is null");
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0; ctx->comm_nranks = 1;
    return PTMI_OK;
}

extern "C" int ptmi_comm_init_rank(ptmi_ctx *ctx, int nranks, int rank, const uint8_t id[PTMI_UNIQUE_ID_BYTES]) {
    if (!ctx || !id) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_comm_init_rank: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_comm_init_rank: need 0 <= rank < nranks");
    int rc = need_rccl(ctx);
    if (rc) return rc;
    ptmi_comm_destroy(ctx);
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    PTMI_NCCL(ctx, g_rccl.CommInitRank(&comm, nranks, u, rank));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_nranks = nranks;
    return PTMI_OK;
}

// One process driving n contexts (the Go host: a goroutine per GPU): rank i = ctxs[i].  The communicators are created
// inside one RCCL group, so a single thread can make the call.
extern "C" int ptmi_comm_init_all(ptmi_ctx **ctxs, int n) {
    if (!ctxs || n < 1) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_comm_init_all: need n >= 1 contexts");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_comm_init_all: null context");
        for (int j = 0; j < i; ++j)
            if (ctxs[j]->device == ctxs[i]->device)
                return set_err(ctxs[i], PTMI_ERR_INVALID, "ptmi_comm_init_all: two contexts on one GPU (RCCL wants one rank per device)");
    }
    int rc = need_rccl(ctxs[0]);
    if (rc) return rc;
    ncclUniqueId u;
    PTMI_NCCL(ctxs[0], g_rccl.GetUniqueId(&u));
    for (int i = 0; i < n; ++i) ptmi_comm_destroy(ctxs[i]);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);
    // On any failure after GroupStart the group is closed, every communicator it did create is destroyed and the caller's
    // device is current again: the contexts are left without communicators (worlds of one), never half-initialised.
    auto fail = [&](ptmi_ctx *at, ncclResult_t e, const char *what, bool group_open) {
        if (group_open) (void)g_rccl.GroupEnd();
        for (ncclComm_t c : comms) if (c) (void)g_rccl.CommDestroy(c);
        if (dev_before >= 0) (void)hipSetDevice(dev_before);
        return nccl_err(at, e, what);
    };
    ncclResult_t e = g_rccl.GroupStart();
    if (e != ncclSuccess) return fail(ctxs[0], e, "ncclGroupStart", false);
    for (int i = 0; i < n; ++i) {
        hipError_t he = hipSetDevice(ctxs[i]->device);
        e = he == hipSuccess ? g_rccl.CommInitRank(&comms[(size_t)i], n, u, i) : ncclUnhandledCudaError;
        if (e != ncclSuccess) return fail(ctxs[i], e, "ncclCommInitRank", true);
    }
    e = g_rccl.GroupEnd();
    if (e != ncclSuccess) return fail(ctxs[0], e, "ncclGroupEnd", false);
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[(size_t)i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_nranks = n; }
    if (dev_before >= 0) (void)hipSetDevice(dev_before);
    return PTMI_OK;
}

extern "C" int ptmi_comm_rank(const ptmi_ctx *ctx, int *rank, int *nranks) {
    if (!ctx || !rank || !nranks) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_comm_rank: null argument");
    *rank = ctx->comm_rank; *nranks = ctx->comm_nranks;
    return PTMI_OK;
}

extern "C" int ptmi_reduce_frame(ptmi_ctx *ctx, const float *send_accum, const uint32_t *send_counts, float *recv_accum,
                                 uint32_t *recv_counts, int32_t width, int32_t height, int root, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_reduce_frame: ctx is null");
    if (!send_accum || !send_counts || width <= 0 || height <= 0) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_reduce_frame: bad argument");
    if (root < 0 || root >= ctx->comm_nranks) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_reduce_frame: root out of range");
    if (ctx->comm_rank == root && (!recv_accum || !recv_counts)) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_reduce_frame: the root needs receive buffers");
    const size_t npix = (size_t)width * (size_t)height;
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->comm) {               // no communicator: a world of one
        if (ctx->comm_nranks != 1) return set_err(ctx, PTMI_ERR_COMM, "ptmi_reduce_frame: no communicator");
        if (recv_accum != send_accum) PTMI_HIP(ctx, hipMemcpyAsync(recv_accum, send_accum, npix * 12, hipMemcpyDeviceToDevice, s));
        if (recv_counts != send_counts) PTMI_HIP(ctx, hipMemcpyAsync(recv_counts, send_counts, npix * 4, hipMemcpyDeviceToDevice, s));
        return PTMI_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // both reductions in one group: one launch, one pass over the ring
    PTMI_NCCL(ctx, g_rccl.GroupStart());
    ncclResult_t e1 = g_rccl.Reduce(send_accum, recv_accum, npix * 3, ncclFloat32, ncclSum, root, comm, s);
    ncclResult_t e2 = e1 == ncclSuccess ? g_rccl.Reduce(send_counts, recv_counts, npix, ncclUint32, ncclSum, root, comm, s) : e1;
    ncclResult_t e3 = g_rccl.GroupEnd();
    if (e1 != ncclSuccess) return nccl_err(ctx, e1, "ncclReduce(accum)");
    if (e2 != ncclSuccess) return nccl_err(ctx, e2, "ncclReduce(counts)");
    if (e3 != ncclSuccess) return nccl_err(ctx, e3, "ncclGroupEnd");
    return PTMI_OK;
}

extern "C" int ptmi_render_sharded(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t row_block,
                                   uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts, int root,
                                   void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_sharded: ctx is null");
    if (!params || !accum_rgb || !counts) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_sharded: null argument");
    if (params->width <= 0 || params->height <= 0) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_sharded: empty image");
    if (root < 0 || root >= ctx->comm_nranks) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_sharded: root out of range");
    const size_t npix = (size_t)params->width * (size_t)params->height;
    hipStream_t s = (hipStream_t)stream;
    // From here on every path leads through the reduce: the other ranks are already on their way into it and would
    // wait for ever for a rank that left early.  The first error (set-device, the clears, the render) is kept and
    // returned after the reduce; the failing rank joins with whatever its buffers hold and the frame is not valid.
    int rc = PTMI_OK;
    std::string first_err;
    auto note = [&](int r) { if (r && !rc) { rc = r; first_err = ctx->err; } };
    auto hip = [&](hipError_t e, const char *what) { if (e != hipSuccess) note(ptmi::hip_err(ctx, e, what)); };
    hip(hipSetDevice(ctx->device), "hipSetDevice");
    // every rank's frame starts at zero: its own rows get this call's samples, the others stay 0 for the sum
    if (!rc) hip(hipMemsetAsync(accum_rgb, 0, npix * 12, s), "hipMemsetAsync(accum)");
    if (!rc) hip(hipMemsetAsync(counts, 0, npix * 4, s), "hipMemsetAsync(counts)");
    if (!rc) note(ptmi_render_rows_dev(ctx, camera, params, row_block, ctx->comm_rank, ctx->comm_nranks, spp_begin, spp_count,
                                       accum_rgb, counts, stream));
    if (ctx->comm_nranks == 1) return rc;
    int rr = ptmi_reduce_frame(ctx, accum_rgb, counts, accum_rgb, counts, params->width, params->height, root, stream);
    if (rc) { ctx->err = first_err; return rc; }
    return rr;
}
