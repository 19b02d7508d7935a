// ptmi_ctx.hpp -- per-GPU context behind the C ABI (include/ptmi.h).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ptmi.h"
#include "bvh_build.hpp"
#include "trav_device.hpp"

// indices into the device counter block (unsigned long long each)
enum {
    CNT_CLOSEST = 0, CNT_ANY = 1, CNT_INNER = 2, CNT_TRI = 3, CNT_LEAF = 4, CNT_DIFFUSE = 5, CNT_SAMPLES = 6,
    CNT_ERROR = 8,         // bit set of integrator self-check failures (ptmi_render.hip: ERR_*), must stay 0
    CNT_INNER_ANY = 9, CNT_TRI_ANY = 10,                         // the any-hit walks' share of CNT_INNER / CNT_TRI
    CNT_VISIT_RET = 11, CNT_VISIT_DIFF = 12, CNT_VISIT_GEN = 13, // queue entries shaded, by kernel
    CNT_FRAMES = 14,                                             // parked frames read back
    CNT_ENTRIES = 15,                                            // extend-queue entries traced
    CNT_WORDS = 16,        // one shard = 16 words = one 128-byte line
    // Atomics on one line are served one at a time, ~12 ns each, whichever word they address (measured,
    // tools/ubench_atomic.hip): 20 000 waves adding to one counter block take 0.25 ms for that alone.  The block is
    // therefore replicated CNT_SHARDS times, a workgroup adds to shard blockIdx % CNT_SHARDS and the host sums.
    CNT_SHARDS = 64,
    // Ray cursors of the persistent batch kernels: a ring of lines, one line per launch.  _dev calls are asynchronous
    // on caller-chosen streams, so two launches of one context can overlap on the device; each takes the next line
    // of the ring (cleared on its own stream just before the launch), so up to CNT_CURSOR_LINES launches of a context
    // may be in flight at once without sharing a cursor.
    CNT_CURSOR_LINES = 64,
    CNT_CURSOR = CNT_SHARDS * CNT_WORDS,
    CNT_TOTAL_WORDS = (CNT_SHARDS + CNT_CURSOR_LINES) * CNT_WORDS
};

__device__ __forceinline__ unsigned long long *counter_shard(unsigned long long *counters) {
    return counters + (size_t)(blockIdx.x % CNT_SHARDS) * CNT_WORDS;
}

struct ptmi_ctx {
    int device = 0;
    std::string err;

    // BVH: host copy in the reference's linear form + device copy in packed form
    bool have_bvh = false;
    ptmi::LinearBVH lin;
    ptmi::BvhStats stats;
    double build_ms = 0.0;
    uint64_t packed_inner = 0, packed_tris = 0;
    int max_depth = 0;
    // Entries the traversal stack needs.  max_depth counts the root as 1 (GetStats(1), main.go:809), so a leaf at the
    // deepest level has max_depth - 1 inner ancestors; a walk defers at most one child per inner node on its current
    // path, and an entry pushed at a deeper level is popped before one from a shallower level, so the entries in use
    // always belong to distinct levels: max_depth - 1 at most.  (The kernels come in stack sizes 16 / 20 / 24 / 28 / 32 /
    // 48 / 64; the LDS stack is what bounds the resident waves: 32 per CU with 20 entries, 26 with 24, 22 with 28, 20 with 32.)
    int stack_entries() const { return max_depth > 1 ? max_depth - 1 : 1; }
    float4 *d_nodes = nullptr;
    float4 *d_tris = nullptr;
    uint32_t root_ref = ptmi::D_REF_EMPTY_LEAF;
    uint32_t bounds_finite = 0;
    uint32_t dummy_tri = 0;
    std::vector<float> verts;          // kept for the scene upload (emissive sampling, normals)
    std::vector<uint32_t> tri_idx;

    // staging for the host-pointer entry points (grow-only)
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // second stream of the chunked host-pointer batches (created on first use)

    // scene + wavefront integrator state (ptmi_render.hip)
    struct RenderState *render = nullptr;

    // extend kernel selection and tuning (1 = persistent vote-scheduled kernel, 0 = one-shot if-if kernel)
    int extend_variant = 1;
    int persist_waves_per_cu = 32;      // upper bound; the launch takes min(this, what the LDS stacks allow)
    int num_cus = 256;
    int pool_paths = 1 << 27;           // most paths in flight in the integrator's pool ...
    uint64_t pool_bytes_max = 160ull << 30;  // ... and most device memory the pool may take (of 288 GB; ~1 KB per path)
    int implicit_fresh = 1;             // fresh samples without queue records where extend never writes into one (straight single-segment rays)
    int render_shrink_grids = 1;        // size the draining iterations' grids by the paths the host last saw alive
    int chunk_msamples = 1024;          // samples per pass over the region, in 2^20 (bounds the per-sample output buffer: 12 B each)
    int extend_vote = 1;                // extend (straight rays, one segment per call) as a vote-scheduled persistent kernel (0: one-shot)
    long long render_vote_min = -1;     // fewest rays a vote-scheduled launch of the integrator takes (-1: four times three chunks per resident wave, about 5 M)
    int render_vote_chunk = 512;        // queue entries reserved per cursor atomic in the integrator's vote-scheduled kernels
    int render_vote_refill_min = 32;    // ... and idle lanes that trigger a refill there (the batch kernels: vote_refill_min)
    int segment_shortcut = 1;           // the curved march skips the walk for segments far outside the tree's tight box
    int connect_vote = 1;               // connect as a vote-scheduled persistent kernel (0: one-shot waves always)
    int shade_split = 1;                // shade by class with the lean kernels (0: everything through the general kernel)
    int render_lanes = -1;              // independent wavefront loops side by side on two streams (1 .. 4; -1: 2 or 1 by tree size, render_core)
    int render_batch_iters = 2;         // wavefront iterations enqueued between two looks at the queue counters (render_core)
    int vote_refill_min = 16;           // refill once this many lanes are idle
    int vote_tri_bias = -8;             // inner step wins the vote when nI + bias >= nT
    int vote_chunk = 128;               // rays reserved per cursor atomic (multiple of 64)

    // per-kernel timing of the render calls (ptmi_set_profiling): totals in ms, by PROF_* class, and event bookkeeping
    bool profiling = false;
    double prof_ms[8] = {};
    uint64_t prof_iterations = 0, prof_launches = 0;
    struct ProfSpan { int cls; hipEvent_t a, b; };
    std::vector<ProfSpan> prof_open;            // spans of the pass in flight
    std::vector<hipEvent_t> prof_free;          // events ready for reuse

    // work counters
    bool counting = false;
    unsigned long long *d_counters = nullptr;
    uint32_t cursor_seq = 0;            // next line of the cursor ring (calls on one ctx are serialised by the caller)
    // One event per line of the ring, recorded behind the launch that uses the line: the 65th launch in flight waits for
    // the line's previous user instead of sharing (and re-zeroing) a live kernel's cursor (ptmi_api.hip: ctx_scratch).
    hipEvent_t cursor_ev[CNT_CURSOR_LINES] = {};
    bool cursor_pending[CNT_CURSOR_LINES] = {};
    // resident workgroups per CU of the persistent kernels, per kernel function (per context: contexts may be driven
    // from different OS threads, and different devices may answer differently)
    std::unordered_map<const void *, int> occupancy;
    int waves_per_cu(const void *fn, int block_threads) {
        auto it = occupancy.find(fn);
        if (it != occupancy.end()) return it->second;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block_threads, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 8; }
        occupancy.emplace(fn, per_cu);
        return per_cu;
    }

    float tight_min[3] = {0, 0, 0}, tight_max[3] = {0, 0, 0}, tight_margin = 0.0f;   // see DevBVH
    uint32_t tight_valid = 0;
    // RCCL communicator of this context (ptmi_comm.hip; an ncclComm_t), its rank and the number of ranks
    void *comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;

    // the allocation-free entry point's scratch on this context's device: the address of ptmi_api.hip's file-scope
    // __device__ array, resolved once in ptmi_create with this device current (every GPU has its own copy)
    unsigned long long *noalloc_scratch = nullptr;

    ptmi::DevBVH dev_bvh() const {
        return ptmi::DevBVH{d_nodes, d_tris, root_ref, bounds_finite, dummy_tri,
                            {tight_min[0], tight_min[1], tight_min[2]}, {tight_max[0], tight_max[1], tight_max[2]}, tight_margin, tight_valid};
    }
};

namespace ptmi {
// sums the shards of the device counter block into h[CNT_WORDS] (CNT_ERROR is or-ed); synchronises stream s
int read_counters(ptmi_ctx *ctx, unsigned long long *h, hipStream_t s);
int set_err(ptmi_ctx *ctx, int code, const std::string &msg);
int hip_err(ptmi_ctx *ctx, hipError_t e, const char *what);
int ensure_stage(ptmi_ctx *ctx, size_t bytes);
}  // namespace ptmi

#define PTMI_HIP(ctx, call)                                              \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) return ptmi::hip_err((ctx), e__, #call);  \
    } while (0)
