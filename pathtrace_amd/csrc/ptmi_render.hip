// ptmi_render.hip -- wavefront (queue-based) integrator: generate / extend / shade / connect / accumulate.
//
// Replaces the reference's recursive TraceRay (tracer.go:15-264) and its callees.  The recursion
// branches (a glass hit first traces the refraction subtree, then shades the same hit diffuse or
// reflective; tracer.go:49-91, :115-215) and shares one mutable RefractiveIndexTracker across sibling
// subtrees, so each primary sample walks its own recursion tree in the reference's depth-first order
// with an explicit frame stack in HBM; what is batched across samples are the expensive steps:
//   generate : camera rays (main.go:861-874) into free path slots
//   extend   : closest-hit march per path (tracer.go:34-35, 226-256; BVH walk = trav_device.hpp)
//   shade    : the per-hit logic of TraceRay / HandleDiffuseMaterial / HandleReflectiveMaterial, run
//              until the path needs another extend, a batch of shadow rays, or is finished
//   connect  : any-hit for the shadow-ray queue (QuickCheckIntersection, linear_bvh.go:202)
//   accumulate : per-pixel sums in sample order (Pixel.AddSample, renderer.go:59-61)
// Queues are compacted with one atomic per workgroup (ballot + popcount per wave, summed in LDS); finished paths
// are re-filled with fresh samples so the pool stays full.  A path's live state travels with its queue entry
// (coalesced); what stays behind per path are 128-byte frame lines, one per parked TraceRay invocation (queue order
// is arbitrary, so per-path records move whole lines where a [field][slot] layout cost one sector per word;
// DESIGN.md section 3).  The loop itself is driven from the device (IterArgs below).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ptmi_ctx.hpp"
#include "shade_device.hpp"

using namespace ptmi;

// Experiment build (tools/build_variant.sh regs -DPTMI_FRAME_REGS=1): shade keeps the top frame in 31 registers instead
// of LDS.  Not the product (the kernel cannot afford the registers); kept to reproduce round 1's "a frame word changes
// between a reload and its use" report, see DESIGN.md section 4.
#ifndef PTMI_FRAME_REGS
#define PTMI_FRAME_REGS 0
#endif

namespace {

constexpr int WAVE = 64;
// workgroup barrier for data exchanged through LDS only: waits for this wave's LDS operations, not for its global
// stores (hipcc's __syncthreads() also drains vmcnt, i.e. every frame / queue store the wave has in flight)
__device__ __forceinline__ void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- frame fields (one frame per TraceRay invocation on the explicit recursion stack) ----------
// A frame is 32 words = one 128-byte line, stored per path ([slot][depth][32]): shade builds the top frame in
// LDS ([word][thread]) from the queue entry of a pending call, or loads it there with eight 16-byte loads when a
// child returns to it, works on it there and writes it back only when the path parks under a child.
enum {
    // first 64-byte sector: what a frame that only waits for its one child (or its shadow rays) is finished with
    F_BB = 0,                                                              // bounces | bounceIndex << 16 | flags << 24 (inputs of TraceRay, :15)
    F_SP = 1,                                                              // sample counter of the GI / reflection loop | phase << 24
    F_MAT = 2,
    F_DG = 3,                                                              // doppler * gravity
    F_ALBEDO = 4,
    F_PDIRECT = 7,                                                         // the CALLER's direct lighting, summed from the shadow terms this call carried
    F_DIFF = 10,                                                           // diffuse component; until HandleDiffuseMaterial's final sum: its direct lighting
    F_REFR = 13,
    // second sector: the running sum of the GI / reflection rays and what re-derives the hit for the next one
    F_SUM = 16, F_ENERGY = 19, F_LN = 20,
    F_RO = 23, F_RD = 26, F_T = 29, F_PRIM = 30,                           // the hit being shaded
    // F_ALBEDO / F_MAT: what HandleDiffuseMaterial's final sum needs of the hit (textured albedo, material), kept in the
    // frame so that a frame popped by a returning child does not gather its triangle record again.
    // A frame is one 128-byte line in memory (32 words, the last one unused) and FRW words of LDS while it is on top.
    // A SHORT frame (PH_SHORT in its phase byte) has only its first sector in memory: shade_diff_kernel parks its
    // frames that way when scatterRays == 1 -- the sum is zero and no second ray will ever need the hit again.
    NF = 31, FRW = 31
};
// FL_ANC: a frame below this call waits for its refraction subtree (PH_REFR_WAIT); inherited by every child
// FL_SHORT (records only): the record has no second sector -- lastSurfaceNormal is {} and no terms arrive (a fresh sample)
enum { FL_SPECULAR = 1, FL_FLIPPED = 2, FL_ANC = 4, FL_SHORT = 8 };
// what the top frame is waiting for
enum { PH_EXTEND = 0, PH_REFR_WAIT = 1, PH_SHADOW_WAIT = 2, PH_GI_WAIT = 3, PH_REFL_WAIT = 4, PH_MASK = 0x7F, PH_SHORT = 0x80 };
enum { HS_MISS = 0, HS_HIT = 1, HS_HORIZON = 2 };
// queue counters: each on a 128-byte line of its own (atomics on one line serialise at ~12 ns each whatever word
// they address), the double-buffered queues with one counter per buffer
// Shadow rays are queued by kind (bucket 0 = sky sample, 1.. = light i, then the emissive-triangle sample; kinds
// beyond NSB-1 share the last bucket): a wave of the connect kernel then holds rays of one kind, which have similar
// directions and lengths, instead of a mix in which the short ones idle until the longest has finished.
enum { NSB = 8 };
// ---- shading by class ------------------------------------------------------------------------------------------
// A shade wave that holds a random mix of misses, glass hits, diffuse hits and returning frames executes the union of
// their code (round 2: lane utilisation 0.62, 2 100 wave instructions per wave-visit).  The extend queue is therefore
// sorted by CLASS before it is shaded: extend, which knows the hit, writes one class byte per entry (classify());
// bin_kernel turns the bytes into one index list per class; and three kernels shade the lists, each thread gathering
// the record of the entry its list names:
//   shade_ret_kernel   a call that returns at once (miss, horizon, failed entry test) and unwinds to the primary
//                      sample -- and the frames parked for their shadow rays, which unwind the same way.  No LDS frame,
//                      no staging: registers and the parked frames.
//   shade_diff_kernel  a fresh hit on a plain diffuse surface: HandleDiffuseMaterial's direct lighting + the GI child
//                      (tracer.go:443-648), straight-line code that ends by parking the frame.
//   shade_kernel       everything else (glass, reflective, emissive hit on an indirect ray, accretion disk, any path
//                      under a refraction frame, scatterRays > 1): the general state machine, its waves sorted by
//                      sub-class.
// Class codes (one byte per queue entry, written by extend):
enum { C_RET = 0, C_RET_ANC = 1, C_DIFF = 2, C_DIFF_LEAF = 3, C_GLASS = 4, C_REFL = 5, C_MID = 6, C_EMIT = 7, C_OTHER = 8, C_CODES = 9 };
// Index lists (bin_kernel maps a code to a list with a table the host fills: render_core):
enum { L_RET = 0, L_DIFF = 1, L_GEN = 2,                                 // the general kernel's lists, each padded to whole waves:
       L_G_RET = L_GEN, L_G_GLASS, L_G_REFL, L_G_MID, L_G_EMIT, L_G_DIFF, L_G_OTHER, NLIST, N_GEN = NLIST - L_GEN };
// what extend knows about a triangle's material (one byte per triangle, ptmi_upload_scene): tracer.go:49 / :95 / :116-118
enum { TC_GLASS = 1, TC_ACCR = 2, TC_REFL = 4, TC_MID = 8, TC_EMIT = 16 };
// what extend knows about the pending call (bits 28.. of the record's slot word: the pool has at most 2^28 slots)
enum { CTX_SHIFT = 28, CTX_FAIL = 1, CTX_ANC = 2, CTX_B0 = 4, CTX_BI = 8, SLOT_MASK = (1 << CTX_SHIFT) - 1 };
enum { QC_EXT0 = 0, QC_EXT1 = 1, QC_SHP0 = 2, QC_SHP1 = 3, QC_SHA0 = 4, QC_SHA1 = 5, QC_FREE = 6, QC_SRAY = 7, QC_NEXT = QC_SRAY + NSB,   // QC_NEXT: next sample id to generate
       QC_CURSOR = QC_NEXT + 1, QC_CURSOR_EXT = QC_CURSOR + 1,                  // ray cursors of the vote-scheduled kernels
       QC_LIST = QC_CURSOR_EXT + 1,                                             // lengths of the index lists (bin_kernel)
       QC_COUNT = QC_LIST + NLIST, QC_STRIDE = 16 };
struct ShadowCounts { uint32_t n[NSB]; };
// The wavefront loop is driven from the device: at the start of an iteration a one-thread kernel (iter_prep_kernel)
// turns the queue counters the last shade pass left behind into this iteration's arguments -- one 128-byte line that
// every kernel of the iteration reads its work size from -- and clears the counters for the next pass.  The host
// enqueues iterations without knowing their sizes (every kernel runs on a fixed grid and loops over its share) and only
// looks at the counters once per batch of iterations, one batch behind the queue, to see whether the pool has drained.
struct IterArgs {
    uint32_t n_ext;                // entries of the extend queue to trace and shade (this iteration's fresh samples included)
    uint32_t gen_n, gen_base;      // fresh samples to generate: entries gen_base .. gen_base + gen_n - 1, slots from the free list
    uint32_t n_shp, n_sha;         // parked frames whose shadow rays are answered in this iteration: lean (shade_ret_kernel) / general
    ShadowCounts sray;             // shadow rays per bucket
    uint32_t pad[2];
    unsigned long long first_sample;
};
static_assert(sizeof(IterArgs) <= 128, "IterArgs is one line");
// Path state that travels with the queues (no per-path header in memory).
//   Extend queue entry = the pending TraceRay call of a path: ONE record of whole 64-byte sectors (REC words; 128 bytes
//   for straight rays with up to three shadow-ray kinds), because the shade kernels gather records in class order and a
//   gathered record must not share its sectors with another entry (a [field][entry] layout read through a permutation
//   fetches every sector about twice).  Words:
//      0-2  ray origin             3  slot | CTX_* << 28          4-6  ray direction       7  sample id
//      8-9  spare (extend's answer -- status << 30 | primitive, t -- goes to a dense array beside the queue, q_hit: an
//           8-byte store into a record is a read-modify-write of its sector, extend +12 % when measured)
//     10    RNG key     11  RNG counter
//     12    depth | tracker length << 8 | bounces << 16 (int8) | bounceIndex << 24
//     13    flags | shadow-term mask << 8       14  energy       15  tracker entry 0      16-18  lastSurfaceNormal
//     19..  shadow terms of the caller's rays, 4 words each (term.xyz, position of the ray in the shadow-ray queue):
//           the first REC_INLINE (3; curved rays 2) in words 19.., the others from word 32 on
//     28-29 (curved rays) V_t at the end / at the start of the geodesic -- written by extend
//   Shade writes a wave's records as whole lines (they are contiguous in the next queue), extend reads the first 32 bytes.
//   shadow-wait queue entry = a parked frame waiting for connect's answers, 2 x float4 + nshadow terms:
//     (bits(slot), sample id, RNG key, RNG counter) / (depth | tracker length << 8, tracker entry 0, term mask, flags)
//   Lean entries (no refraction frame below them, scatterRays == 1) fill the queue from the front and are unwound by
//   shade_ret_kernel, the others fill it from the back (entry cap-1-k) and go to the general kernel.
// A pending call has no state beyond its arguments, so a fresh sample and a pushed child touch no per-path memory at
// all, and frame memory is only touched when a frame is parked under a child or waits for its shadow rays.
// Tracker entries beyond the first live in Pool::trk and are only touched by glass hits.
enum { R_O = 0, R_SLOT = 3, R_D = 4, R_SID = 7, R_HIT = 8, R_T = 9, R_KEY = 10, R_CTR = 11, R_PACK = 12, R_FLAGS = 13, R_ENERGY = 14,
       R_TRK0 = 15, R_LN = 16, R_TERMS = 19, R_VT = 28, REC_BASE = 32 };

// self-check failures reported through counters[CNT_ERROR] (a corrupted path is dropped, never dereferenced)
enum { ERR_SLOT = 1, ERR_DEPTH = 2, ERR_PRIM = 4, ERR_MAT = 8, ERR_PHASE = 16, ERR_QUEUE = 32, ERR_CLASS = 64 };

struct DevScene {
    const float4 *stri;                  // shading record per triangle, 5 x float4: A B C, N0 N1 N2, material id
    const uint8_t *tri_class;            // TC_* bits per triangle (classify)
    uint32_t ntris, nmats;
    const float *uvs; const TexDesc *tex; const uint32_t *tex_pixels; uint32_t ntex;   // vnmu.UVs + CachedImage table
    const PtMaterial *mats; const uint32_t *emissive; uint32_t nemissive;
    const PtLight *lights; uint32_t nlights;
    PtSky sky; const uint16_t *sky_img;
    PtBlackHole bh; int has_bh;
};

struct RenderArgs {
    PtCamera cam;
    PtParams p;
    int x0, y0, w, h;
    // row m of the region is image row y0 + (m / row_block) * row_period + m % row_block: a plain rectangle has
    // row_block = row_period = h, a row-interleaved shard (ptmi_render_rows_dev) takes every nshards-th block of rows
    int row_block, row_period;
    // a list of tiles (ptmi_render_tiles_dev) instead: pixel p of the region belongs to tile k with
    // tile_first[k] <= p < tile_first[k+1], row-major inside tile_rect[k] = (x0, y0, w, h)
    const uint32_t *tile_first; const int4 *tile_rect; uint32_t ntiles;
    uint32_t spp_begin, spp_chunk;       // this batch covers samples [spp_begin, spp_begin+spp_chunk) of every pixel
    uint64_t total;                      // w*h*spp_chunk
    // The samples of a pass are dealt to `nlanes` independent wavefront loops in blocks of LANE_BLOCK sample ids, round
    // robin: sample j of lane `lane` is sample id ((j / LANE_BLOCK) * nlanes + lane) * LANE_BLOCK + j % LANE_BLOCK.
    uint32_t lane, nlanes;
    // fresh samples are not written into the extend queue as records: generate writes their rays into a dense array for
    // extend (32 bytes each; making the ray up inside the persistent walk's refill step stalls the wave: extend +13 %)
    // and shade makes up the record of entry gen_base + j on the spot (fresh_head).  Set for straight single-segment
    // rays (extend never moves such a ray, so it has nothing to write back into the record).
    uint32_t implicit_fresh;
    // several frames in one call (ptmi_render_frames_dev): the region is `frame_h` rows per frame, one frame under the
    // other, and frame f is seen through cams[f] (device memory); nullptr: one frame, `cam`
    const PtCamera *cams; int frame_h;
};
enum { LANE_BLOCK_LOG2 = 16, LANE_BLOCK = 1 << LANE_BLOCK_LOG2 };
__host__ __device__ inline uint64_t lane_sample_id(uint64_t j, uint32_t lane, uint32_t nlanes) {
    return nlanes <= 1 ? j : ((((j >> LANE_BLOCK_LOG2) * nlanes + lane) << LANE_BLOCK_LOG2) | (j & (LANE_BLOCK - 1)));
}
// how many of the sample ids 0 .. total-1 belong to `lane` (only the last block can be short, and it is its lane's last)
static uint64_t lane_sample_count(uint64_t total, uint32_t lane, uint32_t nlanes) {
    if (nlanes <= 1) return total;
    const uint64_t full = total >> LANE_BLOCK_LOG2, rest = total & (LANE_BLOCK - 1);
    uint64_t n = (full / nlanes + (lane < full % nlanes ? 1 : 0)) << LANE_BLOCK_LOG2;
    if (rest && full % nlanes == lane) n += rest;
    return n;
}

// Path pool: array of structures per slot.
struct Pool {
    uint32_t cap;
    int nframes, ntrk, nshadow;          // bounces+1, tracker history capacity (multiple of 4), nlights+2
    int rec_q;                           // float4s per extend-queue record (REC / 4; a multiple of 4: whole 64-byte sectors)
    int rec_inline;                      // shadow terms held in words 19.. of a record (the others from word 32 on)
    float4 *fr;                          // [cap][nframes][8]   frames (F_*).  (Tried: [slot][sector][depth], the first sectors of a path's frames side
                                         // by side for the kernel that unwinds them: shade +4 ms per C3 frame -- two half-line writes at
                                         // different times into one 128-byte line cost more than the neighbouring reads save.)
    float *trk;                          // [cap][ntrk]         RefractiveIndexTracker.indices 1 .. (entry 0 travels with the path)
    // queues
    float4 *q_rec[2];                    // extend queue (double buffered): one record per entry (R_*)
    float2 *q_hit;                       // extend's answer per entry of the queue being traced: (status << 30 | primitive, t)
    uint8_t *q_cls;                      // class code of every entry of the queue being traced (extend)
    uint32_t *q_idx;                     // [NLIST][cap] entries of the queue being shaded, by list (bin_kernel)
    float4 *q_shpath[2];                 // paths waiting for their shadow rays, 2 x float4 per entry: lean ones from the front, the others from the back
    float4 *q_shp_sh[2];                 // ... their shadow terms: nshadow x (term.xyz, bits(position of the ray))
    float4 *q_sray_a, *q_sray_b;         // shadow rays: (o.xyz, range) / (d.xyz, -)
    uint8_t *q_sray_occ;                 // connect's answers, by position in the shadow-ray queue (0 free, 1 occluded)
    float4 *q_fresh_a, *q_fresh_b;       // rays of this iteration's fresh samples, dense ((o, slot | ctx), (d, sample id)): what extend reads of them
    uint32_t *q_free[2];                 // slots that finished in the last shade pass (double buffered like the queues: this
                                         // iteration's fresh samples take theirs from [cur], shade fills [cur ^ 1])
    unsigned long long *qcnt;            // QC_COUNT counters, QC_STRIDE words apart
    IterArgs *it;                        // this iteration's arguments (iter_prep_kernel)
    float *out;                          // per-sample radiance [total][3]
};
__device__ __forceinline__ float4 *rec_at(const Pool &pl, int buf, size_t entry) { return pl.q_rec[buf] + entry * (size_t)pl.rec_q; }
// word offset of shadow term k inside a record
__device__ __forceinline__ int term_word(const Pool &pl, int k) { return k < pl.rec_inline ? R_TERMS + 4 * k : REC_BASE + 4 * (k - pl.rec_inline); }

__device__ __forceinline__ uint32_t wave_append(unsigned long long *counter, bool pred) {
    // one atomic per wave: ballot, popcount, prefix by lane
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    int lane = threadIdx.x & 63;
    uint32_t total = (uint32_t)__popcll(mask);
    unsigned long long base = 0;
    int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)total);
    base = __shfl((long long)base, leader);
    uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    return (uint32_t)base + prefix;
}

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ float i2f(int32_t i) { return __int_as_float(i); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ int32_t f2i(float f) { return __float_as_int(f); }

// image pixel of the region's pixel number `pix`
__device__ __forceinline__ void region_pixel(const RenderArgs &ra, uint32_t pix, int &X, int &Y) {
    if (ra.ntiles) {
        uint32_t lo = 0, hi = ra.ntiles - 1;                   // last tile whose first pixel is <= pix
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (ra.tile_first[mid] <= pix) lo = mid; else hi = mid - 1;
        }
        const int4 r = ra.tile_rect[lo];
        const uint32_t q = pix - ra.tile_first[lo];
        X = r.x + (int)(q % (uint32_t)r.z); Y = r.y + (int)(q / (uint32_t)r.z);
        return;
    }
    const int m = (int)(pix / (uint32_t)ra.w);
    X = ra.x0 + (int)(pix % (uint32_t)ra.w);
    Y = ra.y0 + (m / ra.row_block) * ra.row_period + m % ra.row_block;
}

// ------------------------------------------------------------------------------------------ generate
// main.go:861-874 + the TraceRay call of :875 (bounceIndex 0, lastSurfaceNormal {}, not specular,
// NewRefractiveIndexTracker(1.0), energy 1.0)
// Start of a call: every slot is free, nothing is queued.
__global__ __launch_bounds__(256) void iter_init_kernel(Pool pl, uint32_t n0) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += stride) pl.q_free[0][i] = i;
    if (blockIdx.x == 0 && threadIdx.x < QC_COUNT) pl.qcnt[(size_t)threadIdx.x * QC_STRIDE] = threadIdx.x == QC_FREE ? n0 : 0ull;
}

// Start of an iteration on queue buffer `cur` (see IterArgs).  One thread: a dozen counters.  `total` = the number
// of samples of this loop (its lane's share of the pass).
__global__ void iter_prep_kernel(Pool pl, int cur, unsigned long long total, unsigned long long *counters) {
    if (blockIdx.x || threadIdx.x) return;
    auto qc = [&](int w) -> unsigned long long & { return pl.qcnt[(size_t)w * QC_STRIDE]; };
    const int nxt = cur ^ 1;
    unsigned long long n_ext = qc(QC_EXT0 + cur), n_shp = qc(QC_SHP0 + cur), n_sha = qc(QC_SHA0 + cur), n_free = qc(QC_FREE), next = qc(QC_NEXT);
    IterArgs a;
    bool bad = n_ext > pl.cap || n_shp + n_sha > pl.cap || n_free > pl.cap || next > total;
    for (int b = 0; b < NSB; ++b) {
        const unsigned long long n = qc(QC_SRAY + b);
        const unsigned long long room = b < NSB - 1 ? (unsigned long long)pl.cap : (unsigned long long)pl.cap * (unsigned long long)max(pl.nshadow - (NSB - 1), 0);
        bad = bad || n > room;
        a.sray.n[b] = (uint32_t)n;
    }
    if (bad) {          // a counter out of range: raise the error word and let the pool run dry
        atomicOr(&counters[CNT_ERROR], (unsigned long long)ERR_QUEUE);
        n_ext = n_shp = n_sha = n_free = 0; next = total;
        for (int b = 0; b < NSB; ++b) a.sray.n[b] = 0;
    }
    unsigned long long ng = min(n_free, total - next);
    if (n_ext + ng > pl.cap) ng = pl.cap - n_ext;
    a.gen_n = (uint32_t)ng; a.gen_base = (uint32_t)n_ext; a.first_sample = next;
    a.n_ext = (uint32_t)(n_ext + ng); a.n_shp = (uint32_t)n_shp; a.n_sha = (uint32_t)n_sha;
    if (n_ext + ng) atomicAdd(&counters[CNT_ENTRIES], n_ext + ng);
    a.pad[0] = a.pad[1] = 0;
    *pl.it = a;
    qc(QC_NEXT) = next + ng;
    qc(QC_EXT0 + cur) = n_ext + ng;
    qc(QC_EXT0 + nxt) = 0; qc(QC_SHP0 + nxt) = 0; qc(QC_SHA0 + nxt) = 0; qc(QC_FREE) = 0;
    for (int b = 0; b < NSB; ++b) qc(QC_SRAY + b) = 0;
    for (int l = 0; l < NLIST; ++l) qc(QC_LIST + l) = 0;
    qc(QC_CURSOR) = 0; qc(QC_CURSOR_EXT) = 0;
}

// ---- whole-line stores ------------------------------------------------------------------------------------------
// A lane that stores a 128-byte line of its own (a parked frame, a queue record) with eight 16-byte stores touches
// 16 bytes of 64 different lines per instruction.  Instead the wave exchanges the lines through LDS and lane 8 g + j
// writes quarter-line j of the lines of lanes 8 g .. 8 g + 7 in turn: every store instruction then covers eight whole
// lines.  `col` = this lane's column of an LDS area laid out [word][stride] that holds its line (word w at
// col[w * stride]); dst = where the line goes (nullptr: this lane has none); nwords <= 32, the rest of the line is 0.
__device__ __forceinline__ void wave_store_lines(const float *col, int stride, float4 *dst, int nwords) {
    const unsigned long long m_st = __ballot(dst != nullptr);
    if (!m_st) return;
    const int lane = threadIdx.x & 63, grp = lane & ~7, sub = lane & 7;
    const uint32_t lo = (uint32_t)(uintptr_t)dst, hi = (uint32_t)((uintptr_t)dst >> 32);
    const float *base = col - lane;                                  // column of lane 0 of this wave
    for (int i = 0; i < 8; ++i) {
        const int src = grp + i;
        const uint32_t slo = (uint32_t)__shfl((int)lo, src), shi = (uint32_t)__shfl((int)hi, src);
        if (m_st >> src & 1ull) {
            float4 *d = (float4 *)(((uintptr_t)shi << 32) | (uintptr_t)slo);
            const float *c = base + src;
            d[sub] = make_float4(4 * sub < nwords ? c[(4 * sub) * stride] : 0.0f, 4 * sub + 1 < nwords ? c[(4 * sub + 1) * stride] : 0.0f,
                                 4 * sub + 2 < nwords ? c[(4 * sub + 2) * stride] : 0.0f, 4 * sub + 3 < nwords ? c[(4 * sub + 3) * stride] : 0.0f);
        }
    }
}

enum { XCH_WORDS = 16 };
// Half a line per lane through a [16][256] exchange area: lane 8 g + 4 r + q writes quarter q of the half-lines of lanes
// 8 g + r, 8 g + 2 + r, ... (every store instruction covers sixteen 64-byte sectors).  The lane's 16 words of this half
// are in its column already.
__device__ __forceinline__ void wave_store_half(const float *col, float4 *dst, int half) {
    const unsigned long long m_st = __ballot(dst != nullptr);
    if (!m_st) return;
    const int lane = threadIdx.x & 63, grp = lane & ~7, r = (lane >> 2) & 1, q = lane & 3;
    const uint32_t lo = (uint32_t)(uintptr_t)dst, hi = (uint32_t)((uintptr_t)dst >> 32);
    const float *base = col - lane;
    for (int i = 0; i < 4; ++i) {
        const int src = grp + 2 * i + r;
        const uint32_t slo = (uint32_t)__shfl((int)lo, src), shi = (uint32_t)__shfl((int)hi, src);
        if (m_st >> src & 1ull) {
            float4 *d = (float4 *)(((uintptr_t)shi << 32) | (uintptr_t)slo);
            const float *c = base + src;
            d[4 * half + q] = make_float4(c[(4 * q) * 256], c[(4 * q + 1) * 256], c[(4 * q + 2) * 256], c[(4 * q + 3) * 256]);
        }
    }
}

// What extend tells the binning about an entry (C_*): tracer.go's branches in the order TraceRay takes them -- the
// entry test (:16), miss / horizon (:250-263), glass (:49), the accretion disk (:95), reflectivity (:116-163), an
// emitter seen by an indirect ray (:461-465), and the plain diffuse surface with or without bounces left (:622).
__device__ __forceinline__ uint32_t classify(int status, uint32_t tc, uint32_t ctx) {      // tc = TC_* bits of the hit triangle
    if (status != HS_HIT || (ctx & CTX_FAIL)) return (ctx & CTX_ANC) ? C_RET_ANC : C_RET;
    if (tc & TC_GLASS) return C_GLASS;
    if (tc & TC_ACCR) return C_OTHER;
    if (tc & TC_REFL) return C_REFL;
    if (tc & TC_MID) return C_MID;
    if ((tc & TC_EMIT) && (ctx & CTX_BI)) return C_EMIT;
    return (ctx & CTX_B0) ? C_DIFF_LEAF : C_DIFF;
}
// ptmi_upload_scene writes a triangle's TC_* bits into the spare fourth word of its packed traversal record, so the
// vote-scheduled walk has them at hand with the hit (a gather by primitive at store time stalls the whole wave's
// refill step: extend +12 % when measured)
__global__ __launch_bounds__(256) void patch_tri_class_kernel(float4 *tris, uint32_t nrecords, const uint8_t *__restrict__ tri_class, uint32_t ntris) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrecords) return;
    float4 v = tris[3 * (size_t)r + 2];
    const uint32_t prim = f2u(v.y);
    v.w = u2f(prim < ntris ? (uint32_t)tri_class[prim] : 0u);
    tris[3 * (size_t)r + 2] = v;
}
// CTX_* bits of a pending call with these arguments
__device__ __forceinline__ uint32_t call_ctx(int bounces, int bindex, float energy, uint32_t flags) {
    return ((bounces < 0 || energy < (float)1e-2) ? (uint32_t)CTX_FAIL : 0u) | ((flags & FL_ANC) ? (uint32_t)CTX_ANC : 0u) |
           (bounces <= 0 ? (uint32_t)CTX_B0 : 0u) | (bindex > 0 ? (uint32_t)CTX_BI : 0u);
}
__device__ __forceinline__ uint32_t pack_call(int depth, uint32_t trk_len, int bounces, int bindex) {
    return ((uint32_t)depth & 0xFFu) | (trk_len & 0xFFu) << 8 | ((uint32_t)bounces & 0xFFu) << 16 | ((uint32_t)bindex & 0xFFu) << 24;
}

// A fresh sample is a pending call and nothing else.  The first sector of its record (the whole record: FL_SHORT) --
// main.go:862-875: the camera ray, bounceIndex 0, lastSurfaceNormal {}, not specular, NewRefractiveIndexTracker(1.0),
// energy 1.0; TraceRay's entry test (tracer.go:16) can only fail here for bounces < 0 (CTX_FAIL).  Fresh sample j of the
// iteration is entry gen_base + j of the extend queue and takes slot q_free[cur][j].
struct RecHead { float4 r0, r1, r2, r3; };
__device__ __forceinline__ RecHead fresh_head(const Pool &pl, const RenderArgs &ra, int cur, uint32_t j) {
    const uint32_t slot = pl.q_free[cur][j];
    const uint32_t sid = (uint32_t)lane_sample_id(pl.it->first_sample + j, ra.lane, ra.nlanes);   // total < 2^32/3 (render_core)
    const uint32_t pix = sid / ra.spp_chunk, smp = sid % ra.spp_chunk;
    int X, Y;
    region_pixel(ra, pix, X, Y);
    V3 pos = mk(ra.cam.position), fwd = mk(ra.cam.forward), right = mk(ra.cam.right), up = mk(ra.cam.up);
    float frustum = ra.cam.frustum_distance;
    if (ra.cams) {                                  // the frame this row belongs to, and its camera
        const int f = Y / ra.frame_h;
        Y -= f * ra.frame_h;
        const PtCamera *c = ra.cams + f;
        pos = mk(c->position); fwd = mk(c->forward); right = mk(c->right); up = mk(c->up); frustum = c->frustum_distance;
    }
    Rng rng = rng_init(ra.p.seed, (uint32_t)(Y * ra.p.width + X), ra.spp_begin + smp);
    const float rx = ((float)X + rng_f32(rng)) / (float)ra.p.width;
    const float ry = ((float)Y + rng_f32(rng)) / (float)ra.p.height;
    const float px = (rx - 0.5f) * 2.0f, py = (ry - 0.5f) * 2.0f;
    const V3 rayOrigin = add(add(add(pos, scale(fwd, frustum)), scale(up, py)), scale(right, px));
    const V3 d = normalize(mk(rayOrigin.x - pos.x, rayOrigin.y - pos.y, rayOrigin.z - pos.z));
    RecHead h;
    h.r0 = make_float4(pos.x, pos.y, pos.z, u2f(slot | call_ctx(ra.p.bounces, 0, 1.0f, 0u) << CTX_SHIFT));
    h.r1 = make_float4(d.x, d.y, d.z, u2f(sid));
    h.r2 = make_float4(0.0f, 0.0f, u2f(rng.key), u2f(rng.ctr));                                  // R_HIT, R_T spare; R_KEY, R_CTR
    h.r3 = make_float4(u2f(pack_call(0, 1u, ra.p.bounces, 0)), u2f((uint32_t)FL_SHORT), 1.0f, 1.0f);   // depth 0, tracker {1.0}; no terms; energy; trk0
    return h;
}
// The first sector of entry e of the extend queue: from its record, or made up for a fresh sample that has none.
__device__ __forceinline__ RecHead call_head(const Pool &pl, const RenderArgs &ra, int cur, uint32_t e) {
    const uint32_t gen_base = pl.it->gen_base;
    if (ra.implicit_fresh && e >= gen_base) return fresh_head(pl, ra, cur, e - gen_base);
    const float4 *rec = rec_at(pl, cur, e);
    RecHead h;
    h.r0 = rec[0]; h.r1 = rec[1]; h.r2 = rec[2]; h.r3 = rec[3];
    return h;
}
// ... and the ray alone (extend)
__device__ __forceinline__ void call_ray(const Pool &pl, const RenderArgs &ra, int cur, uint32_t e, float4 &qa, float4 &qb) {
    const uint32_t gen_base = pl.it->gen_base;
    if (ra.implicit_fresh && e >= gen_base) { qa = pl.q_fresh_a[e - gen_base]; qb = pl.q_fresh_b[e - gen_base]; return; }
    const float4 *rec = rec_at(pl, cur, e);
    qa = rec[0]; qb = rec[1];
}
// The fresh samples' rays for extend (implicit_fresh), or, where extend does write into records (a marched or curved
// ray), their records: one sector each at the end of the extend queue (no atomic: entry gen_base + i), written as
// whole sectors.
__global__ __launch_bounds__(256) void generate_rays_kernel(Pool pl, RenderArgs ra, int cur) {
    const uint32_t n = pl.it->gen_n;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const RecHead h = fresh_head(pl, ra, cur, i);
        pl.q_fresh_a[i] = h.r0; pl.q_fresh_b[i] = h.r1;
    }
}
__global__ __launch_bounds__(256) void generate_kernel(Pool pl, RenderArgs ra, int cur) {
    // the 256 records a workgroup writes per trip are contiguous, so they are put together in LDS ([word][thread]) and
    // written out as whole lines (per-thread 16-byte stores 128 bytes apart reach 3.2 TB/s)
    __shared__ float s_rec[16 * 256];
    const uint32_t n = pl.it->gen_n, q_base = pl.it->gen_base;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        float4 *dst = nullptr;
        if (i < n) {
            const RecHead h = fresh_head(pl, ra, cur, i);
            float *w = s_rec + threadIdx.x;
            const float4 q[4] = {h.r0, h.r1, h.r2, h.r3};
#pragma unroll
            for (int k = 0; k < 4; ++k) { w[(4 * k) * 256] = q[k].x; w[(4 * k + 1) * 256] = q[k].y; w[(4 * k + 2) * 256] = q[k].z; w[(4 * k + 3) * 256] = q[k].w; }
            dst = rec_at(pl, cur, (size_t)q_base + i);
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes, before its lanes read them
        wave_store_half(s_rec + threadIdx.x, dst, 0);
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Work counters of the integrator's walks (ptmi_set_counting: the COUNT = true instantiations): inner nodes popped,
// triangles tested, leaves visited -- what the render leg's roofline prices (64 / 48 bytes each).  The query counts
// (CNT_CLOSEST / CNT_ANY) are kept by every instantiation.
template <bool COUNT, bool ANY>
__device__ __forceinline__ void flush_walk_counters(const TravCounters &c, unsigned long long *counters) {
    if (!COUNT) return;
    uint32_t vi = c.inner, vt = c.tri, vl = c.leaf;
    for (int off = 32; off > 0; off >>= 1) { vi += __shfl_xor(vi, off); vt += __shfl_xor(vt, off); vl += __shfl_xor(vl, off); }
    if ((threadIdx.x & 63) == 0) {
        counters = counter_shard(counters);
        atomicAdd(&counters[CNT_INNER], (unsigned long long)vi);
        atomicAdd(&counters[CNT_TRI], (unsigned long long)vt);
        atomicAdd(&counters[CNT_LEAF], (unsigned long long)vl);
        if (ANY) { atomicAdd(&counters[CNT_INNER_ANY], (unsigned long long)vi); atomicAdd(&counters[CNT_TRI_ANY], (unsigned long long)vt); }
    }
}

// ------------------------------------------------------------------------------------------ extend
// The march loop of TraceRay (tracer.go:33-35, 226-256): closest hit on the segment [0, stepSize];
// on a miss advance (straight) or RK4-step the geodesic (curved) and test the horizon.
// Queue entries carry the ray and the answer goes into the queue's hit array, so the kernel touches no per-path memory;
// it writes the ray back into its entry only when shade needs the moved one (a hit after marching, a curved ray).
// Runs on a fixed grid and loops over the queue; leaves at once when the launch is large enough for the vote-scheduled
// kernel to take it (vote_from).
template <int STACK, bool CURVED, bool COUNT>
__global__ __launch_bounds__(WAVE) void render_extend_kernel(DevBVH bvh, Pool pl, DevScene sc, RenderArgs ra, int cur,
                                                             uint32_t vote_from, unsigned long long *counters) {
    __shared__ uint32_t smem[STACK * WAVE];
    const uint32_t n = pl.it->n_ext;
    if (n >= vote_from) return;
    uint32_t nq = 0;
    TravCounters tc;
    for (uint32_t base = blockIdx.x * WAVE; base < n; base += gridDim.x * WAVE) {
        const uint32_t i = base + threadIdx.x;
        if (i >= n) continue;
        float4 *rec = rec_at(pl, cur, i);
        float4 qa, qb;
        call_ray(pl, ra, cur, i, qa, qb);
        V3 o = mk(qa.x, qa.y, qa.z);
        V3 d = mk(qb.x, qb.y, qb.z);
        RayState st = {0, 0, 0, 0, 0, 0, 0, 0};
        float vt0 = 1.0f;
        float st_r_sq = 0.0f;                                                      // state_r_sq(st): the horizon test's sum, reused by the next step
        if (CURVED) { st = get_initial_state(o, d, sc.bh); vt0 = st.vt; st_r_sq = state_r_sq(st); }   // tracer.go:28-31
        int status = HS_MISS, prim = -1;
        float t = 0.0f;
        bool moved = false;
        bool d_stale = false;                                                      // curved: d has not been normalised from st yet
        for (int step = 0; step < ra.p.max_steps; ++step) {
            ++nq;
            // a geodesic spends most of its 3 000 segments far from any geometry: those queries are misses without a walk,
            // and for most of them the origin alone says so -- the direction is normalised (:241, an IEEE square root and
            // three divisions) only for the steps that come near the tree, and once at the end for shade
            bool skip = false;
            if (CURVED) {
                skip = point_far_from_tree(bvh, o.x, o.y, o.z, ra.p.step_size);
                if (!skip) {
                    if (d_stale) { d = normalize(mk(st.vx, st.vy, st.vz)); d_stale = false; }
                    skip = segment_outside_tree(bvh, o.x, o.y, o.z, d.x, d.y, d.z, ra.p.step_size);
                }
            }
            if (skip) { prim = -1; t = 0.0f; }
            else closest_hit<WAVE, COUNT>(bvh, smem + threadIdx.x, o.x, o.y, o.z, d.x, d.y, d.z, ra.p.step_size, prim, t, tc);
            if (prim >= 0) { status = HS_HIT; break; }
            moved = true;
            if (!CURVED) {
                o = add(o, scale(d, ra.p.step_size));                              // :226-228
            } else {
                st = rk4_step(st, st_r_sq, ra.p.step_size, sc.bh.rs);              // :231
                V3 rel = mk(st.px, st.py, st.pz);
                o = add(rel, mk(sc.bh.position));                                  // :238
                d_stale = true;                                                    // :241, when someone needs it
                st_r_sq = dot(rel, rel);                                           // = state_r_sq(st), operation for operation
                if (st_r_sq <= sc.bh.rs * sc.bh.rs) { status = HS_HORIZON; break; }   // :250-255
            }
        }
        pl.q_hit[i] = make_float2(u2f((uint32_t)status << 30 | ((uint32_t)prim & 0x3FFFFFFFu)), t);
        pl.q_cls[i] = (uint8_t)classify(status, status == HS_HIT ? (uint32_t)sc.tri_class[prim] : 0u, f2u(qa.w) >> CTX_SHIFT);
        // shade reads the ray from the record: the origin of a hit (tracer.go:37 adds t to the CURRENT origin) and,
        // for a curved ray, the direction it ended with (the sky lookup of a miss, :260-263)
        if (CURVED) {
            if (d_stale) d = normalize(mk(st.vx, st.vy, st.vz));
            rec[0] = make_float4(o.x, o.y, o.z, qa.w);
            rec[1] = make_float4(d.x, d.y, d.z, qb.w);
            ((float2 *)rec)[R_VT / 2] = make_float2(st.vt, vt0);
        } else if (moved && status == HS_HIT) {
            rec[0] = make_float4(o.x, o.y, o.z, qa.w);
        }
    }
    for (int off = 32; off > 0; off >>= 1) nq += __shfl_xor(nq, off);
    if ((threadIdx.x & 63) == 0 && nq) atomicAdd(&counter_shard(counters)[CNT_CLOSEST], (unsigned long long)nq);
    flush_walk_counters<COUNT, false>(tc, counters);
}

// extend for straight rays with one segment per call (max_steps == 1, no black hole): persistent waves walking
// vote-scheduled single steps (vote_walk, trav_device.hpp), queue entries dealt in chunks from one cursor.  Same output
// as render_extend_kernel: the answer into the record (a ray that is traced once never moves before a hit, and shade
// does not look at the origin of a miss, so the ray itself is never written back) and the entry's class byte.
// A persistent wave reserves three chunks of queue entries before it starts (one in use, two ahead): the chunk is
// `chunk_max` entries when the launch feeds every wave that much, less for smaller launches (never below one wave's 64).
__device__ __forceinline__ uint32_t vote_chunk(uint64_t n, uint32_t chunk_max) {
    const uint64_t fair = n / (3ull * gridDim.x) / WAVE * WAVE;
    return (uint32_t)min((uint64_t)chunk_max, max((uint64_t)WAVE, fair));
}
struct ExtTag { uint32_t entry, ctx; };
template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(STACK <= 20 ? 8 : 1, 8))) void render_extend_vote_kernel(DevBVH bvh, Pool pl, RenderArgs ra, int cur, uint32_t vote_from, unsigned long long *counters, int refill_min,
                                                                  int tri_bias, uint32_t chunk_max) {
    __shared__ uint32_t smem[STACK * WAVE];
    const uint32_t n = pl.it->n_ext;
    if (n < vote_from || (uint64_t)blockIdx.x * WAVE >= n) return;          // the one-shot kernel takes small launches
    const float step = ra.p.step_size;
    TravCounters cnt;
    vote_walk<WAVE, false, COUNT, ExtTag, true>(
        bvh, smem + threadIdx.x, n, pl.qcnt + (size_t)QC_CURSOR_EXT * QC_STRIDE, refill_min, tri_bias, vote_chunk(n, chunk_max), cnt,
        [&](uint64_t idx, NextRay &nx, ExtTag &tag) {
            float4 qa, qb;
            call_ray(pl, ra, cur, (uint32_t)idx, qa, qb);
            tag.entry = (uint32_t)idx; tag.ctx = f2u(qa.w) >> CTX_SHIFT;
            nx.ox = qa.x; nx.oy = qa.y; nx.oz = qa.z; nx.dx = qb.x; nx.dy = qb.y; nx.dz = qb.z; nx.tmax = step;
        },
        [&](const ExtTag &tag, const RayTrav &r) {
            const bool hit = r.best >= 0;
            pl.q_hit[tag.entry] =
                make_float2(u2f((uint32_t)(hit ? HS_HIT : HS_MISS) << 30 | ((uint32_t)r.best & 0x3FFFFFFFu)), hit ? r.best_t : 0.0f);
            pl.q_cls[tag.entry] = (uint8_t)classify(hit ? HS_HIT : HS_MISS, r.best_aux, tag.ctx);
        });
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counter_shard(counters)[CNT_CLOSEST], (unsigned long long)n);
    flush_walk_counters<COUNT, false>(cnt, counters);
}

// ------------------------------------------------------------------------------------------ bin
// Turns the class bytes extend wrote into one index list per class (the host's table maps a code to a list).  A
// workgroup takes BIN_CHUNK consecutive entries, wave w the w-th quarter of them, and appends its entries of a list in
// entry order: one atomic per workgroup and list (a few thousand per launch; they serialise at ~12 ns on a list's
// counter line).  Within a chunk the order of the queue is kept, so neighbouring rays stay neighbours.
enum { BIN_PER_LANE = 32, BIN_CHUNK = 256 * BIN_PER_LANE };
struct BinMap { uint8_t list[16]; };
__global__ __launch_bounds__(256) void bin_kernel(Pool pl, BinMap map) {
    const uint32_t n = pl.it->n_ext;
    const uint32_t chunk0 = blockIdx.x * (uint32_t)BIN_CHUNK;
    if (chunk0 >= n) return;
    __shared__ uint32_t s_cnt[NLIST][4], s_base[NLIST][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t first = chunk0 + (uint32_t)wv * (64u * BIN_PER_LANE);
    uint32_t code[BIN_PER_LANE];          // list of entry first + 64 k + lane (NLIST: none)
    uint32_t cnt[NLIST];
#pragma unroll
    for (int l = 0; l < NLIST; ++l) cnt[l] = 0;
#pragma unroll
    for (int k = 0; k < BIN_PER_LANE; ++k) {
        const uint32_t e = first + 64u * k + lane;
        uint32_t c = e < n ? pl.q_cls[e] : 0xFFu;
        code[k] = c < (uint32_t)C_CODES ? map.list[c] : (uint32_t)NLIST;
    }
#pragma unroll
    for (int k = 0; k < BIN_PER_LANE; ++k) {
#pragma unroll
        for (int l = 0; l < NLIST; ++l) cnt[l] += (uint32_t)__popcll(__ballot(code[k] == (uint32_t)l));     // wave-uniform
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < NLIST; ++l) s_cnt[l][wv] = cnt[l];
    }
    __syncthreads();
    if (threadIdx.x < NLIST) {
        const int l = threadIdx.x;
        const uint32_t c0 = s_cnt[l][0], c1 = s_cnt[l][1], c2 = s_cnt[l][2], c3 = s_cnt[l][3], total = c0 + c1 + c2 + c3;
        uint32_t base = 0;
        if (total) base = (uint32_t)atomicAdd(&pl.qcnt[(size_t)(QC_LIST + l) * QC_STRIDE], (unsigned long long)total);
        s_base[l][0] = base; s_base[l][1] = base + c0; s_base[l][2] = base + c0 + c1; s_base[l][3] = base + c0 + c1 + c2;
    }
    __syncthreads();
    uint32_t pos[NLIST];
#pragma unroll
    for (int l = 0; l < NLIST; ++l) pos[l] = s_base[l][wv];
#pragma unroll
    for (int k = 0; k < BIN_PER_LANE; ++k) {
        const uint32_t e = first + 64u * k + lane;
#pragma unroll
        for (int l = 0; l < NLIST; ++l) {
            const unsigned long long m = __ballot(code[k] == (uint32_t)l);
            if (code[k] == (uint32_t)l) {
                const uint32_t at = pos[l] + (uint32_t)__popcll(m & lt);
                if (at < pl.cap) pl.q_idx[(size_t)l * pl.cap + at] = e;
            }
            pos[l] += (uint32_t)__popcll(m);
        }
    }
}

// ------------------------------------------------------------------------------------------ connect
// Bucket b's rays sit at q_sray[b * cap ...]; ray number idx of the launch counts through the buckets laid end to end.
__device__ __forceinline__ size_t sray_entry(const Pool &pl, const ShadowCounts &sn, uint32_t idx) {
    uint32_t rem = idx, b = 0;
#pragma unroll
    for (int q = 0; q < NSB - 1; ++q) if (b == (uint32_t)q && rem >= sn.n[q]) { rem -= sn.n[q]; b = q + 1; }
    return (size_t)b * pl.cap + rem;
}
__device__ __forceinline__ uint64_t sray_total(const ShadowCounts &sn) {
    uint64_t n = 0;
#pragma unroll
    for (int k = 0; k < NSB; ++k) n += sn.n[k];
    return n;
}

// One-shot waves on a fixed grid, for launches too small for the vote-scheduled kernel.
template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void render_connect_kernel(DevBVH bvh, Pool pl, int cur, uint32_t vote_from, unsigned long long *counters) {
    __shared__ uint32_t smem[STACK * WAVE];
    const ShadowCounts sn = pl.it->sray;
    const uint64_t n = sray_total(sn);
    if (n >= vote_from) return;
    TravCounters tc;
    for (uint32_t base = blockIdx.x * WAVE; base < n; base += gridDim.x * WAVE) {
        const uint32_t j = base + threadIdx.x;
        if (j >= n) continue;
        const size_t e = sray_entry(pl, sn, j);
        const float4 a = pl.q_sray_a[e], bb = pl.q_sray_b[e];
        pl.q_sray_occ[e] = any_hit<WAVE, COUNT, true>(bvh, smem + threadIdx.x, a.x, a.y, a.z, bb.x, bb.y, bb.z, a.w, tc) ? 1 : 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&counter_shard(counters)[CNT_ANY], (unsigned long long)n);
    flush_walk_counters<COUNT, true>(tc, counters);
}

// connect: persistent waves walking vote-scheduled single steps (vote_walk, trav_device.hpp), shadow rays dealt in
// chunks from one cursor over the buckets laid end to end (a chunk is nearly always of one kind).
template <int STACK, bool COUNT>
__global__ __launch_bounds__(WAVE) void render_connect_vote_kernel(DevBVH bvh, Pool pl, int cur, uint32_t vote_from, unsigned long long *counters,
                                                                   int refill_min, int tri_bias, uint32_t chunk_max) {
    __shared__ uint32_t smem[STACK * WAVE];
    const ShadowCounts sn = pl.it->sray;
    const uint64_t n = sray_total(sn);
    if (n < vote_from || (uint64_t)blockIdx.x * WAVE >= n) return;
    TravCounters cnt;
#ifdef PTMI_PROBE_OCCLUDED
    uint32_t occ_rays = 0, occ_inner = 0, occ_tri = 0, mark_i = 0, mark_t = 0;
#endif
    // The walk takes the far child first (vote_inner_step_any: the occlusion bit does not depend on the order); the
    // counting variant walks in the same order, so the render leg's roofline prices the work that is done.
    // (Tried: a per-lane cache of the last occluding triangle, tested before the walk.  connect 125.2 -> 124.3 ms per C3
    // frame, 171.1 -> 171.0 on the 10.4 M-triangle scene: the cache rarely holds the next ray's occluder.)
    vote_walk<WAVE, true, COUNT, uint32_t, false, true>(
        bvh, smem + threadIdx.x, n, pl.qcnt + (size_t)QC_CURSOR * QC_STRIDE, refill_min, tri_bias, vote_chunk(n, chunk_max), cnt,
        [&](uint64_t idx, NextRay &nx, uint32_t &pos) {
            const size_t e = sray_entry(pl, sn, (uint32_t)idx);
            const float4 a = pl.q_sray_a[e], bb = pl.q_sray_b[e];
            nx.ox = a.x; nx.oy = a.y; nx.oz = a.z; nx.tmax = a.w; nx.dx = bb.x; nx.dy = bb.y; nx.dz = bb.z; pos = (uint32_t)e;
        },
#ifdef PTMI_PROBE_OCCLUDED
        [&](uint32_t pos, const RayTrav &r) {          // experiment build: the occluded rays' share of the walk (counting variant)
            pl.q_sray_occ[pos] = r.best ? 1 : 0;
            if (r.best) { occ_rays++; occ_inner += cnt.inner - mark_i; occ_tri += cnt.tri - mark_t; }
            mark_i = cnt.inner; mark_t = cnt.tri;
        });
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { occ_rays += __shfl_xor(occ_rays, off); occ_inner += __shfl_xor(occ_inner, off); occ_tri += __shfl_xor(occ_tri, off); }
        if (threadIdx.x == 0) {
            unsigned long long *cs = counter_shard(counters);
            atomicAdd(&cs[CNT_VISIT_RET], (unsigned long long)occ_rays); atomicAdd(&cs[CNT_VISIT_DIFF], (unsigned long long)occ_inner);
            atomicAdd(&cs[CNT_VISIT_GEN], (unsigned long long)occ_tri);
        }
    }
#else
        [&](uint32_t pos, const RayTrav &r) { pl.q_sray_occ[pos] = r.best ? 1 : 0; });   // the answer goes beside the ray (one byte)
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counter_shard(counters)[CNT_ANY], (unsigned long long)n);
    flush_walk_counters<COUNT, true>(cnt, counters);
}

// ------------------------------------------------------------------------------------------ shade

struct Hit {            // a frame's hit, re-derived from the frame's stored ray + (t, prim)
    V3 ro, rd, P, normal, A, B, C;
    V3 dnormal, albedo;  // what HandleDiffuseMaterial shades with (bump-mapped normal, textured albedo: tracer.go:468-514)
    int prim;
    uint32_t mat;
};

struct STri { V3 A, B, C, N0, N1, N2; uint32_t mat; };
__device__ __forceinline__ STri load_stri(const DevScene &sc, uint32_t prim) {
    const float4 *p = sc.stri + (size_t)prim * 5;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
    STri t;
    t.A = mk(q0.x, q0.y, q0.z); t.B = mk(q0.w, q1.x, q1.y); t.C = mk(q1.z, q1.w, q2.x);
    t.N0 = mk(q2.y, q2.z, q2.w); t.N1 = mk(q3.x, q3.y, q3.z); t.N2 = mk(q3.w, q4.x, q4.y);
    t.mat = f2u(q4.z);
    return t;
}
// (Tried: a material record as four 16-byte loads instead of field-by-field dword loads through a reference: shade +13 ms
// per C3 frame -- the kernels read 5-7 of its 16 words, and the bytes returned count as much as the instructions.)
__device__ __forceinline__ const PtMaterial &load_material(const DevScene &sc, uint32_t i) { return sc.mats[i]; }
__device__ __forceinline__ bool is_emissive(const PtMaterial &m) { return m.ke[0] > 0.0f || m.ke[1] > 0.0f || m.ke[2] > 0.0f; }

// (re)derives the hit of a frame from what the frame stores: ray, t, primitive, and whether the glass branch flipped
// the normal (tracer.go:37-53, :468-514)
__device__ __forceinline__ void derive_hit(const DevScene &sc, V3 ro, V3 rd, float t, int prim, bool flipped, Hit &h, uint32_t &err) {
    h.ro = ro; h.rd = rd;
    h.prim = prim;
    if ((uint32_t)h.prim >= sc.ntris) { err |= ERR_PRIM; h.prim = 0; }
    h.P = add(h.ro, scale(h.rd, t));                                        // tracer.go:37
    STri tr = load_stri(sc, (uint32_t)h.prim);
    if (tr.mat >= sc.nmats) { err |= ERR_MAT; tr.mat = 0; }
    h.A = tr.A; h.B = tr.B; h.C = tr.C;
    V3 n = normalize(interpolate_normal(h.P, h.A, h.B, h.C, tr.N0, tr.N1, tr.N2));   // :38-46
    if (flipped) n = scale(n, -1.0f);                                       // :53
    h.normal = n;
    h.mat = tr.mat;
    h.dnormal = n;
    diffuse_surface(sc.mats[h.mat], sc.uvs, sc.tex, sc.ntex, sc.tex_pixels, (uint32_t)h.prim, h.P, h.A, h.B, h.C,
                    h.albedo, h.dnormal);
}

// Direct lighting of HandleDiffuseMaterial (:523-618): draws in the reference's order and hands every shadow ray to
// stage(kind, direction, range, term) -- `term` is what the final sum adds if the ray is free (lights: albedo *
// Sample(...), :557-559).  The terms travel with the path's next queue entry and are summed at its next visit, once
// `connect` has answered the rays.  Returns the kinds staged (bit k = kind k); sh_o / sh_eo = the rays' origins (sky
// and light rays; emissive sample).
template <typename Stage>
__device__ __forceinline__ uint32_t direct_lighting(const DevScene &sc, const RenderArgs &ra, const Hit &h, Rng &rng, V3 &sh_o, V3 &sh_eo,
                                                    uint32_t &err, Stage stage) {
    const V3 albedo = h.albedo;
    const V3 rayOrigin = add(h.P, scale(h.dnormal, (float)0.001));                       // :528
    uint32_t need = 0;
    sh_o = rayOrigin;
    if (sc.sky.kind != PTMI_SKY_NONE) {                                                 // :531-542
        V3 t1, t2; tangent_frame(h.dnormal, t1, t2);
        V3 rn = cosine_sample2(rng, h.dnormal, t1, t2);
        need |= 1u; stage(0, rn, 100000.0f, cmul(sky_sample(sc.sky, sc.sky_img, rn), albedo));
    }
    for (uint32_t li = 0; li < sc.nlights; ++li) {                                      // :545-560
        const PtLight &L = sc.lights[li];
        if (L.kind == PTMI_LIGHT_SUN) {                                                 // lights.go:18-28
            V3 ld = mk(L.vec);
            float ndotr = dot(ld, h.dnormal);
            if (!(ndotr < 0.0f)) { need |= 1u << (1 + li); stage(1 + (int)li, ld, ra.p.step_size, cmul(albedo, scale(mk(L.color), ndotr * L.intensity))); }
        } else {                                                                        // lights.go:38-54
            V3 ld = normalize(sub(mk(L.vec), rayOrigin));
            V3 toLight = sub(mk(L.vec), rayOrigin);
            float distance = length(toLight);
            toLight = normalize(toLight);
            float ndotl = dot(h.dnormal, toLight);
            if (!(ndotl <= 0.0f)) {
                float att = 1.0f / (distance * distance);
                need |= 1u << (1 + li); stage(1 + (int)li, ld, distance, cmul(albedo, scale(mk(L.color), ndotl * L.intensity * att)));
            }
        }
    }
    if (sc.nemissive > 0) {                                                             // :562-618
        V3 eo = add(h.P, scale(h.dnormal, (float)0.01));
        sh_eo = eo;
        uint32_t choice = rng_intn(rng, sc.nemissive);
        uint32_t et = sc.emissive[choice];
        if (et >= sc.ntris) { err |= ERR_PRIM; et = 0; }
        STri e = load_stri(sc, et);
        if (e.mat >= sc.nmats) { err |= ERR_MAT; e.mat = 0; }
        V3 lightPoint = sample_triangle_point(rng, e.A, e.B, e.C);
        V3 lsn = interpolate_normal(lightPoint, e.A, e.B, e.C, e.N0, e.N1, e.N2);
        V3 toLight = sub(lightPoint, eo);
        float distance = length(toLight);
        toLight = normalize(toLight);
        float ndotl = dot(toLight, h.dnormal);
        if (!(ndotl <= 0.0f)) {
            V3 ec = mk(0, 0, 0);
            float sndorl = -dot(toLight, lsn);
            if (!(sndorl <= 0.0f)) {
                float geometryTerm = ndotl * sndorl / (distance * distance);
                float pdf = triangle_area(e.A, e.B, e.C);
                pdf = 1.0f / (pdf * (float)sc.nemissive);
                float pdf_brdf = dot(h.dnormal, toLight) / PT_PI32;
                float pdf_sa = pdf * (distance * distance) / sndorl;
                float weight = mis_weight(pdf_sa, pdf_brdf);
                V3 le = mk(load_material(sc, e.mat).ke);
                V3 brdf = scale(albedo, (float)(1.0 / 3.14159265358979323846));
                ec = scale(cmul(le, brdf), geometryTerm * weight / pdf);
            }
            need |= 1u << (1 + sc.nlights); stage(1 + (int)sc.nlights, toLight, distance - (float)0.01, ec);
        }
    }
    return need;
}

// HandleDiffuseMaterial's final sum (:658-667): direct lighting (the free shadow rays' terms), ambient, the averaged
// GI rays, and the emitter seen directly.
__device__ __forceinline__ V3 diffuse_final(const DevScene &sc, const RenderArgs &ra, uint32_t fmat, int bounces, int bindex, V3 direct,
                                            V3 gi_sum, V3 albedo) {
    const PtMaterial &m = load_material(sc, fmat);
    V3 indirect = bounces > 0 ? gi_sum : mk(0, 0, 0);
    V3 fin = direct;
    fin = add(fin, scale(albedo, ra.p.ambient));
    fin = add(fin, scale(indirect, 1.0f));
    if (bindex == 0 && is_emissive(m)) fin = add(fin, mk(m.ke));
    return fin;
}
// TraceRay's component sum (:218-223)
__device__ __forceinline__ V3 combine_components(V3 diffuse, V3 reflective, V3 refr, float dg) {
    float diffuseScale = 1.0f;
    if (length(refr) > 0.0f) diffuseScale = (float)0.1;
    V3 fin = add(add(add(scale(diffuse, diffuseScale), mk(0, 0, 0)), reflective), refr);
    return scale(fin, dg);
}
// HandleReflectiveMaterial's result once its rays are in (:716-721), plus the refraction TraceRay adds to it (:165, :211)
__device__ __forceinline__ V3 reflective_final(const DevScene &sc, const RenderArgs &ra, uint32_t fmat, V3 sum, V3 refr) {
    const PtMaterial &m = load_material(sc, fmat);
    sum = scale(sum, 1.0f / (float)ra.p.scatter_rays);
    sum = cmul(sum, mk(m.ks));
    return add(sum, refr);
}

// The caller's direct lighting a queue entry carries: the contributions of the shadow rays connect found free, added in
// the order of HandleDiffuseMaterial (sky sample :538-540, lights :557-559, emissive sample :617).  term(k) = (term.xyz,
// bits(position of ray k in the shadow-ray queue)).
template <typename Term>
__device__ __forceinline__ V3 sum_free_terms(const Pool &pl, uint32_t &pneed, uint32_t &err, Term term) {
    V3 pdirect = mk(0, 0, 0);
    if (pneed >> pl.nshadow) { err |= ERR_PHASE; pneed = 0; }
    for (int k = 0; k < pl.nshadow; ++k) {
        if (!(pneed >> k & 1u)) continue;
        const float4 v = term(k);
        uint32_t qi = f2u(v.w);
        if (qi >= pl.cap * (uint32_t)pl.nshadow) { err |= ERR_SLOT; qi = 0; }
        if (pl.q_sray_occ[qi] == 0) pdirect = add(pdirect, mk(v.x, v.y, v.z));
    }
    return pdirect;
}
// The caller's direct lighting a RECORD carries.  The inline terms sit in words 19 .. 30 (not 16-byte aligned): the
// record's second sector is read as four aligned 16-byte loads -- only when a term arrives -- and the terms are picked
// out of them (twelve scattered dword loads per visit when measured against this: the vector-memory path is what the
// shade kernels share with nothing to spare).
__device__ __forceinline__ V3 sum_record_terms(const Pool &pl, const float4 *rec, uint32_t &pneed, uint32_t &err) {
    if (pneed >> pl.nshadow) { err |= ERR_PHASE; pneed = 0; }
    V3 pdirect = mk(0, 0, 0);
    if (!pneed) return pdirect;
    auto take = [&](float4 v) {
        uint32_t qi = f2u(v.w);
        if (qi >= pl.cap * (uint32_t)pl.nshadow) { err |= ERR_SLOT; qi = 0; }
        if (pl.q_sray_occ[qi] == 0) pdirect = add(pdirect, mk(v.x, v.y, v.z));
    };
    const float4 q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
    if (pneed & 1u) take(make_float4(q4.w, q5.x, q5.y, q5.z));
    if (pneed & 2u) take(make_float4(q5.w, q6.x, q6.y, q6.z));
    if (pl.rec_inline > 2 && (pneed & 4u)) take(make_float4(q6.w, q7.x, q7.y, q7.z));
    for (int k = pl.rec_inline; k < pl.nshadow; ++k)
        if (pneed >> k & 1u) take(rec[REC_BASE / 4 + (k - pl.rec_inline)]);
    return pdirect;
}

// States of the per-path machine inside one shade invocation.
enum { ST_EXTENDED, ST_MATERIAL, ST_SHADOW_DONE, ST_GI_LOOP, ST_REFL_LOOP, ST_DIFFUSE_FINAL, ST_COMBINE, ST_RETURN };

// LDS of one shade workgroup: the top frame of every thread (FRW words) and its staged shadow rays (SRW words each),
// [word][thread].  160 KB per CU / 3 workgroups = 106 allocation granules of 512 bytes = 53 words per thread: a scene
// with three shadow-ray kinds (sky sample, one light, emissive sample) takes 31 + 20 = 51 -- measured, 53 words already
// leave only two workgroups resident (tools/lds_probe.hip: the occupancy API still says three) -- which is why the
// workgroup's queue bookkeeping aliases the frame area (dead once a wave has left the state machine), the flags share
// a frame word with the bounce counters, and neither the rays' origins (they follow from the hit point) nor the sky
// sample's constant range are staged.
// staged per shadow ray: direction, contribution, range -- the sky sample (kind 0) has a constant range (:538) and
// takes 6 words, every other kind 7
enum { SRW = 7 };
__device__ __forceinline__ int stage_word(int k) { return k == 0 ? 0 : SRW * k - 1; }
constexpr float SKY_RAY_RANGE = 100000.0f;

// ---- queue bookkeeping shared by the shade kernels ---------------------------------------------------------------
// One atomic per workgroup and queue: ballot + popcount per wave, the four waves summed in LDS.  Queues: 0 extend,
// 1 shadow-wait (lean), 2 shadow-wait (general, filled from the back), 3 free list, 4.. the NSB shadow-ray buckets.
enum { BK_EXT = 0, BK_SHP = 1, BK_SHA = 2, BK_FREE = 3, BK_RAY = 4, BK_N = BK_RAY + NSB };
struct TailOut {              // what one path leaves a shade kernel with
    bool want_extend = false, want_shadow = false, finished = false, anc = false;
    uint32_t need_out = 0;    // kinds of the shadow rays it staged
    uint32_t slot = 0;
};
struct TailPos {              // where its entries go
    uint32_t ext = 0, shp = 0, fin = 0;
    bool shp_back = false;
    uint32_t ray[NSB - 1] = {};   // position in bucket b of its ray of kind b
    uint32_t ray_last = 0;        // first position in the shared last bucket
};
// sq: LDS words [BK_N][4] (as float bit patterns: one access type per LDS word); the workgroup's waves must all call.
__device__ __forceinline__ TailPos tail_reserve(const Pool &pl, int cur, float *sq, const TailOut &o, uint32_t diffuse_events,
                                                unsigned long long *counters, int visit_counter, bool visited, uint32_t frames_read) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool shp_lean = o.want_shadow && !o.anc, shp_anc = o.want_shadow && o.anc;
    const unsigned long long m_ext = __ballot(o.want_extend), m_shp = __ballot(shp_lean), m_sha = __ballot(shp_anc), m_fin = __ballot(o.finished);
    // shadow rays by bucket: kinds 0 .. NSB-2 are one ray per lane at most (ballot), the last bucket can take several
    const uint32_t last_bits = o.need_out >> (NSB - 1);
    uint32_t incl = (uint32_t)__popc(last_bits);
    if (pl.nshadow >= NSB) {
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
    }
    unsigned long long m_ray[NSB - 1];
#pragma unroll
    for (int b = 0; b < NSB - 1; ++b) m_ray[b] = __ballot((o.need_out >> b) & 1u);
    for (int off = 32; off > 0; off >>= 1) { diffuse_events += __shfl_xor(diffuse_events, off); frames_read += __shfl_xor(frames_read, off); }
    const uint32_t nvisited = (uint32_t)__popcll(__ballot(visited));
#define SQ(q, w) f2u(sq[(q) * 4 + (w)])
#define SQ_SET(q, w, v) (sq[(q) * 4 + (w)] = u2f(v))
    if (lane == 63) {
        SQ_SET(BK_EXT, wv, (uint32_t)__popcll(m_ext)); SQ_SET(BK_SHP, wv, (uint32_t)__popcll(m_shp)); SQ_SET(BK_SHA, wv, (uint32_t)__popcll(m_sha));
        SQ_SET(BK_FREE, wv, (uint32_t)__popcll(m_fin));
#pragma unroll
        for (int b = 0; b < NSB - 1; ++b) SQ_SET(BK_RAY + b, wv, (uint32_t)__popcll(m_ray[b]));
        SQ_SET(BK_RAY + NSB - 1, wv, pl.nshadow >= NSB ? incl : 0u);
    }
    lds_barrier();
    if (threadIdx.x < BK_N) {
        const int q = threadIdx.x;
        const uint32_t c0 = SQ(q, 0), c1 = SQ(q, 1), c2 = SQ(q, 2), c3 = SQ(q, 3);
        const uint32_t total = c0 + c1 + c2 + c3;
        const int nxt = cur ^ 1;
        const int w = q == BK_EXT ? QC_EXT0 + nxt : q == BK_SHP ? QC_SHP0 + nxt : q == BK_SHA ? QC_SHA0 + nxt : q == BK_FREE ? QC_FREE : QC_SRAY + (q - BK_RAY);
        uint32_t base = 0;
        if (total) base = (uint32_t)atomicAdd(&pl.qcnt[(size_t)w * QC_STRIDE], (unsigned long long)total);
        SQ_SET(q, 0, base); SQ_SET(q, 1, base + c0); SQ_SET(q, 2, base + c0 + c1); SQ_SET(q, 3, base + c0 + c1 + c2);
        if (q == BK_FREE && total) atomicAdd(&counter_shard(counters)[CNT_SAMPLES], (unsigned long long)total);
    }
    if (lane == 0) {
        unsigned long long *cs = counter_shard(counters);
        if (diffuse_events) atomicAdd(&cs[CNT_DIFFUSE], (unsigned long long)diffuse_events);
#if defined(PTMI_PROBE_OCCLUDED)
        (void)nvisited;                 // the probe borrows the three visit counters
#elif defined(PTMI_PROBE_LOCALITY)
        if (nvisited && visit_counter == CNT_VISIT_RET) atomicAdd(&cs[visit_counter], (unsigned long long)nvisited);
#else
        if (nvisited) atomicAdd(&cs[visit_counter], (unsigned long long)nvisited);
#endif
        if (frames_read) atomicAdd(&cs[CNT_FRAMES], (unsigned long long)frames_read);
    }
    lds_barrier();
    TailPos p;
    p.ext = SQ(BK_EXT, wv) + (uint32_t)__popcll(m_ext & lt);
    p.shp_back = shp_anc;
    p.shp = shp_anc ? SQ(BK_SHA, wv) + (uint32_t)__popcll(m_sha & lt) : SQ(BK_SHP, wv) + (uint32_t)__popcll(m_shp & lt);
    p.fin = SQ(BK_FREE, wv) + (uint32_t)__popcll(m_fin & lt);
#pragma unroll
    for (int b = 0; b < NSB - 1; ++b) p.ray[b] = SQ(BK_RAY + b, wv) + (uint32_t)__popcll(m_ray[b] & lt);
    p.ray_last = SQ(BK_RAY + NSB - 1, wv) + (incl - (uint32_t)__popc(last_bits));
#undef SQ
#undef SQ_SET
    return p;
}
// entry of the shadow-wait queue at position `pos` of its side (lean from the front, general from the back)
__device__ __forceinline__ size_t shp_entry(const Pool &pl, uint32_t pos, bool back) { return back ? (size_t)pl.cap - 1 - pos : (size_t)pos; }

// Writes a path's staged shadow rays into their buckets and hands back the term (contribution, queue position) of each
// through put(k, term).  s_ray = the staging area ([word][thread] columns of 256).
// (Tried: the ray carries the address of its term and connect stores the answer there, so that the summing visit needs
// no gather -- shade -5 ms per C3 frame, connect +9: a 4-byte store into a record of its own sector is a read-modify-
// write of that sector, where the answers by ray position are nearly sequential.)
template <typename Put>
__device__ __forceinline__ void flush_shadow_rays(const Pool &pl, const float *s_ray, uint32_t need_out, const TailPos &tp, V3 sh_o, V3 sh_eo, Put put) {
    uint32_t pos_last = tp.ray_last;
    for (int k = 0; k < pl.nshadow; ++k) {
        if (!(need_out >> k & 1u)) continue;
        uint32_t pos;
        if (k < NSB - 1) {
            pos = 0;
#pragma unroll
            for (int b = 0; b < NSB - 1; ++b) if (b == k) pos = tp.ray[b];
        } else {
            pos = pos_last++;
        }
        const int b = k < NSB - 1 ? k : NSB - 1;
        const size_t qi = (size_t)b * pl.cap + pos;
        const V3 o = k == pl.nshadow - 1 ? sh_eo : sh_o;            // the emissive sample starts from its own offset (:563)
        const float *sr = s_ray + (size_t)stage_word(k) * 256 + threadIdx.x;
        pl.q_sray_a[qi] = make_float4(o.x, o.y, o.z, k ? sr[1536] : SKY_RAY_RANGE);
        pl.q_sray_b[qi] = make_float4(sr[0], sr[256], sr[512], 0.0f);
        put(k, make_float4(sr[768], sr[1024], sr[1280], u2f((uint32_t)qi)));
    }
}

// ------------------------------------------------------------------------------------------ shade: returning calls
// List L_RET -- calls that return without shading a hit: the entry test failed (tracer.go:16), the march ended in the
// horizon (:253-255) or in nothing (:260-263: the sky) -- and the lean side of the shadow-wait queue: frames parked
// for their shadow rays (bounces <= 0, so HandleDiffuseMaterial goes straight to its final sum).  With scatterRays == 1
// and no refraction frame below (the host and classify() route everything else to the general kernel) such a path
// unwinds all the way to its primary sample: every frame it returns to is waiting for its only GI or reflection ray
// (:640-647, :715-721), finishes, and returns in turn.  Per level: one 128-byte frame line read, a dozen additions.
// No LDS frame, no staging area: the kernel runs at full occupancy.
__global__ __launch_bounds__(256) void shade_ret_kernel(Pool pl, DevScene sc, RenderArgs ra, int cur, unsigned long long *counters) {
    const uint32_t na = (uint32_t)pl.qcnt[(size_t)(QC_LIST + L_RET) * QC_STRIDE], nb = pl.it->n_shp;
    const uint32_t na_pad = (na + 63u) & ~63u;                  // the parked frames start on a wave boundary
    if (blockIdx.x * blockDim.x >= na_pad + nb) return;
    __shared__ float s_q[BK_N * 4];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool pending = i < na, parked = i >= na_pad && i - na_pad < nb;
    TailOut to;
    uint32_t diffuse_events = 0, frames_read = 0;
    if (pending || parked) {
        uint32_t err = 0, sid = 0, slot = 0;
        int depth = 0;
        V3 ret = mk(0, 0, 0), pd = mk(0, 0, 0);
        auto frame_at = [&](int dep) -> const float4 * { return pl.fr + ((size_t)slot * pl.nframes + dep) * 8; };
        if (pending) {
            uint32_t e = pl.q_idx[(size_t)L_RET * pl.cap + i];
            if (e >= pl.it->n_ext) { err |= ERR_QUEUE; e = 0; }
            const float4 *rec = rec_at(pl, cur, e);
            const RecHead hd = call_head(pl, ra, cur, e);
            const float4 r0 = hd.r0, r1 = hd.r1, r3 = hd.r3;
            slot = f2u(r0.w) & (uint32_t)SLOT_MASK;
            const uint32_t ctx = f2u(r0.w) >> CTX_SHIFT;
            sid = f2u(r1.w);
            const uint32_t status = f2u(pl.q_hit[e].x) >> 30;
            depth = (int)(f2u(r3.x) & 0xFFu);
            uint32_t pneed = f2u(r3.y) >> 8;
            if (slot >= pl.cap) { err |= ERR_SLOT; slot = 0; }
            if (depth >= pl.nframes) { err |= ERR_DEPTH; depth = 0; }
            pd = sum_record_terms(pl, rec, pneed, err);
            if (status == (uint32_t)HS_HIT && !(ctx & CTX_FAIL)) err |= ERR_CLASS;          // a hit to shade does not belong here
            if (!(ctx & CTX_FAIL) && status == (uint32_t)HS_MISS && sc.sky.kind != PTMI_SKY_NONE)
                ret = sky_sample(sc.sky, sc.sky_img, mk(r1.x, r1.y, r1.z));                 // :260-263
        } else {
            const size_t k = i - na_pad;
            const float4 e0 = pl.q_shpath[cur][2 * k], e1 = pl.q_shpath[cur][2 * k + 1];
            slot = f2u(e0.x); sid = f2u(e0.y);
            depth = (int)(f2u(e1.x) & 0xFFu);
            uint32_t pneed = f2u(e1.z);
            if (slot >= pl.cap) { err |= ERR_SLOT; slot = 0; }
            if (depth >= pl.nframes) { err |= ERR_DEPTH; depth = 0; }
            const float4 *terms = pl.q_shp_sh[cur] + k * (size_t)pl.nshadow;
            const V3 direct = sum_free_terms(pl, pneed, err, [&](int t) { return terms[t]; });
            // the parked frame: HandleDiffuseMaterial's final sum with no GI rays (bounces <= 0), then TraceRay's (:218-223)
            frames_read++;
            const float4 *f = frame_at(depth);
            const float4 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
            const uint32_t bb = f2u(f0.x), sp = f2u(f0.y);
            uint32_t fmat = f2u(f0.z);
            if (fmat >= sc.nmats) { err |= ERR_MAT; fmat = 0; }
            const int bounces = (int)(int16_t)(bb & 0xFFFFu), bindex = (int)((bb >> 16) & 0xFFu);
            if (((sp >> 24) & PH_MASK) != (uint32_t)PH_SHADOW_WAIT) err |= ERR_PHASE;
            if (bounces > 0) err |= ERR_CLASS;
            V3 gi = mk(0, 0, 0);
            if (!(sp >> 24 & PH_SHORT)) { const float4 f4 = f[4]; gi = mk(f4.x, f4.y, f4.z); }
            const V3 albedo = mk(f1.x, f1.y, f1.z), refr = mk(f3.y, f3.z, f3.w);
            const V3 fin = diffuse_final(sc, ra, fmat, bounces, bindex, direct, gi, albedo);
            diffuse_events++;
            ret = combine_components(fin, mk(0, 0, 0), refr, f0.w);
            pd = mk(f1.w, f2.x, f2.y);
        }
        // unwind: hand `ret` to the caller frame, and with it the caller's direct lighting this call carried
        while (depth > 0 && !err) {
            depth--;
            frames_read++;
            const float4 *f = frame_at(depth);
            const float4 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
            const uint32_t bb = f2u(f0.x), sp = f2u(f0.y);
            uint32_t fmat = f2u(f0.z);
            if (fmat >= sc.nmats) { err |= ERR_MAT; fmat = 0; }
            const int bounces = (int)(int16_t)(bb & 0xFFFFu), bindex = (int)((bb >> 16) & 0xFFu);
            const uint32_t ph = (sp >> 24) & PH_MASK, s = sp & 0xFFFFFFu;
            const float dg = f0.w;
            V3 fdiff = mk(f2.z, f2.w, f3.x);
            const V3 refr = mk(f3.y, f3.z, f3.w), albedo = mk(f1.x, f1.y, f1.z);
            V3 fsum = mk(0, 0, 0);
            if (!(sp >> 24 & PH_SHORT)) { const float4 f4 = f[4]; fsum = mk(f4.x, f4.y, f4.z); }
            if (s != 0u) err |= ERR_CLASS;                                  // scatterRays == 1: the frame waits for its only ray
            if (ph == (uint32_t)PH_GI_WAIT) {                               // :640-647, :658-667
                fdiff = pd;                                                 // the first scattered ray carried this frame's shadow terms
                fsum = add(fsum, cmul(ret, albedo));
                fsum = scale(fsum, 1.0f / (float)ra.p.scatter_rays);
                const V3 fin = diffuse_final(sc, ra, fmat, bounces, bindex, fdiff, fsum, albedo);
                diffuse_events++;
                ret = combine_components(fin, mk(0, 0, 0), refr, dg);
            } else if (ph == (uint32_t)PH_REFL_WAIT) {                      // :715-721
                fsum = add(fsum, ret);
                ret = combine_components(fdiff, reflective_final(sc, ra, fmat, fsum, refr), refr, dg);
            } else {
                err |= ERR_PHASE;
            }
            pd = mk(f1.w, f2.x, f2.y);
        }
        if (err) { atomicOr(&counter_shard(counters)[CNT_ERROR], (unsigned long long)err); ret = mk(0, 0, 0); }
        float *o = pl.out + 3 * (size_t)(sid < ra.total ? sid : 0);          // (one 12-byte store instead: shade +12 ms per C3 frame)
        o[0] = ret.x; o[1] = ret.y; o[2] = ret.z;
        to.finished = true; to.slot = slot;
    }
#ifdef PTMI_PROBE_LOCALITY
    {   // experiment build: how many 12 MB windows of the frame area do the slots of a wave fall into?
        const bool act = pending || parked;
        const uint32_t win = to.slot >> 15;
        const unsigned long long m = __ballot(act);
        uint32_t distinct = 0;
        bool mine = act;
        for (;;) {
            const unsigned long long b = __ballot(mine);
            if (!b) break;
            const uint32_t w = (uint32_t)__shfl((int)win, __ffsll((long long)b) - 1);
            if (mine && win == w) mine = false;
            distinct++;
        }
        if ((threadIdx.x & 63) == 0 && m) {
            atomicAdd(&counter_shard(counters)[CNT_VISIT_DIFF], 1ull);
            atomicAdd(&counter_shard(counters)[CNT_VISIT_GEN], (unsigned long long)distinct);
        }
    }
#endif
    const TailPos tp = tail_reserve(pl, cur, s_q, to, diffuse_events, counters, CNT_VISIT_RET, pending || parked, frames_read);
    if (to.finished) pl.q_free[cur ^ 1][tp.fin] = to.slot;
}

// ------------------------------------------------------------------------------------------ shade: fresh diffuse hits
// List L_DIFF -- a pending call whose ray hit a plain diffuse surface (no glass, no accretion disk, reflectivity < 0.1,
// not an emitter seen by an indirect ray): the entry of TraceRay (:36-48), HandleDiffuseMaterial's direct lighting
// (:523-618) and its first GI ray (:620-647), or, with no bounces left, the wait for the shadow rays.  Straight-line
// code: the frame is put together in registers and parked, the child (or the shadow-wait entry) goes into its queue.
// LDS: the staged shadow rays ([word][thread], as in the general kernel), reused as the exchange area of the whole-line
// stores once the rays are out: 16 words per thread at least, 20 with three shadow-ray kinds -- four workgroups per CU
// and more where the general kernel has three.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void shade_diff_kernel(Pool pl, DevScene sc, RenderArgs ra, int cur, unsigned long long *counters) {
    const uint32_t n = (uint32_t)pl.qcnt[(size_t)(QC_LIST + L_DIFF) * QC_STRIDE];
    if (blockIdx.x * blockDim.x >= n) return;
    __shared__ float s_q[BK_N * 4];
    extern __shared__ float s_ray[];           // max(SRW * nshadow - 1, XCH_WORDS) * 256 floats
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tid = threadIdx.x;
    TailOut to;
    V3 sh_o = mk(0, 0, 0), sh_eo = mk(0, 0, 0);
    float fw[32];                              // the frame this visit parks (F_*), word 31 unused
    float rw[20];                              // the child's record, words 0 .. 18 (its terms follow from the staged rays)
#pragma unroll
    for (int k = 0; k < 32; ++k) fw[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 20; ++k) rw[k] = 0.0f;
    float4 *frame_dst = nullptr;
    uint32_t sid = 0, depth_trk = 0, rng_key = 0, rng_ctr = 0;
    float trk0 = 1.0f;
    if (i < n) {
        uint32_t err = 0;
        uint32_t e = pl.q_idx[(size_t)L_DIFF * pl.cap + i];
        if (e >= pl.it->n_ext) { err |= ERR_QUEUE; e = 0; }
        const float4 *rec = rec_at(pl, cur, e);
        const RecHead hd = call_head(pl, ra, cur, e);
        const float4 r0 = hd.r0, r1 = hd.r1, r2 = hd.r2, r3 = hd.r3;
        const float4 r4 = (f2u(r3.y) & FL_SHORT) ? make_float4(0, 0, 0, 0) : rec[4];
        uint32_t slot = f2u(r0.w) & (uint32_t)SLOT_MASK;
        if (slot >= pl.cap) { err |= ERR_SLOT; slot = 0; }
        sid = f2u(r1.w);
        Rng rng; rng.key = f2u(r2.z); rng.ctr = f2u(r2.w);
        const float2 hq = pl.q_hit[e];
        const uint32_t hitword = f2u(hq.x), packed = f2u(r3.x), fword = f2u(r3.y);
        int depth = (int)(packed & 0xFFu);
        if (depth >= pl.nframes) { err |= ERR_DEPTH; depth = 0; }
        uint32_t trk_len = (packed >> 8) & 0xFFu;
        if ((int)trk_len > pl.ntrk) { err |= ERR_DEPTH; trk_len = 0; }
        const int bounces = (int)(int8_t)((packed >> 16) & 0xFFu), bindex = (int)(packed >> 24);
        const uint32_t flags = fword & 0xFFu & ~(uint32_t)(FL_FLIPPED | FL_SHORT);
        uint32_t pneed = fword >> 8;
        const float energy = r3.z;
        trk0 = r3.w;
        const V3 ln = mk(r4.x, r4.y, r4.z);
        const V3 pdirect = sum_record_terms(pl, rec, pneed, err);
        if ((hitword >> 30) != (uint32_t)HS_HIT || bounces < 0 || energy < (float)1e-2) err |= ERR_CLASS;
        const int hprim = (int)(hitword & 0x3FFFFFFFu);
        const V3 cur_o = mk(r0.x, r0.y, r0.z), cur_d = mk(r1.x, r1.y, r1.z);
        // ---- TraceRay :36-48, :93-113 (no disk), :49-91 (no glass), :116-118 (reflectivity < 0.1: HandleDiffuseMaterial)
        Hit h;
        derive_hit(sc, cur_o, cur_d, hq.y, hprim, false, h, err);
        const PtMaterial &m = sc.mats[h.mat];
        const float reflectivity = (m.ks[0] + m.ks[1] + m.ks[2]) / 3.0f;
        if ((m.flags & PTMI_MAT_GLASS) || ((m.flags & PTMI_MAT_ACCRETION) && sc.has_bh) || !(reflectivity < (float)0.1) ||
            (bindex > 0 && is_emissive(m)))
            err |= ERR_CLASS;
        // ---- HandleDiffuseMaterial :523-618
        const uint32_t need = direct_lighting(sc, ra, h, rng, sh_o, sh_eo, err, [&](int k, V3 d, float range, V3 term) {
            float *sr = s_ray + (size_t)stage_word(k) * 256 + tid;
            sr[0] = d.x; sr[256] = d.y; sr[512] = d.z;
            sr[768] = term.x; sr[1024] = term.y; sr[1280] = term.z;
            if (k) sr[1536] = range;
        });
        uint32_t phase;
        if (bounces > 0) {                                                              // :622-647: the first (s = 0) GI ray
            V3 t1, t2; tangent_frame(h.dnormal, t1, t2);
            const V3 dir = cosine_sample2(rng, h.dnormal, t1, t2);
            const V3 co = add(h.P, scale(h.dnormal, (float)0.001));
            phase = PH_GI_WAIT;
            // the child: TraceRay(bounces-1, bounceIndex+1, lastSurfaceNormal = normal, not specular, the frame's energy)
            const uint32_t cflags = flags & (uint32_t)FL_ANC;
            rw[R_O] = co.x; rw[R_O + 1] = co.y; rw[R_O + 2] = co.z;
            rw[R_SLOT] = u2f(slot | call_ctx(bounces - 1, bindex + 1, energy, cflags) << CTX_SHIFT);
            rw[R_D] = dir.x; rw[R_D + 1] = dir.y; rw[R_D + 2] = dir.z;
            rw[R_SID] = u2f(sid);
            rw[R_KEY] = u2f(rng.key); rw[R_CTR] = u2f(rng.ctr);
            rw[R_PACK] = u2f(pack_call(depth + 1, trk_len, bounces - 1, bindex + 1));
            rw[R_FLAGS] = u2f(cflags | need << 8);
            rw[R_ENERGY] = energy; rw[R_TRK0] = trk0;
            rw[R_LN] = h.dnormal.x; rw[R_LN + 1] = h.dnormal.y; rw[R_LN + 2] = h.dnormal.z;
            to.want_extend = true;
        } else {
            phase = PH_SHADOW_WAIT;
            if (!need) err |= ERR_CLASS;            // no ray to wait for: the general kernel's case (no sky, no light reaches, no emitter)
            to.want_shadow = true;
        }
        to.anc = (flags & FL_ANC) != 0 || ra.p.scatter_rays != 1;     // who unwinds a parked frame: shade_ret_kernel or the general kernel
        to.need_out = need;
        to.slot = slot;
        depth_trk = (uint32_t)depth | trk_len << 8;
        rng_key = rng.key; rng_ctr = rng.ctr;
        // ---- the frame (F_*)
        fw[F_BB] = u2f(((uint32_t)bounces & 0xFFFFu) | ((uint32_t)bindex & 0xFFu) << 16 | flags << 24);
        fw[F_SP] = u2f((phase | (ra.p.scatter_rays == 1 ? (uint32_t)PH_SHORT : 0u)) << 24);
        fw[F_MAT] = u2f(h.mat); fw[F_ENERGY] = energy;
        fw[F_LN] = ln.x; fw[F_LN + 1] = ln.y; fw[F_LN + 2] = ln.z;
        fw[F_RO] = cur_o.x; fw[F_RO + 1] = cur_o.y; fw[F_RO + 2] = cur_o.z;
        fw[F_RD] = cur_d.x; fw[F_RD + 1] = cur_d.y; fw[F_RD + 2] = cur_d.z;
        fw[F_T] = hq.y; fw[F_PRIM] = i2f(h.prim);
        fw[F_DG] = 1.0f;
        fw[F_ALBEDO] = h.albedo.x; fw[F_ALBEDO + 1] = h.albedo.y; fw[F_ALBEDO + 2] = h.albedo.z;
        fw[F_PDIRECT] = pdirect.x; fw[F_PDIRECT + 1] = pdirect.y; fw[F_PDIRECT + 2] = pdirect.z;
        frame_dst = pl.fr + ((size_t)slot * pl.nframes + depth) * 8;
        if (err) {          // self-check failed: report, drop the path with zero radiance
            atomicOr(&counter_shard(counters)[CNT_ERROR], (unsigned long long)err);
            float *o = pl.out + 3 * (size_t)(sid < ra.total ? sid : 0);
            o[0] = o[1] = o[2] = 0.0f;
            to.want_extend = to.want_shadow = false; to.need_out = 0; to.finished = true; frame_dst = nullptr;
        }
    }
    const TailPos tp = tail_reserve(pl, cur, s_q, to, 0u, counters, CNT_VISIT_DIFF, i < n, 0u);
    // shadow rays out of the staging area; their terms into the child's record (first rec_inline of them: registers,
    // stored with the record's lines) or beside the shadow-wait entry
    float4 *rec_dst = to.want_extend ? rec_at(pl, cur ^ 1, tp.ext) : nullptr;
    float tw[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) tw[k] = 0.0f;
    float4 *shp_terms = nullptr;
    if (to.want_shadow) {
        const size_t ps = shp_entry(pl, tp.shp, tp.shp_back);
        pl.q_shpath[cur ^ 1][2 * ps] = make_float4(u2f(to.slot), u2f(sid), u2f(rng_key), u2f(rng_ctr));
        pl.q_shpath[cur ^ 1][2 * ps + 1] = make_float4(u2f(depth_trk), trk0, u2f(to.need_out), u2f(to.anc ? (uint32_t)FL_ANC : 0u));
        shp_terms = pl.q_shp_sh[cur ^ 1] + ps * (size_t)pl.nshadow;
    }
    if (to.finished) pl.q_free[cur ^ 1][tp.fin] = to.slot;
    if (to.need_out) {
        const int ninl = pl.rec_inline;
        flush_shadow_rays(pl, s_ray, to.need_out, tp, sh_o, sh_eo, [&](int k, float4 term) {
            if (shp_terms) { shp_terms[k] = term; return; }
            if (k >= ninl) { rec_dst[REC_BASE / 4 + (k - ninl)] = term; return; }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j == k) { tw[4 * j] = term.x; tw[4 * j + 1] = term.y; tw[4 * j + 2] = term.z; tw[4 * j + 3] = term.w; }
        });
    }
    // whole-line stores: the parked frame and the child's record, each in two halves through the exchange area (the
    // staged rays are out; every lane only touches its own wave's columns)
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float *col = s_ray + tid;
    auto store_line = [&](float4 *dst, int halves, auto word) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half >= halves) break;
#pragma unroll
            for (int k = 0; k < 16; ++k) col[k * 256] = word(16 * half + k);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wave_store_half(col, dst, half);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    store_line(frame_dst, ra.p.scatter_rays == 1 ? 1 : 2, [&](int k) { return fw[k]; });      // a SHORT frame is its first sector
    store_line(rec_dst, 2, [&](int k) { return k < R_TERMS ? rw[k] : (k < R_TERMS + 12 ? tw[k - R_TERMS] : 0.0f); });
}

// ------------------------------------------------------------------------------------------ shade: the general kernel
// Every other class, one list each, laid end to end with each list padded to whole waves (a wave then holds one class:
// non-hits that unwind through refraction frames or scatterRays > 1, glass, reflective, mid-reflective, emitters hit by
// indirect rays, diffuse hits the lean kernel does not take, the accretion disk), then the general side of the
// shadow-wait queue.  Reads each entry's record (gathered), writes the next buffers.  The grid covers a full pool;
// workgroups beyond this iteration's work leave.
struct GenLists { uint32_t n[N_GEN]; };
__device__ __forceinline__ GenLists gen_lists(const Pool &pl) {
    GenLists g;
#pragma unroll
    for (int l = 0; l < N_GEN; ++l) g.n[l] = (uint32_t)pl.qcnt[(size_t)(QC_LIST + L_GEN + l) * QC_STRIDE];
    return g;
}
__device__ __forceinline__ uint32_t gen_padded_total(const GenLists &g) {
    uint32_t t = 0;
#pragma unroll
    for (int l = 0; l < N_GEN; ++l) t += (g.n[l] + 63u) & ~63u;
    return t;
}
// thread i of the general kernel -> (list, position); false: a padding lane
__device__ __forceinline__ bool gen_entry(const GenLists &g, uint32_t i, int &list, uint32_t &pos) {
    uint32_t rem = i;
    int l = 0;
#pragma unroll
    for (int q = 0; q < N_GEN - 1; ++q) {
        const uint32_t pad = (g.n[q] + 63u) & ~63u;
        if (l == q && rem >= pad) { rem -= pad; l = q + 1; }
    }
    list = L_GEN + l; pos = rem;
    return rem < g.n[l];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void shade_kernel(Pool pl, DevScene sc, RenderArgs ra, int cur,
                                                                                               unsigned long long *counters) {
    const GenLists gl = gen_lists(pl);
    const uint32_t na = gen_padded_total(gl), nb = pl.it->n_sha;
    if (blockIdx.x * blockDim.x >= na + nb) return;
    __shared__ __align__(16) float s_fw[FRW * 256];
    extern __shared__ float s_ray[];           // (SRW * nshadow - 1) * 256 floats
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int my_list = 0; uint32_t my_pos = 0;
    const bool pending = i < na && gen_entry(gl, i, my_list, my_pos);
    const bool live = pending || (i >= na && i - na < nb);
    uint32_t slot = 0;
    bool want_extend = false, want_shadow = false, finished = false, anc_out = false;
    uint32_t n_sray = 0;                       // shadow rays this path wants to enqueue
    uint32_t need_out = 0;                     // their kinds (bit k = shadow ray kind k)
    uint32_t diffuse_events = 0, frames_read = 0;
    V3 out_o = mk(0, 0, 0), out_d = mk(0, 0, 0);   // the ray of a path that wants an extend (goes into its queue entry)
    uint32_t out_sid = 0, out_key = 0, out_ctr = 0, out_pack = 0, out_flags = 0, out_ctx = 0, out_depth_trk = 0;
    float out_energy = 0.0f, out_trk0 = 1.0f;
    V3 out_ln = mk(0, 0, 0);
    V3 sh_o = mk(0, 0, 0), sh_eo = mk(0, 0, 0);    // origins of the staged shadow rays (sky / light rays; emissive sample)
    float4 *store_to = nullptr;                    // where the top frame (still in LDS) is parked when the path leaves
    if (live) {
        float4 qa = make_float4(0, 0, 0, 0), qb = qa;
        uint32_t pneed = 0;                                                 // shadow terms that arrive with this entry
        uint32_t err = 0;
        uint32_t sid, packed, in_flags = 0;
        Rng rng;
        float trk0, in_energy = 0.0f, ht = 0.0f, vt0 = 1.0f, vt1 = 1.0f;
        int hs = HS_MISS, hprim = -1, in_bounces = 0, in_bindex = 0;
        V3 in_ln = mk(0, 0, 0);
        V3 pdirect;
        if (pending) {
            uint32_t e = pl.q_idx[(size_t)my_list * pl.cap + my_pos];
            if (e >= pl.it->n_ext) { err |= ERR_QUEUE; e = 0; }
            const float4 *rec = rec_at(pl, cur, e);
            const RecHead hd = call_head(pl, ra, cur, e);
            qa = hd.r0; qb = hd.r1;
            const float4 r2 = hd.r2, r3 = hd.r3;
            const float4 r4 = (f2u(r3.y) & FL_SHORT) ? make_float4(0, 0, 0, 0) : rec[4];
            slot = f2u(qa.w) & (uint32_t)SLOT_MASK;
            sid = f2u(qb.w);
            const float2 hq = pl.q_hit[e];
            const uint32_t hitword = f2u(hq.x);
            hs = (int)(hitword >> 30);
            hprim = hs == HS_HIT ? (int)(hitword & 0x3FFFFFFFu) : -1;
            ht = hq.y;
            rng.key = f2u(r2.z); rng.ctr = f2u(r2.w);
            packed = f2u(r3.x);
            in_bounces = (int)(int8_t)((packed >> 16) & 0xFFu); in_bindex = (int)(packed >> 24);
            in_flags = f2u(r3.y) & 0xFFu;
            pneed = f2u(r3.y) >> 8;
            in_energy = r3.z; trk0 = r3.w;
            in_ln = mk(r4.x, r4.y, r4.z);
            if (sc.has_bh) { const float2 v = ((const float2 *)rec)[R_VT / 2]; vt1 = v.x; vt0 = v.y; }
            if (slot >= pl.cap) { err |= ERR_SLOT; slot = 0; }
            pdirect = sum_record_terms(pl, rec, pneed, err);
        } else {
            const size_t k = shp_entry(pl, i - na, true);
            const float4 e0 = pl.q_shpath[cur][2 * k], e1 = pl.q_shpath[cur][2 * k + 1];
            slot = f2u(e0.x);
            sid = f2u(e0.y); rng.key = f2u(e0.z); rng.ctr = f2u(e0.w);
            packed = f2u(e1.x);
            trk0 = e1.y;
            pneed = f2u(e1.z);
            if (slot >= pl.cap) { err |= ERR_SLOT; slot = 0; }
            const float4 *terms = pl.q_shp_sh[cur] + k * (size_t)pl.nshadow;
            pdirect = sum_free_terms(pl, pneed, err, [&](int t) { return terms[t]; });
        }
        int depth = (int)(packed & 0xFFu);
        if (depth >= pl.nframes) { err |= ERR_DEPTH; depth = 0; }
        uint32_t trk_len = (packed >> 8) & 0xFFu;
        if ((int)trk_len > pl.ntrk) { err |= ERR_DEPTH; trk_len = 0; }
        V3 cur_o = mk(qa.x, qa.y, qa.z), cur_d = mk(qb.x, qb.y, qb.z);      // the path's current ray (pending calls only)
        float *trk = pl.trk + (size_t)slot * pl.ntrk;
        // arguments of a child call pushed in this invocation (they go into its queue entry when the path parks)
        int arg_bounces = 0, arg_bindex = 0; uint32_t arg_flags = 0; float arg_energy = 0.0f; V3 arg_ln = mk(0, 0, 0);
        float4 *frames = pl.fr + (size_t)slot * pl.nframes * 8;

        // The top frame lives in LDS, [word][thread] (conflict-free): it is read and written all over the state
        // machine below, and 32 more loop-carried registers per lane are what this kernel cannot afford.
        const uint32_t tid = threadIdx.x;
#if PTMI_FRAME_REGS
        float fw_reg[FRW];
#define FW(field) fw_reg[(field)]
#else
#define FW(field) s_fw[(field) * 256 + tid]
#endif
        auto load_frame = [&](int dep) {
            frames_read++;
            const float4 *f = frames + (size_t)dep * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 v = f[k]; FW(4 * k) = v.x; FW(4 * k + 1) = v.y; FW(4 * k + 2) = v.z; FW(4 * k + 3) = v.w; }
            if (f2u(FW(F_SP)) >> 24 & PH_SHORT) {   // only the first sector is in memory: the sum is zero, the hit is never needed again
                FW(F_SUM) = 0.0f; FW(F_SUM + 1) = 0.0f; FW(F_SUM + 2) = 0.0f;
                return;
            }
#pragma unroll
            for (int k = 4; k < 8; ++k) { const float4 v = f[k]; FW(4 * k) = v.x; FW(4 * k + 1) = v.y; FW(4 * k + 2) = v.z; if (4 * k + 3 < FRW) FW(4 * k + 3) = v.w; }
        };
        // A frame is parked (under a child, or to wait for its shadow rays) only as the LAST thing a visit does, so the
        // store is deferred to the end of the kernel, where the whole wave writes the parked frames together: 8 lanes per
        // 128-byte line (one lane storing its own line takes eight instructions that each touch 16 bytes of 64 lines).
        auto store_frame = [&](int dep) {
#if PTMI_FRAME_REGS
            float4 *f = frames + (size_t)dep * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = make_float4(FW(4 * k), FW(4 * k + 1), FW(4 * k + 2), 4 * k + 3 < FRW ? FW(4 * k + 3) : 0.0f);
#else
            store_to = frames + (size_t)dep * 8;
#endif
        };
#define FV(field) mk(FW(field), FW((field) + 1), FW((field) + 2))
#define FV_SET(field, v) do { V3 v__ = (v); FW(field) = v__.x; FW((field) + 1) = v__.y; FW((field) + 2) = v__.z; } while (0)
#define FI(field) f2i(FW(field))
#define FU(field) f2u(FW(field))
#define FI_SET(field, v) (FW(field) = i2f(v))
#define FU_SET(field, v) (FW(field) = u2f(v))
        // bounces | bounceIndex << 16 | flags << 24 and sample counter | phase << 24 share a word each
        auto f_bounces = [&]() -> int { return (int)(int16_t)(FU(F_BB) & 0xFFFFu); };
        auto f_bindex = [&]() -> int { return (int)((FU(F_BB) >> 16) & 0xFFu); };
        auto f_flags = [&]() -> uint32_t { return FU(F_BB) >> 24; };
        auto set_flags = [&](uint32_t fl) { FU_SET(F_BB, (FU(F_BB) & 0xFFFFFFu) | fl << 24); };
        auto f_s = [&]() -> int { return (int)(FU(F_SP) & 0xFFFFFFu); };
        auto f_phase = [&]() -> uint32_t { return (FU(F_SP) >> 24) & PH_MASK; };
        auto set_s = [&](int s) { FU_SET(F_SP, (FU(F_SP) & 0xFF000000u) | ((uint32_t)s & 0xFFFFFFu)); };
        auto set_phase = [&](uint32_t ph) { FU_SET(F_SP, (FU(F_SP) & 0xFFFFFFu) | ph << 24); };
        // RefractiveIndexTracker.indices: entry 0 in a register (it travels with the path), the rest in Pool::trk
        auto trk_get = [&](uint32_t e) -> float { return e == 0 ? trk0 : trk[e - 1]; };
        auto trk_set = [&](uint32_t e, float v) { if (e == 0) trk0 = v; else trk[e - 1] = v; };
        auto trk_current = [&]() -> float { return trk_len == 0 ? 1.0f : trk_get(trk_len - 1); };
        auto trk_previous = [&]() -> float { return trk_len < 2 ? 1.0f : trk_get(trk_len - 2); };

        // (re)derive the hit of the top frame from what the frame stores
        auto load_hit = [&](Hit &h) { derive_hit(sc, FV(F_RO), FV(F_RD), FW(F_T), FI(F_PRIM), (f_flags() & FL_FLIPPED) != 0, h, err); };
        // park the top frame and make a child TraceRay invocation the pending call; the caller requests its extend.
        // A child inherits FL_ANC, and gets it when the frame it leaves behind waits for its refraction subtree.
        auto push_child = [&](int dep, int bounces, int bindex, V3 lastN, bool specular, float energy, V3 o, V3 d) {
            store_frame(dep);
            arg_bounces = bounces; arg_bindex = bindex;
            arg_flags = (specular ? (uint32_t)FL_SPECULAR : 0u) | ((f_flags() & FL_ANC) || f_phase() == (uint32_t)PH_REFR_WAIT ? (uint32_t)FL_ANC : 0u);
            arg_energy = energy; arg_ln = lastN;
            cur_o = o; cur_d = d;
        };

        int state;
        V3 ret = mk(0, 0, 0);
        Hit h;
        bool have_hit = false;
        // entry: a pending call starts from its arguments, a parked frame comes back with its shadow answers
        if (pending) {
            FU_SET(F_BB, ((uint32_t)in_bounces & 0xFFFFu) | ((uint32_t)in_bindex & 0xFFu) << 16 | (in_flags & ~(uint32_t)(FL_FLIPPED | FL_SHORT)) << 24);
            FW(F_ENERGY) = in_energy;
            FW(F_LN) = in_ln.x; FW(F_LN + 1) = in_ln.y; FW(F_LN + 2) = in_ln.z;
            FV_SET(F_PDIRECT, pdirect);                                     // handed back to the caller when this call returns
            FU_SET(F_SP, (uint32_t)PH_EXTEND << 24);
            state = ST_EXTENDED;
        } else {
            load_frame(depth);
            if (f_phase() == PH_SHADOW_WAIT) { FV_SET(F_DIFF, pdirect); state = ST_SHADOW_DONE; }
            else { err |= ERR_PHASE; state = ST_RETURN; depth = 0; }
        }
        for (;;) {
            if (err) {      // self-check failed: report, drop the path with zero radiance
                atomicOr(&counter_shard(counters)[CNT_ERROR], (unsigned long long)err);
                float *o = pl.out + 3 * (size_t)(sid < ra.total ? sid : 0);
                o[0] = o[1] = o[2] = 0.0f;
                want_extend = want_shadow = false; n_sray = 0; need_out = 0; finished = true; break;
            }
            if (state == ST_EXTENDED) {
                // TraceRay entry test (tracer.go:16) for the primary ray; children are tested before being pushed
                if (f_bounces() < 0 || FW(F_ENERGY) < (float)1e-2) { ret = mk(0, 0, 0); state = ST_RETURN; continue; }
                if (hs == HS_HORIZON) { ret = mk(0, 0, 0); state = ST_RETURN; continue; }           // :253-255
                if (hs == HS_MISS) {                                                                // :260-263
                    ret = sc.sky.kind == PTMI_SKY_NONE ? mk(0, 0, 0) : sky_sample(sc.sky, sc.sky_img, cur_d);
                    state = ST_RETURN; continue;
                }
                FV_SET(F_RO, cur_o); FV_SET(F_RD, cur_d);
                FW(F_T) = ht; FI_SET(F_PRIM, hprim);
                set_flags(f_flags() & ~(uint32_t)FL_FLIPPED);
                load_hit(h); have_hit = true;
                FV_SET(F_ALBEDO, h.albedo); FU_SET(F_MAT, h.mat);
                const PtMaterial &m = sc.mats[h.mat];
                // doppler / gravitational factors and the disk colour depend only on this hit (:93-113)
                float dg = 1.0f;
                V3 diff0 = mk(0, 0, 0);
                if ((m.flags & PTMI_MAT_ACCRETION) && sc.has_bh) {
                    V3 rel = sub(h.P, mk(sc.bh.position));
                    V3 tang = normalize(cross(mk(0, 1, 0), rel));
                    V3 vel = scale(tang, (float)0.9999);
                    float vpar = dot(scale(h.rd, -1.0f), vel);
                    float doppler = sqrtf((1.0f + vpar) / (1.0f - vpar));
                    float grav = vt0 / vt1;
                    dg = doppler * grav;
                    if (sc.bh.has_disk) diff0 = disk_color(sc.bh, h.P);
                }
                FW(F_DG) = dg; FV_SET(F_DIFF, diff0);
                FV_SET(F_REFR, mk(0, 0, 0));
                if (m.flags & PTMI_MAT_GLASS) {                                                     // :49-91
                    float ri = m.ni;
                    bool goingOut = false;
                    if (dot(h.normal, h.rd) > 0.0f) {
                        h.normal = scale(h.normal, -1.0f); goingOut = true; ri = trk_previous();
                        set_flags(f_flags() | FL_FLIPPED);
                    }
                    bool tir;
                    V3 rdir = get_refracted_ray(h.rd, h.normal, trk_current(), ri, tir);
                    V3 rorg = add(h.P, scale(rdir, (float)0.001));
                    const int bounces = f_bounces(), bindex = f_bindex();
                    const float energy = FW(F_ENERGY);
                    int cbi = bindex + 1;
                    if (!tir) {
                        if (goingOut) { if (trk_len > 0) trk_len--; } else if ((int)trk_len < pl.ntrk) { trk_set(trk_len, ri); trk_len++; }
                        cbi = bindex;
                    }
                    const float ce = energy * (float)0.95;
                    if (!(ce < (float)1e-2 || bounces - 1 < 0)) {
                        set_phase(PH_REFR_WAIT);
                        push_child(depth, bounces - 1, cbi, FV(F_LN), (f_flags() & FL_SPECULAR) != 0, ce, rorg, rdir);
                        depth++; want_extend = true; break;
                    }
                    // the child returns Vec3{} at its entry test: refraction = {} * energy
                    FV_SET(F_REFR, scale(mk(0, 0, 0), energy));
                    have_hit = false;       // the normal may have been flipped: re-derive the diffuse normal from the flag
                }
                state = ST_MATERIAL; continue;
            }
            if (state == ST_MATERIAL) {
                if (!have_hit) { load_hit(h); have_hit = true; }
                const PtMaterial &m = sc.mats[h.mat];
                const float reflectivity = (m.ks[0] + m.ks[1] + m.ks[2]) / 3.0f;                    // :116
                bool take_reflective;
                if (reflectivity < (float)0.1) take_reflective = false;
                else if (reflectivity < (float)0.9) take_reflective = rng_f32(rng) < reflectivity;  // :163
                else take_reflective = true;
                FV_SET(F_SUM, mk(0, 0, 0)); set_s(0);
                if (take_reflective) { state = ST_REFL_LOOP; continue; }
                const int bindex = f_bindex();
                if (bindex > 0 && is_emissive(m)) {                                                 // :461-465
                    V3 dc = mk(m.ke);
                    if (!(f_flags() & FL_SPECULAR)) {                                               // :143-158
                        float pdf_brdf = dot(h.rd, FV(F_LN)) / PT_PI32;
                        float area = triangle_area(h.A, h.B, h.C);
                        float pdf_area = 1.0f / ((float)sc.nemissive * area);
                        float cosLight = gomaxf(0.0f, dot(h.rd, h.normal));
                        float distance = length(sub(h.P, h.ro));
                        float pdf_sa = pdf_area * distance * distance / cosLight;
                        dc = scale(dc, mis_weight(pdf_brdf, pdf_sa));
                    }
                    FV_SET(F_DIFF, dc);
                    state = ST_COMBINE; continue;
                }
                // direct lighting of HandleDiffuseMaterial (:523-618): the shadow rays are staged in LDS
                const uint32_t need = direct_lighting(sc, ra, h, rng, sh_o, sh_eo, err, [&](int k, V3 d, float range, V3 term) {
                    float *sr = s_ray + (size_t)stage_word(k) * 256 + tid;
                    sr[0] = d.x; sr[256] = d.y; sr[512] = d.z;
                    sr[768] = term.x; sr[1024] = term.y; sr[1280] = term.z;
                    if (k) sr[1536] = range;
                    n_sray++;
                });
                need_out = need;
                FV_SET(F_DIFF, mk(0, 0, 0));         // this frame's direct lighting until the terms have been summed
                // A frame with a GI child to wait for anyway does not park for its shadow rays: they are traced while
                // the child is, their terms travel with the child's entry, and the child hands the sum back when it
                // returns (same draws, same sums, one visit less).
                if (need && f_bounces() <= 0) { set_phase(PH_SHADOW_WAIT); store_frame(depth); want_shadow = true; break; }
                state = f_bounces() > 0 ? ST_GI_LOOP : ST_DIFFUSE_FINAL; continue;                  // :622
            }
            if (state == ST_SHADOW_DONE) {
                state = f_bounces() > 0 ? ST_GI_LOOP : ST_DIFFUSE_FINAL; continue;
            }
            if (state == ST_GI_LOOP) {                                                              // :636-647
                const int s = f_s();
                if (s < ra.p.scatter_rays) {
                    if (!have_hit) { load_hit(h); have_hit = true; }
                    V3 t1, t2; tangent_frame(h.dnormal, t1, t2);
                    V3 dir = cosine_sample2(rng, h.dnormal, t1, t2);
                    // bounces > 0 and the energy is this frame's own (already >= 1e-2): the child always runs
                    set_phase(PH_GI_WAIT);
                    push_child(depth, f_bounces() - 1, f_bindex() + 1, h.dnormal, false,
                               FW(F_ENERGY), add(h.P, scale(h.dnormal, (float)0.001)), dir);
                    depth++; want_extend = true; break;
                }
                FV_SET(F_SUM, scale(FV(F_SUM), 1.0f / (float)ra.p.scatter_rays));
                state = ST_DIFFUSE_FINAL; continue;
            }
            if (state == ST_DIFFUSE_FINAL) {                                                        // :658-667
                uint32_t fmat = FU(F_MAT);
                if (fmat >= sc.nmats) { err |= ERR_MAT; fmat = 0; }
                const V3 fin = diffuse_final(sc, ra, fmat, f_bounces(), f_bindex(), FV(F_DIFF), FV(F_SUM), FV(F_ALBEDO));
                diffuse_events++;
                FV_SET(F_DIFF, fin);
                FV_SET(F_SUM, mk(0, 0, 0));          // reflective component stays zero on this branch
                state = ST_COMBINE; continue;
            }
            if (state == ST_REFL_LOOP) {                                                            // :670-722
                uint32_t fmat = FU(F_MAT);
                if (fmat >= sc.nmats) { err |= ERR_MAT; fmat = 0; }
                const PtMaterial &m = load_material(sc, fmat);
                const int bounces = f_bounces();
                bool pushed = false;
                while (f_s() < ra.p.scatter_rays) {
                    if (!have_hit) { load_hit(h); have_hit = true; }
                    float roughness = 1.0f / (1.0f + m.ns / 100.0f);
                    float dp = dot(h.rd, h.normal);
                    V3 reflDir = normalize(sub(h.rd, scale(h.normal, 2.0f * dp)));
                    V3 sd = sample_glossy(rng, reflDir, roughness);
                    if (bounces - 1 < 0) {          // the child returns Vec3{} at its entry test (:16)
                        FV_SET(F_SUM, add(FV(F_SUM), mk(0, 0, 0)));
                        set_s(f_s() + 1);
                        continue;
                    }
                    set_phase(PH_REFL_WAIT);
                    push_child(depth, bounces - 1, f_bindex() + 1, h.normal, true, FW(F_ENERGY),
                               add(h.P, scale(h.normal, (float)0.001)), sd);
                    depth++; want_extend = true; pushed = true; break;
                }
                if (pushed) break;
                FV_SET(F_SUM, reflective_final(sc, ra, fmat, FV(F_SUM), FV(F_REFR)));               // :716-721, :165, :211
                state = ST_COMBINE; continue;
            }
            if (state == ST_COMBINE) {                                                              // :218-223
                ret = combine_components(FV(F_DIFF), FV(F_SUM), FV(F_REFR), FW(F_DG));
                state = ST_RETURN; continue;
            }
            // ST_RETURN: hand `ret` to the caller frame, and with it the caller's direct lighting this call carried
            if (depth == 0) {
                float *o = pl.out + 3 * (size_t)sid;
                o[0] = ret.x; o[1] = ret.y; o[2] = ret.z;
                finished = true; break;
            }
            const V3 pd = FV(F_PDIRECT);
            depth--; have_hit = false;
            load_frame(depth);
            const uint32_t ph = f_phase();
            if (ph == PH_REFR_WAIT) {                                                               // :78, :89
                FV_SET(F_REFR, scale(ret, FW(F_ENERGY)));
                state = ST_MATERIAL;
            } else if (ph == PH_GI_WAIT) {                                                          // :640-645
                const int s = f_s();
                if (s == 0) FV_SET(F_DIFF, pd);      // the first scattered ray carried this frame's shadow terms
                V3 c = cmul(ret, FV(F_ALBEDO));
                FV_SET(F_SUM, add(FV(F_SUM), c));
                set_s(s + 1);
                state = ST_GI_LOOP;
            } else {                                                                                // PH_REFL_WAIT :715
                FV_SET(F_SUM, add(FV(F_SUM), ret));
                set_s(f_s() + 1);
                state = ST_REFL_LOOP;
            }
        }
        if (err && !finished) {
            atomicOr(&counter_shard(counters)[CNT_ERROR], (unsigned long long)err);
            want_extend = want_shadow = false; n_sray = 0; need_out = 0; finished = true; store_to = nullptr;
        }
        if (!finished) {      // the state that travels with the path, for its next queue entry
            out_sid = sid; out_key = rng.key; out_ctr = rng.ctr; out_trk0 = trk0;
            out_depth_trk = (uint32_t)depth | trk_len << 8;
            // a parked frame is unwound by the lean kernel unless a refraction frame waits below it
            if (want_shadow) anc_out = (f_flags() & FL_ANC) != 0 || ra.p.scatter_rays != 1;
            // the child pushed right after staging carries the staged rays' terms (need_out), any other child none
            if (want_extend) {
                out_pack = pack_call(depth, trk_len, arg_bounces, arg_bindex);
                out_flags = arg_flags | need_out << 8;
                out_ctx = call_ctx(arg_bounces, arg_bindex, arg_energy, arg_flags);
                out_energy = arg_energy; out_ln = arg_ln;
                out_o = cur_o; out_d = cur_d;
            }
        }
#undef FW
#undef FV
#undef FV_SET
#undef FI
#undef FU
#undef FI_SET
#undef FU_SET
    }
    // ---- parked frames: the wave stores them together, 8 lanes per line (the frames are still in this wave's LDS
    // columns; the compaction below reuses them) ----
    wave_store_lines(s_fw + threadIdx.x, 256, store_to, FRW);
    // ---- queue compaction: one atomic per workgroup and queue (tail_reserve; its bookkeeping words have a small LDS
    // area of their own: other waves may still be in their state machines, so the frame area cannot lend them) ----
    __shared__ float s_q[BK_N * 4];
    TailOut to;
    to.want_extend = want_extend; to.want_shadow = want_shadow; to.finished = finished; to.anc = anc_out; to.need_out = need_out; to.slot = slot;
    const TailPos tp = tail_reserve(pl, cur, s_q, to, diffuse_events, counters, CNT_VISIT_GEN, live, frames_read);
    {
        // The child's record is put together in the frame area (this thread's column: the frame store above has read
        // it, and LDS accesses of one wave complete in order) and stored as a whole line, like the frames.
        float *w = s_fw + threadIdx.x;
        float4 *rec_dst = nullptr, *shp_terms = nullptr;
        if (want_extend) {
            rec_dst = rec_at(pl, cur ^ 1, tp.ext);
            w[R_O * 256] = out_o.x; w[(R_O + 1) * 256] = out_o.y; w[(R_O + 2) * 256] = out_o.z;
            w[R_SLOT * 256] = u2f(slot | out_ctx << CTX_SHIFT);
            w[R_D * 256] = out_d.x; w[(R_D + 1) * 256] = out_d.y; w[(R_D + 2) * 256] = out_d.z;
            w[R_SID * 256] = u2f(out_sid);
            w[R_HIT * 256] = 0.0f; w[R_T * 256] = 0.0f;
            w[R_KEY * 256] = u2f(out_key); w[R_CTR * 256] = u2f(out_ctr);
            w[R_PACK * 256] = u2f(out_pack); w[R_FLAGS * 256] = u2f(out_flags);
            w[R_ENERGY * 256] = out_energy; w[R_TRK0 * 256] = out_trk0;
            w[R_LN * 256] = out_ln.x; w[(R_LN + 1) * 256] = out_ln.y; w[(R_LN + 2) * 256] = out_ln.z;
            for (int k = R_TERMS; k < FRW; ++k) w[k * 256] = 0.0f;
        }
        if (want_shadow) {
            const size_t ps = shp_entry(pl, tp.shp, tp.shp_back);
            pl.q_shpath[cur ^ 1][2 * ps] = make_float4(u2f(slot), u2f(out_sid), u2f(out_key), u2f(out_ctr));
            pl.q_shpath[cur ^ 1][2 * ps + 1] = make_float4(u2f(out_depth_trk), out_trk0, u2f(need_out), u2f(anc_out ? (uint32_t)FL_ANC : 0u));
            shp_terms = pl.q_shp_sh[cur ^ 1] + ps * (size_t)pl.nshadow;
        }
        if (finished) pl.q_free[cur ^ 1][tp.fin] = slot;
        if (n_sray && (rec_dst || shp_terms)) {
            const int ninl = pl.rec_inline;
            flush_shadow_rays(pl, s_ray, need_out, tp, sh_o, sh_eo, [&](int k, float4 term) {
                if (shp_terms) { shp_terms[k] = term; return; }
                if (k >= ninl) { rec_dst[REC_BASE / 4 + (k - ninl)] = term; return; }
                float *t = w + (size_t)(R_TERMS + 4 * k) * 256;
                t[0] = term.x; t[256] = term.y; t[512] = term.z; t[768] = term.w;
            });
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes, before its lanes read them
        wave_store_lines(w, 256, rec_dst, FRW);
    }
}

// ------------------------------------------------------------------------------------------ accumulate
// Pixel.AddSample's running sums (renderer.go:29, 59-61): R += c.X ... in sample order, one thread per pixel.
// The per-sample output is pixel-major ([pixel][sample][3]) like the sample numbering, so read directly a thread's
// samples are 12 bytes every 3 KB (1.2 TB/s, 5.4 ms per C3 frame; staged: 2.8 ms).  The workgroup therefore stages tiles of ACC_TILE samples of its 256
// pixels through LDS: the loads walk each pixel's 12 * ACC_TILE contiguous bytes with consecutive lanes (whole sectors),
// and every thread then adds ITS pixel's samples from LDS in sample order -- the same additions in the same order.
// Tried and dropped: sample-major output (shade's writes of samples that finish together scatter instead: no gain) and
// sample-major numbering with it (neighbouring samples are then neighbouring pixels instead of one pixel's jittered
// rays, and the traversal kernels lose more coherence than accumulate gains: C3 -3 %, the 10 M-triangle scene -9 %).
// frame_width = 0: accum / counts are the region's own [h][w] buffers; otherwise they are full frames of that
// width and pixel (x, y) of the image is element y * frame_width + x.
enum { ACC_TILE = 16, ACC_ROW = 3 * ACC_TILE + 1 };       // + 1 word: rows 49 words apart, no bank conflicts
// STATS: also the rest of Pixel.AddSample (renderer.go:25-82) in stats[PTMI_PIXEL_STATS * pixel + ..]: MinLuminance,
// MaxLuminance (:63-78, colorToLuminance :84-86: the first sample a pixel ever gets sets both, later ones widen the range
// with the reference's `<` / `>`), the running Mean and M2 per channel (:33-51, Welford) and Variance (:53-57), each
// operation as the source writes it, in sample order.
template <bool STATS>
__global__ __launch_bounds__(256) void accumulate_kernel(const float *__restrict__ out, uint32_t npix, uint32_t spp, float *accum,
                                                         uint32_t *counts, RenderArgs ra, int frame_width, float *lum) {
    __shared__ float s_tile[256 * ACC_ROW];
    const uint32_t p0 = blockIdx.x * 256u, p = p0 + threadIdx.x;
    const uint32_t np = min(256u, npix - p0);                          // pixels of this workgroup
    const bool live = p < npix;
    size_t q = p;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    float lmin = 0.0f, lmax = 0.0f, mr = 0.0f, mg = 0.0f, mb = 0.0f, m2r = 0.0f, m2g = 0.0f, m2b = 0.0f, var = 0.0f;
    uint32_t have = 0;                                                 // the pixel's SampleCount
    if (live) {
        if (frame_width > 0) {
            int x, y;
            region_pixel(ra, p, x, y);
            q = (size_t)y * (size_t)frame_width + (size_t)x;
        }
        r = accum[3 * q]; g = accum[3 * q + 1]; b = accum[3 * q + 2];
        if (STATS && (have = counts[q]) != 0) {
            const float *st = lum + (size_t)PTMI_PIXEL_STATS * q;
            lmin = st[0]; lmax = st[1]; mr = st[2]; mg = st[3]; mb = st[4]; m2r = st[5]; m2g = st[6]; m2b = st[7]; var = st[8];
        }
    }
    const float *base = out + 3 * (size_t)p0 * spp;
    for (uint32_t s0 = 0; s0 < spp; s0 += ACC_TILE) {
        const uint32_t ns = min((uint32_t)ACC_TILE, spp - s0), row = 3 * ns;      // floats per pixel in this tile
        for (uint32_t j = threadIdx.x; j < np * row; j += 256) {
            const uint32_t px = j / row, k = j - px * row;
            s_tile[px * ACC_ROW + k] = base[(size_t)px * spp * 3 + (size_t)s0 * 3 + k];
        }
        __syncthreads();
        if (live) {
            const float *t = s_tile + threadIdx.x * ACC_ROW;
            for (uint32_t s = 0; s < ns; ++s) {
                r += t[3 * s]; g += t[3 * s + 1]; b += t[3 * s + 2];
                if (STATS) {
                    const float cx = t[3 * s], cy = t[3 * s + 1], cz = t[3 * s + 2];
                    ++have;                                                                      // :29
                    const float n = (float)have;
                    const float dx = cx - mr, dy = cy - mg, dz = cz - mb;                        // :33-37
                    mr += dx / n; mg += dy / n; mb += dz / n;                                    // :39-41
                    m2r += dx * (cx - mr); m2g += dy * (cy - mg); m2b += dz * (cz - mb);         // :43-51
                    if (have > 1) var = ((m2r + m2g) + m2b) / (3.0f * (n - 1.0f));               // :53-57
                    const float l = ((float)0.2126 * cx + (float)0.7152 * cy) + (float)0.0722 * cz;   // :84-86
                    if (have == 1) { lmin = lmax = l; }                                          // :67-70
                    else { if (l < lmin) lmin = l; if (l > lmax) lmax = l; }                     // :72-77
                }
            }
        }
        __syncthreads();
    }
    if (live) {
        accum[3 * q] = r; accum[3 * q + 1] = g; accum[3 * q + 2] = b;
        counts[q] += spp;
        if (STATS && spp) {
            float *st = lum + (size_t)PTMI_PIXEL_STATS * q;
            st[0] = lmin; st[1] = lmax; st[2] = mr; st[3] = mg; st[4] = mb; st[5] = m2r; st[6] = m2g; st[7] = m2b; st[8] = var;
        }
    }
}

// Tile.GetNoisiestPixel's convergence test on Pixel.Contrast = MaxLuminance - MinLuminance (tile.go:51, renderer.go:81):
// 1 = the pixel goes on, 0 = it is done (a NaN contrast is never < 0.01: the pixel goes on, as in the reference).
__global__ __launch_bounds__(256) void adaptive_mask_kernel(const float *__restrict__ stats, uint32_t npix, uint8_t *mask) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const float *st = stats + (size_t)PTMI_PIXEL_STATS * i;
    mask[i] = (st[1] - st[0] < (float)0.01) ? 0 : 1;
}

}  // namespace

// ============================================================================================ host side

struct RenderState {
    // scene (device)
    bool have_scene = false;
    float4 *d_stri = nullptr;            // per-triangle shading record (DevScene::stri)
    uint8_t *d_tri_class = nullptr;      // TC_* bits per triangle (DevScene::tri_class)
    uint32_t nmats = 0;
    float *d_uvs = nullptr; TexDesc *d_tex = nullptr; uint32_t *d_tex_pixels = nullptr; uint32_t ntex = 0;
    PtMaterial *d_mats = nullptr; uint32_t *d_emissive = nullptr; uint32_t nemissive = 0;
    PtLight *d_lights = nullptr; uint32_t nlights = 0;
    PtSky sky{}; uint16_t *d_sky_img = nullptr;
    PtBlackHole bh{}; int has_bh = 0;
    // pool (device), sized for (cap, bounces, nlights)
    char *pool_mem = nullptr; size_t pool_bytes = 0;
    uint32_t pool_cap = 0; int pool_bounces = -2, pool_nlights = -1, pool_lanes = 0, pool_curved = -1;
    enum { MAX_LANES = 4, QC_RING = 2 };
    Pool pools[MAX_LANES]{};             // one per lane (render_core)
    hipStream_t lane_stream[MAX_LANES] = {};   // lanes 1 .. run here, lane 0 on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {};
    // the host's view of the queue counters: per lane one pinned copy per batch of iterations in flight (render_core)
    unsigned long long *h_qc = nullptr;  // pinned, MAX_LANES x QC_RING x QC_COUNT * QC_STRIDE words
    hipEvent_t ev_qc[MAX_LANES][QC_RING] = {};
    float *d_out = nullptr; size_t out_bytes = 0;
    char *d_tiles = nullptr; size_t tiles_bytes = 0;   // tile list of ptmi_render_tiles_dev
    PtCamera *d_cams = nullptr; size_t cams_n = 0;     // cameras of ptmi_render_frames_dev
    float *d_accum = nullptr; uint32_t *d_counts = nullptr; size_t accum_pix = 0;
};

namespace ptmi {
// the mesh under the scene changed (a new BVH was uploaded): the per-triangle records, UVs and emissive list belong
// to the old mesh, so the scene has to be uploaded again before the next render call
void invalidate_scene(ptmi_ctx *ctx) {
    RenderState *rs = ctx->render;
    if (!rs) return;
    rs->have_scene = false;
    for (void *p : {(void *)rs->d_uvs, (void *)rs->d_tex, (void *)rs->d_tex_pixels}) if (p) (void)hipFree(p);
    rs->d_uvs = nullptr; rs->d_tex = nullptr; rs->d_tex_pixels = nullptr; rs->ntex = 0;
}
void free_scene(ptmi_ctx *ctx) {
    RenderState *rs = ctx->render;
    if (!rs) return;
    void *ptrs[] = {rs->d_uvs, rs->d_tex, rs->d_tex_pixels, rs->d_stri, rs->d_tri_class, rs->d_mats, rs->d_emissive, rs->d_lights,
                    rs->d_sky_img, rs->pool_mem, rs->d_out, rs->d_accum, rs->d_counts, rs->d_tiles, rs->d_cams};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &lane : rs->ev_qc) for (hipEvent_t e : lane) if (e) (void)hipEventDestroy(e);
    if (rs->ev_fork) (void)hipEventDestroy(rs->ev_fork);
    for (hipEvent_t e : rs->ev_join) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : rs->lane_stream) if (st) (void)hipStreamDestroy(st);
    if (rs->h_qc) (void)hipHostFree(rs->h_qc);
    delete rs;
    ctx->render = nullptr;
}
}  // namespace ptmi

template <typename T>
static int upload(ptmi_ctx *ctx, T *&dst, const T *src, size_t n) {
    if (dst) { (void)hipFree(dst); dst = nullptr; }
    PTMI_HIP(ctx, hipMalloc((void **)&dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) PTMI_HIP(ctx, hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PTMI_OK;
}

extern "C" int ptmi_upload_scene(ptmi_ctx *ctx, const PtMaterial *materials, uint64_t nmaterials,
                                 const uint32_t *tri_material, const float *corner_normals, const PtLight *lights,
                                 uint64_t nlights, const PtSky *sky, const uint16_t *sky_image,
                                 const PtBlackHole *blackhole) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_upload_scene: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_upload_scene: build or upload a BVH first");
    const uint64_t ntris = ctx->tri_idx.size() / 3;
    if (!materials || !nmaterials || (!tri_material && ntris) || (!corner_normals && ntris) || !sky || (!lights && nlights))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: null array");
    // shade stages a path's shadow rays (one per light + 2) in LDS: 7 KB per ray kind beside the 32 KB of frames
    if (nlights > 16) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: at most 16 lights");
    if (ntris >= (1ull << 30)) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: at most 2^30 - 1 triangles (a queue record keeps the primitive in 30 bits)");
    for (uint64_t t = 0; t < ntris; ++t)
        if (tri_material[t] >= nmaterials) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: material index out of range");
    if (sky->kind < 0 || sky->kind > 3) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: bad sky kind");
    if (sky->kind == PTMI_SKY_IMAGE && (!sky_image || sky->image_width < 1 || sky->image_height < 1))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_scene: image sky without image");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->render) ctx->render = new RenderState();
    RenderState *rs = ctx->render;
    rs->have_scene = false;
    int rc;
    {   // shading records: what shade needs about a hit triangle in one 80-byte gather (vnmu.Vertices / Normals /
        // Materials by tri.Index, tracer.go:38-48)
        std::vector<float> rec((size_t)ntris * 20 + 4, 0.0f);
        for (uint64_t t = 0; t < ntris; ++t) {
            float *r = rec.data() + t * 20;
            for (int c = 0; c < 3; ++c) {
                const float *v = ctx->verts.data() + 3 * (size_t)ctx->tri_idx[3 * t + c];
                r[3 * c] = v[0]; r[3 * c + 1] = v[1]; r[3 * c + 2] = v[2];
            }
            std::memcpy(r + 9, corner_normals + 9 * t, 9 * sizeof(float));
            const uint32_t mid = tri_material[t];
            std::memcpy(r + 18, &mid, 4);
        }
        float *d = nullptr;
        if ((rc = upload(ctx, d, rec.data(), rec.size()))) return rc;
        if (rs->d_stri) (void)hipFree(rs->d_stri);
        rs->d_stri = (float4 *)d;
    }
    {   // what extend needs to know about a hit triangle to name the entry's class (classify): the branch TraceRay takes
        // on its material -- glass by name (tracer.go:49), the accretion disk (:95, only with a black hole in the
        // scene), reflectivity = mean(Ks) against 0.1 / 0.9 (:116-163, the same float32 expression as the kernels'),
        // emitter (:461)
        std::vector<uint8_t> mc((size_t)nmaterials), tc((size_t)ntris + 1, 0);
        for (uint64_t k = 0; k < nmaterials; ++k) {
            const PtMaterial &m = materials[k];
            const float reflectivity = (m.ks[0] + m.ks[1] + m.ks[2]) / 3.0f;
            uint8_t c = 0;
            if (m.flags & PTMI_MAT_GLASS) c |= TC_GLASS;
            if ((m.flags & PTMI_MAT_ACCRETION) && blackhole) c |= TC_ACCR;
            if (!(reflectivity < (float)0.1)) c |= (reflectivity < (float)0.9) ? TC_MID : TC_REFL;
            if (m.ke[0] > 0.0f || m.ke[1] > 0.0f || m.ke[2] > 0.0f) c |= TC_EMIT;
            mc[k] = c;
        }
        for (uint64_t t = 0; t < ntris; ++t) tc[t] = mc[tri_material[t]];
        if ((rc = upload(ctx, rs->d_tri_class, tc.data(), tc.size()))) return rc;
        // ... and into the packed traversal records (classify: the vote-scheduled extend reads them from there)
        const uint32_t nrec = (uint32_t)ctx->packed_tris;
        if (nrec) {
            hipLaunchKernelGGL(patch_tri_class_kernel, dim3((nrec + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_tris, nrec, rs->d_tri_class, (uint32_t)ntris);
            PTMI_HIP(ctx, hipGetLastError());
            PTMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if ((rc = upload(ctx, rs->d_mats, materials, (size_t)nmaterials))) return rc;
    if ((rc = upload(ctx, rs->d_lights, lights, (size_t)nlights))) return rc;
    std::vector<uint32_t> em;                                   // renderer.go:311-321
    for (uint64_t t = 0; t < ntris; ++t) {
        const PtMaterial &m = materials[tri_material[t]];
        if (m.ke[0] > 0.0f || m.ke[1] > 0.0f || m.ke[2] > 0.0f) em.push_back((uint32_t)t);
    }
    if ((rc = upload(ctx, rs->d_emissive, em.data(), em.size()))) return rc;
    rs->nemissive = (uint32_t)em.size();
    rs->nlights = (uint32_t)nlights;
    rs->nmats = (uint32_t)nmaterials;
    rs->sky = *sky;
    if (sky->kind == PTMI_SKY_IMAGE) {
        if ((rc = upload(ctx, rs->d_sky_img, sky_image, (size_t)sky->image_width * sky->image_height * 3))) return rc;
    }
    rs->has_bh = blackhole ? 1 : 0;
    if (blackhole) rs->bh = *blackhole; else std::memset(&rs->bh, 0, sizeof rs->bh);
    rs->have_scene = true;
    return PTMI_OK;
}

extern "C" int ptmi_upload_textures(ptmi_ctx *ctx, const float *uvs, uint32_t ntextures, const int32_t *widths,
                                    const int32_t *heights, const uint32_t *pixels) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_upload_textures: ctx is null");
    RenderState *rs = ctx->render;
    if (!rs || !rs->have_scene) return set_err(ctx, PTMI_ERR_NO_SCENE, "ptmi_upload_textures: upload a scene first");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    PTMI_HIP(ctx, hipDeviceSynchronize());
    if (ntextures == 0) {
        for (void *p : {(void *)rs->d_uvs, (void *)rs->d_tex, (void *)rs->d_tex_pixels}) if (p) (void)hipFree(p);
        rs->d_uvs = nullptr; rs->d_tex = nullptr; rs->d_tex_pixels = nullptr; rs->ntex = 0;
        return PTMI_OK;
    }
    if (!uvs || !widths || !heights || !pixels) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_textures: null array");
    std::vector<TexDesc> desc(ntextures);
    uint64_t total = 0;
    for (uint32_t i = 0; i < ntextures; ++i) {
        if (widths[i] < 1 || heights[i] < 1) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_textures: empty image");
        if (total > 0xFFFFFFFFull) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_upload_textures: more than 2^32 texels");
        desc[i].offset = (uint32_t)total; desc[i].width = widths[i]; desc[i].height = heights[i]; desc[i].pad = 0;
        total += (uint64_t)widths[i] * (uint64_t)heights[i];
    }
    int rc;
    const size_t ntris = ctx->tri_idx.size() / 3;
    if ((rc = upload(ctx, rs->d_uvs, uvs, ntris * 6))) return rc;
    if ((rc = upload(ctx, rs->d_tex, desc.data(), desc.size()))) return rc;
    if ((rc = upload(ctx, rs->d_tex_pixels, pixels, (size_t)total))) return rc;
    rs->ntex = ntextures;
    return PTMI_OK;
}

static size_t carve(size_t &off, size_t bytes) {
    size_t o = off;
    off += (bytes + 255) / 256 * 256;
    return o;
}

static int tracker_words(int bounces) {
    // every TraceRay invocation can push one index and the sibling subtrees share the tracker, so its history is
    // bounded by the number of invocations, 2^(bounces+1)-1 (capped at 64 entries, like the oracle)
    int ntrk = bounces + 1 >= 6 ? 64 : (1 << (bounces + 1)) + 1;
    return (ntrk + 3) / 4 * 4;
}
// float4s of an extend-queue record: words 0 .. 31 hold the call and the first rec_inline shadow terms (a curved
// ray's record keeps extend's V_t pair in words 28-29 instead of a third term), the other terms follow from word 32;
// whole 64-byte sectors
static int record_inline_terms(bool curved) { return curved ? 2 : 3; }
static int record_quads(int nshadow, bool curved) {
    const int extra = std::max(0, nshadow - record_inline_terms(curved));
    return (REC_BASE + 4 * extra + 15) / 16 * 4;
}
// device bytes one path of the pool takes (records + its share of every queue)
static uint64_t pool_bytes_per_path(int bounces, int nlights, bool curved) {
    const uint64_t nframes = (uint64_t)bounces + 1, nshadow = (uint64_t)nlights + 2;
    // frames + tracker history + shadow-ray queue (ray, answer) and the shadow terms beside the shadow-wait queue (double
    // buffered) + extend queue (records, double buffered; class byte; index lists) + shadow-wait queue (double
    // buffered) + free list
    return 128 * nframes + 4 * (uint64_t)tracker_words(bounces) + (32 + 1 + 16 * 2) * nshadow + 16 * (uint64_t)record_quads((int)nshadow, curved) * 2 +
           8 + 1 + 4 * NLIST + 32 * 2 + 4 * 2 + 32;
}

static int ensure_pool(ptmi_ctx *ctx, uint32_t cap, int bounces, int lanes, bool curved) {
    RenderState *rs = ctx->render;
    if (rs->pool_mem && rs->pool_cap >= cap && rs->pool_bounces == bounces && rs->pool_nlights == (int)rs->nlights &&
        rs->pool_lanes >= lanes && rs->pool_curved == (int)curved)
        return PTMI_OK;                  // a larger pool of the same shape serves (tiles of different sizes)
    const int nframes = bounces + 1, nshadow = (int)rs->nlights + 2;
    const int ntrk = tracker_words(bounces);
    const int rec_q = record_quads(nshadow, curved);
    const size_t c = cap;
    struct Off { size_t fr, trk, qs0, qs1, ra, rb, occ, rec0, rec1, hit, cls, idx, ss0, ss1, free_, free1, fa, fb, cnt, it; } of[RenderState::MAX_LANES];
    size_t off = 0;
    for (int l = 0; l < lanes; ++l) {
        Off &o = of[l];
        o.fr = carve(off, 128 * c * (size_t)nframes); o.trk = carve(off, 4 * c * ntrk);
        o.qs0 = carve(off, 32 * c); o.qs1 = carve(off, 32 * c);
        o.ra = carve(off, 16 * c * nshadow); o.rb = carve(off, 16 * c * nshadow); o.occ = carve(off, c * nshadow);
        o.ss0 = carve(off, 16 * c * nshadow); o.ss1 = carve(off, 16 * c * nshadow);
        o.rec0 = carve(off, 16 * c * rec_q); o.rec1 = carve(off, 16 * c * rec_q);
        o.hit = carve(off, 8 * c); o.cls = carve(off, c + BIN_CHUNK); o.idx = carve(off, 4 * c * NLIST);
        o.free_ = carve(off, 4 * c); o.free1 = carve(off, 4 * c); o.fa = carve(off, 16 * c); o.fb = carve(off, 16 * c); o.cnt = carve(off, 8 * QC_COUNT * QC_STRIDE); o.it = carve(off, sizeof(IterArgs));
    }
    // the memory in place is laid out anew when it is large enough (another number of lanes or bounces, a pass under
    // ptmi_set_profiling: no render call is in flight here); a 120 GB hipMalloc takes seconds
    if (!rs->pool_mem || rs->pool_bytes < off) {
        if (rs->pool_mem) { (void)hipFree(rs->pool_mem); rs->pool_mem = nullptr; rs->pool_bytes = 0; rs->pool_cap = 0; }
        const size_t want = (off + (4u << 20)) >> 20 << 20;      // slack: the lanes' fixed blocks and alignment differ by a few KB between layouts
        if (hipMalloc((void **)&rs->pool_mem, want) != hipSuccess) {
            (void)hipGetLastError();
            rs->pool_mem = nullptr; rs->pool_cap = 0;
            return PTMI_ERR_OOM;
        }
        rs->pool_bytes = want;
    }
    char *b = rs->pool_mem;
    for (int l = 0; l < lanes; ++l) {
        const Off &o = of[l];
        Pool p{};
        p.cap = cap; p.nframes = nframes; p.nshadow = nshadow; p.ntrk = ntrk;
        p.rec_q = rec_q; p.rec_inline = record_inline_terms(curved);
        p.fr = (float4 *)(b + o.fr); p.trk = (float *)(b + o.trk);
        p.q_shp_sh[0] = (float4 *)(b + o.ss0); p.q_shp_sh[1] = (float4 *)(b + o.ss1);
        p.q_sray_occ = (uint8_t *)(b + o.occ);
        p.q_rec[0] = (float4 *)(b + o.rec0); p.q_rec[1] = (float4 *)(b + o.rec1);
        p.q_hit = (float2 *)(b + o.hit); p.q_cls = (uint8_t *)(b + o.cls); p.q_idx = (uint32_t *)(b + o.idx);
        p.q_shpath[0] = (float4 *)(b + o.qs0); p.q_shpath[1] = (float4 *)(b + o.qs1);
        p.q_sray_a = (float4 *)(b + o.ra); p.q_sray_b = (float4 *)(b + o.rb);
        p.q_free[0] = (uint32_t *)(b + o.free_); p.q_free[1] = (uint32_t *)(b + o.free1);
        p.q_fresh_a = (float4 *)(b + o.fa); p.q_fresh_b = (float4 *)(b + o.fb); p.qcnt = (unsigned long long *)(b + o.cnt);
        p.it = (IterArgs *)(b + o.it);
        rs->pools[l] = p;
    }
    rs->pool_cap = cap; rs->pool_bounces = bounces; rs->pool_nlights = (int)rs->nlights; rs->pool_lanes = lanes; rs->pool_curved = (int)curved;
    return PTMI_OK;
}

// A persistent wave of the vote-scheduled kernels reserves three chunks of queue entries before it starts (one in use,
// two ahead), so a launch needs at least 3 * 64 rays per resident wave for every wave to get work at all; measured
// (PTMI_RENDER_VOTE_MIN sweeps on C1, C3 and the 10 M-triangle scene, profiles/r02c) the vote-scheduled kernels are
// ahead from about four times that, 5 M rays on this chip (round 1, with scattered answer stores: eight times).  Below the
// bound the one-shot kernels run (the draining iterations at the end of a call, small tiles rendered one per call),
// above it the vote-scheduled ones with chunks of up to render_vote_chunk entries.  Both kernels are enqueued every
// iteration and the one the queue length does not belong to leaves at once (the length is only known on the device).
// PTMI_RENDER_VOTE_MIN overrides the bound (tests force the vote-scheduled kernels at small sizes with 0).
static uint32_t vote_from_for(const ptmi_ctx *ctx, int per_cu) {
    const uint64_t waves = (uint64_t)ctx->num_cus * (uint64_t)per_cu;
    const uint64_t least = ctx->render_vote_min >= 0 ? (uint64_t)ctx->render_vote_min : waves * 3 * WAVE * 4;
    return (uint32_t)std::min<uint64_t>(least, 0xFFFFFFFEull);
}
constexpr uint32_t VOTE_NEVER = 0xFFFFFFFFu;

// ---- per-kernel timing (ptmi_set_profiling) ----
enum { PROF_PREPARE = 0, PROF_GENERATE, PROF_EXTEND, PROF_CONNECT, PROF_SHADE, PROF_ACCUMULATE, PROF_CLASSES };
static hipEvent_t prof_event(ptmi_ctx *ctx) {
    if (!ctx->prof_free.empty()) { hipEvent_t e = ctx->prof_free.back(); ctx->prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
// brackets what `launch` enqueues on stream s with two events when profiling is on
template <typename F>
static void timed(ptmi_ctx *ctx, int cls, hipStream_t s, F launch) {
    if (!ctx->profiling) { launch(); return; }
    hipEvent_t a = prof_event(ctx), b = prof_event(ctx);
    if (a) (void)hipEventRecord(a, s);
    launch();
    if (b) (void)hipEventRecord(b, s);
    if (a && b) ctx->prof_open.push_back({cls, a, b});
    else { if (a) ctx->prof_free.push_back(a); if (b) ctx->prof_free.push_back(b); }
}
// after the pass's stream has been synchronised: turn the spans into milliseconds
static void prof_collect(ptmi_ctx *ctx) {
    for (const auto &sp : ctx->prof_open) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { ctx->prof_ms[sp.cls] += (double)ms; ctx->prof_launches++; }
        else (void)hipGetLastError();
        ctx->prof_free.push_back(sp.a); ctx->prof_free.push_back(sp.b);
    }
    ctx->prof_open.clear();
}

// One iteration of the wavefront loop on queue buffer `cur`, enqueued without knowing its sizes (IterArgs); the grids
// cover the most paths the call can have in flight (the pool, or all of the call's samples if those are fewer, or
// `live_bound`: what the host last saw alive or not yet started -- that number only falls, so an old look is a bound;
// it is what keeps the draining iterations from launching a pool's worth of workgroups that leave at once).
template <int STACK, bool COUNT>
static int enqueue_iteration_t(ptmi_ctx *ctx, const Pool &pl, const DevScene &sc, const RenderArgs &ra, const BinMap &bm, uint64_t lane_total,
                               uint64_t live_bound, int cur, hipStream_t s) {
    const uint64_t most = std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)pl.cap, lane_total, live_bound}));
    const unsigned cap_wgs = (unsigned)((most + 255) / 256), cap_waves = (unsigned)((most + WAVE - 1) / WAVE);
    if (ctx->profiling) ctx->prof_iterations++;
    timed(ctx, PROF_PREPARE, s, [&] {
        hipLaunchKernelGGL(iter_prep_kernel, dim3(1), dim3(64), 0, s, pl, cur, (unsigned long long)lane_total, ctx->d_counters); });
    timed(ctx, PROF_GENERATE, s, [&] {
        if (ra.implicit_fresh) hipLaunchKernelGGL(generate_rays_kernel, dim3(cap_wgs), dim3(256), 0, s, pl, ra, cur);
        else hipLaunchKernelGGL(generate_kernel, dim3(cap_wgs), dim3(256), 0, s, pl, ra, cur); });
    // extend
    uint32_t vote_from = VOTE_NEVER;
    if (!sc.has_bh && ra.p.max_steps == 1 && ctx->extend_vote) {
        const int per_cu = std::min(ctx->waves_per_cu((const void *)render_extend_vote_kernel<STACK, COUNT>, WAVE), ctx->persist_waves_per_cu);
        vote_from = vote_from_for(ctx, per_cu);
        const unsigned pgrid = (unsigned)std::min<uint64_t>(cap_waves, (uint64_t)ctx->num_cus * per_cu);
        if (most >= vote_from)          // a call this small never has a queue that long: one launch less per iteration
            timed(ctx, PROF_EXTEND, s, [&] {
                hipLaunchKernelGGL((render_extend_vote_kernel<STACK, COUNT>), dim3(pgrid), dim3(WAVE), 0, s, ctx->dev_bvh(), pl, ra, cur, vote_from,
                                   ctx->d_counters, ctx->render_vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->render_vote_chunk); });
    }
    if (vote_from > 0) {
        const unsigned grid = (unsigned)std::min<uint64_t>(cap_waves, ((uint64_t)vote_from + WAVE - 1) / WAVE);
        timed(ctx, PROF_EXTEND, s, [&] {
            if (sc.has_bh)
                hipLaunchKernelGGL((render_extend_kernel<STACK, true, COUNT>), dim3(grid), dim3(WAVE), 0, s, ctx->dev_bvh(), pl, sc, ra, cur,
                                   vote_from, ctx->d_counters);
            else
                hipLaunchKernelGGL((render_extend_kernel<STACK, false, COUNT>), dim3(grid), dim3(WAVE), 0, s, ctx->dev_bvh(), pl, sc, ra, cur,
                                   vote_from, ctx->d_counters); });
    }
    // connect
    const uint64_t ray_waves = (most * (uint64_t)pl.nshadow + WAVE - 1) / WAVE;
    vote_from = VOTE_NEVER;
    if (ctx->connect_vote) {
        const int per_cu = std::min(ctx->waves_per_cu((const void *)render_connect_vote_kernel<STACK, COUNT>, WAVE), ctx->persist_waves_per_cu);
        vote_from = vote_from_for(ctx, per_cu);
        const unsigned pgrid = (unsigned)std::min<uint64_t>(ray_waves, (uint64_t)ctx->num_cus * per_cu);
        if (most * (uint64_t)pl.nshadow >= vote_from)
            timed(ctx, PROF_CONNECT, s, [&] {
                hipLaunchKernelGGL((render_connect_vote_kernel<STACK, COUNT>), dim3(pgrid), dim3(WAVE), 0, s, ctx->dev_bvh(), pl, cur, vote_from,
                                   ctx->d_counters, ctx->render_vote_refill_min, ctx->vote_tri_bias, (uint32_t)ctx->render_vote_chunk); });
    }
    if (vote_from > 0) {
        const unsigned grid = (unsigned)std::min<uint64_t>(ray_waves, ((uint64_t)vote_from + WAVE - 1) / WAVE);
        timed(ctx, PROF_CONNECT, s, [&] {
            hipLaunchKernelGGL((render_connect_kernel<STACK, COUNT>), dim3(grid), dim3(WAVE), 0, s, ctx->dev_bvh(), pl, cur, vote_from, ctx->d_counters); });
    }
    // shade: the class bytes extend wrote become index lists, then one kernel per class (all three count as shade time)
    const size_t ray_lds = (size_t)(SRW * pl.nshadow - 1) * 256 * sizeof(float);
    const size_t diff_lds = std::max(ray_lds, (size_t)XCH_WORDS * 256 * sizeof(float));
    const unsigned bin_wgs = (unsigned)((most + BIN_CHUNK - 1) / BIN_CHUNK);
    const unsigned ret_wgs = (unsigned)((most + 64 + 255) / 256), gen_wgs = (unsigned)((most + 64 * N_GEN + 255) / 256);
    timed(ctx, PROF_SHADE, s, [&] { hipLaunchKernelGGL(bin_kernel, dim3(bin_wgs), dim3(256), 0, s, pl, bm); });
    timed(ctx, PROF_SHADE, s, [&] { hipLaunchKernelGGL(shade_diff_kernel, dim3(cap_wgs), dim3(256), diff_lds, s, pl, sc, ra, cur, ctx->d_counters); });
    timed(ctx, PROF_SHADE, s, [&] { hipLaunchKernelGGL(shade_ret_kernel, dim3(ret_wgs), dim3(256), 0, s, pl, sc, ra, cur, ctx->d_counters); });
    timed(ctx, PROF_SHADE, s, [&] { hipLaunchKernelGGL(shade_kernel, dim3(gen_wgs), dim3(256), ray_lds, s, pl, sc, ra, cur, ctx->d_counters); });
    PTMI_HIP(ctx, hipGetLastError());
    return PTMI_OK;
}
static int enqueue_iteration(ptmi_ctx *ctx, const Pool &pl, const DevScene &sc, const RenderArgs &ra, const BinMap &bm, uint64_t lane_total,
                             uint64_t live_bound, int cur, hipStream_t s) {
    const int dep = ctx->stack_entries();
    if (dep <= 16) return ctx->counting ? enqueue_iteration_t<16, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<16, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    if (dep <= 20) return ctx->counting ? enqueue_iteration_t<20, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<20, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    if (dep <= 24) return ctx->counting ? enqueue_iteration_t<24, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<24, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    if (dep <= 28) return ctx->counting ? enqueue_iteration_t<28, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<28, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    if (dep <= 32) return ctx->counting ? enqueue_iteration_t<32, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<32, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    if (dep <= 48) return ctx->counting ? enqueue_iteration_t<48, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<48, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
    return ctx->counting ? enqueue_iteration_t<64, true>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s) : enqueue_iteration_t<64, false>(ctx, pl, sc, ra, bm, lane_total, live_bound, cur, s);
}

// Renders samples [spp_begin, spp_begin+spp_count) of the rectangle into d_out chunks and accumulates them
// into d_accum / d_counts (device pointers); or, when out_samples_host != nullptr, copies raw samples out.
// A row-interleaved shard is given by row_block > 0 (then y0 = first row of the shard, h = its number of rows,
// row_period = row_block * shards, and d_accum / d_counts are full frames).
static int render_core(ptmi_ctx *ctx, const PtCamera *cam, const PtParams *prm, int x0, int y0, int w, int h,
                       uint32_t spp_begin, uint32_t spp_count, float *d_accum, uint32_t *d_counts, float *out_samples_host,
                       hipStream_t s, int row_block = 0, int row_period = 0, const uint32_t *d_tile_first = nullptr,
                       const int4 *d_tile_rect = nullptr, uint32_t ntiles = 0, const PtCamera *d_cams = nullptr, int frame_h = 0,
                       float *d_lum = nullptr) {
    RenderState *rs = ctx->render;
    if (!rs || !rs->have_scene) return set_err(ctx, PTMI_ERR_NO_SCENE, "ptmi_render: no scene uploaded");
    if (!cam || !prm) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: null camera / params");
    // a tile list: (x0, y0, w, h) = (0, 0, region pixels, 1), validated by the caller
    const bool rows_mode = row_block > 0 || ntiles > 0;
    if (w <= 0 || h <= 0 || prm->width <= 0 || prm->height <= 0 || x0 < 0 || y0 < 0 || (!ntiles && x0 + w > prm->width) ||
        (!rows_mode && !d_cams && y0 + h > prm->height))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: rectangle outside the image");
    if (!rows_mode || ntiles) { row_block = h; row_period = h; }
    else if (y0 + ((h - 1) / row_block) * row_period + (h - 1) % row_block >= prm->height)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: row shard outside the image");
    if (prm->bounces > 30 || prm->scatter_rays < 1 || prm->max_steps < 0)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: need bounces <= 30, scatter_rays >= 1, max_steps >= 0");
    if (spp_count == 0) return PTMI_OK;
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t npix = (uint64_t)w * h;
    if (npix * 3 >= 0xFFFFFFFFull) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: rectangle too large");
    const int bounces = std::max(prm->bounces, 0);

    // samples per pixel per batch: bound the per-sample output buffer (<= ~3 GiB, < 2^32/3 samples)
    uint32_t spp_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp_count, ((uint64_t)ctx->chunk_msamples << 20) / npix));
    // the per-sample output is laid out [pixel][sample] over the WHOLE sample range, which only one pass produces
    // (ptmi_render_samples caps a call at 2^28 samples, so the buffer stays bounded)
    if (out_samples_host) spp_chunk = spp_count;
    // The per-sample output buffer first (12 bytes per sample of a pass); when device memory is short the pass is
    // halved until it fits (more passes, same sums: accumulate adds them in sample order), PTMI_ERR_OOM when even one
    // sample per pixel does not -- or for ptmi_render_samples, whose layout needs the whole range in one pass.
    while (rs->out_bytes < npix * spp_chunk * 12) {
        if (rs->d_out) { (void)hipFree(rs->d_out); rs->d_out = nullptr; rs->out_bytes = 0; }
        if (hipMalloc((void **)&rs->d_out, npix * spp_chunk * 12) == hipSuccess) { rs->out_bytes = npix * spp_chunk * 12; break; }
        (void)hipGetLastError();
        rs->d_out = nullptr;
        if (out_samples_host || spp_chunk == 1)
            return set_err(ctx, PTMI_ERR_OOM, "ptmi_render: out of device memory for the per-sample output buffer (" +
                                              std::to_string(npix * spp_chunk * 12) + " bytes)");
        spp_chunk = (spp_chunk + 1) / 2;
    }
    const uint64_t max_total = npix * spp_chunk;
    // Every wavefront iteration costs time that does not depend on the number of paths (the tails of three kernels
    // that wait for their longest rays), so the pool is as large as the work and the memory budget allow: on C3 at
    // 256 spp 32 M paths reach 1540 Msamples/s, 64 M 1640, 128 M with the frame in one pass 1730.
    const bool curved = rs->has_bh != 0;
    const uint64_t path_bytes = pool_bytes_per_path(bounces, (int)rs->nlights, curved);
    const uint64_t by_targets = 0xFFFFFFFFull / ((uint64_t)(bounces + 1) * (rs->nlights + 2));   // shadow-ray queue positions are 32-bit
    const uint64_t by_budget = std::min(by_targets, std::max<uint64_t>(1u << 12, (ctx->pool_bytes_max - (5u << 20)) / path_bytes));   // 5 MiB: counters, alignment, slack
    // Two independent loops ("lanes"), each on half of the pool and of the samples and on a stream of its own, when
    // there is work for both: shade waits on memory most of the time while extend and connect keep the VALUs busy, so
    // one lane's traversal kernels fill the tails of the other's shade.  Measured on one box with the 128 M-path pool
    // (C3, 256 spp: 1633 / 1712 / 1701 / 1637 Msamples/s with 1 / 2 / 3 / 4 lanes; the 672 MB tree at 4K: 853 / 865 /
    // 870), so two lanes by default (render_lanes < 0) whatever the tree.  The image does not depend on the split.
    // (with the round's final kernels: C3 at 256 spp 1 846 / 1 877 / 1 731 Msamples/s with 2 / 3 / 4 lanes, the C4-class frame
    // 932 / 939 / 856, C3 at 32 spp 1 508 / 1 490 / 1 368: three lanes for passes of 192 M samples and more)
    const int want_lanes = ctx->render_lanes < 0 ? (max_total >= (192ull << 20) ? 3 : 2) : ctx->render_lanes;
    const int lanes = (want_lanes >= 2 && max_total >= (1ull << 22) && !ctx->profiling) ? std::min(want_lanes, (int)RenderState::MAX_LANES) : 1;
    // what the work asks for, rounded up; what the limits allow, rounded down (ptmi_set_pool_limit's are ceilings)
    // (the limit in whole blocks of 3 072 paths, so that the pool holds the same number of paths with 1, 2, 3 or 4 lanes and
    // a pass under ptmi_set_profiling -- one lane -- fits the memory the two-lane passes before it allocated)
    const uint64_t cap_limit = std::max<uint64_t>(256, std::min<uint64_t>((uint64_t)ctx->pool_paths, by_budget) / 3072 * 3072 / lanes);
    uint32_t cap = (uint32_t)std::min<uint64_t>(((max_total + lanes - 1) / lanes + 255) / 256 * 256, cap_limit);
    // the per-sample output buffer is in place (above); now as large a pool as fits
    int rc;
    while ((rc = ensure_pool(ctx, cap, bounces, lanes, curved)) == PTMI_ERR_OOM && cap > (1u << 16))
        cap = (cap / 2 + 255) / 256 * 256;          // device memory is short (other contexts on this GPU): halve
    if (rc == PTMI_ERR_OOM) return set_err(ctx, PTMI_ERR_OOM, "ptmi_render: out of device memory for the path pool");
    if (rc) return rc;
    if (!rs->h_qc) {
        for (auto &lane : rs->ev_qc) for (hipEvent_t &e : lane) PTMI_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        PTMI_HIP(ctx, hipEventCreateWithFlags(&rs->ev_fork, hipEventDisableTiming));
        for (int l = 1; l < RenderState::MAX_LANES; ++l) {
            PTMI_HIP(ctx, hipEventCreateWithFlags(&rs->ev_join[l], hipEventDisableTiming));
            PTMI_HIP(ctx, hipStreamCreateWithFlags(&rs->lane_stream[l], hipStreamNonBlocking));
        }
        PTMI_HIP(ctx, hipHostMalloc((void **)&rs->h_qc, RenderState::MAX_LANES * RenderState::QC_RING * QC_COUNT * QC_STRIDE * sizeof(unsigned long long),
                                    hipHostMallocDefault));
    }
    {
        const size_t ray_lds = (size_t)(SRW * rs->pools[0].nshadow - 1) * 256 * sizeof(float);
        if (ray_lds > (28u << 10)) {    // beyond the default 64 KB of LDS per workgroup (gfx950 has 160 KB)
            PTMI_HIP(ctx, hipFuncSetAttribute((const void *)shade_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ray_lds));
            PTMI_HIP(ctx, hipFuncSetAttribute((const void *)shade_diff_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ray_lds));
        }
    }
    // Which list a class of queue entries goes to (bin_kernel).  The lean kernels take what they are written for: the
    // returning calls when scatterRays == 1 (a frame that waits for its only ray finishes when it returns), plain
    // diffuse hits -- those with no bounces left only when the scene has a sky, whose sample guarantees a shadow ray to
    // wait for (with no ray at all the path goes straight on, which is the general kernel's business).
    // PTMI_SHADE_SPLIT=0 sends everything to the general kernel (A/B, and the parity tests' bit-for-bit cross-check).
    BinMap bm{};
    {
        const bool split = ctx->shade_split != 0;
        const bool lean_ret = split && prm->scatter_rays == 1;
        for (uint8_t &l : bm.list) l = L_G_OTHER;
        bm.list[C_RET] = lean_ret ? L_RET : L_G_RET;
        bm.list[C_RET_ANC] = L_G_RET;
        bm.list[C_DIFF] = split ? L_DIFF : L_G_DIFF;
        bm.list[C_DIFF_LEAF] = split && rs->sky.kind != PTMI_SKY_NONE ? L_DIFF : L_G_DIFF;
        bm.list[C_GLASS] = L_G_GLASS; bm.list[C_REFL] = L_G_REFL; bm.list[C_MID] = L_G_MID; bm.list[C_EMIT] = L_G_EMIT;
        bm.list[C_OTHER] = L_G_OTHER;
    }
    DevScene sc{};
    sc.uvs = rs->d_uvs; sc.tex = rs->d_tex; sc.tex_pixels = rs->d_tex_pixels; sc.ntex = rs->ntex;
    sc.stri = rs->d_stri; sc.tri_class = rs->d_tri_class; sc.ntris = (uint32_t)(ctx->tri_idx.size() / 3); sc.nmats = rs->nmats;
    sc.mats = rs->d_mats; sc.emissive = rs->d_emissive; sc.nemissive = rs->nemissive;
    sc.lights = rs->d_lights; sc.nlights = rs->nlights; sc.sky = rs->sky; sc.sky_img = rs->d_sky_img;
    sc.bh = rs->bh; sc.has_bh = rs->has_bh;

    for (uint32_t done = 0; done < spp_count; done += spp_chunk) {
        const uint32_t chunk = std::min(spp_chunk, spp_count - done);
        RenderArgs ra{};
        ra.cam = *cam; ra.p = *prm; ra.x0 = x0; ra.y0 = y0; ra.w = w; ra.h = h;
        ra.row_block = row_block; ra.row_period = row_period;
        ra.tile_first = d_tile_first; ra.tile_rect = d_tile_rect; ra.ntiles = ntiles;
        ra.cams = d_cams; ra.frame_h = frame_h;
        ra.spp_begin = spp_begin + done; ra.spp_chunk = chunk; ra.total = npix * chunk;
        ra.implicit_fresh = (!rs->has_bh && prm->max_steps == 1 && ctx->implicit_fresh) ? 1u : 0u;

        // The wavefront loop.  The device sizes every iteration itself (iter_prep_kernel), so the host enqueues
        // iterations blind, a batch at a time, and looks at the queue counters once per batch -- one batch behind what
        // it has enqueued, so the GPU never waits for the host.  The batches enqueued after the pool drained are empty
        // iterations (every kernel leaves at once): at most 2 * iters_per_batch of them.
        const size_t qc_words = (size_t)QC_COUNT * QC_STRIDE;
        const int iters_per_batch = std::max(1, ctx->render_batch_iters);
        struct Lane {
            Pool pl; RenderArgs ra; hipStream_t st; uint64_t total; uint64_t bound = ~0ull; int cur = 0; bool active = true;
            unsigned long long *hc; hipEvent_t *ev;
        } L[RenderState::MAX_LANES];
        ra.nlanes = (uint32_t)lanes;
        if (lanes > 1) {            // the other lanes' streams join the caller's stream here and hand back at the end of the pass
            PTMI_HIP(ctx, hipEventRecord(rs->ev_fork, s));
            for (int l = 1; l < lanes; ++l) PTMI_HIP(ctx, hipStreamWaitEvent(rs->lane_stream[l], rs->ev_fork, 0));
        }
        for (int l = 0; l < lanes; ++l) {
            Lane &ln = L[l];
            ln.pl = rs->pools[l];   // a larger pool than asked for may be in place: its queues are laid out for ITS capacity
            ln.pl.out = rs->d_out;
            ln.ra = ra; ln.ra.lane = (uint32_t)l;
            ln.st = l == 0 ? s : rs->lane_stream[l];
            ln.total = lane_sample_count(ra.total, (uint32_t)l, (uint32_t)lanes);
            ln.hc = rs->h_qc + (size_t)l * RenderState::QC_RING * qc_words;
            ln.ev = rs->ev_qc[l];
            ln.active = ln.total > 0;
            if (!ln.active) continue;
            const uint32_t n0 = (uint32_t)std::min<uint64_t>(ln.pl.cap, ln.total);
            timed(ctx, PROF_PREPARE, ln.st, [&] { hipLaunchKernelGGL(iter_init_kernel, dim3((n0 + 255) / 256), dim3(256), 0, ln.st, ln.pl, n0); });
            PTMI_HIP(ctx, hipGetLastError());
        }
        auto enqueue_batch = [&](Lane &ln, int slot) -> int {
            for (int k = 0; k < iters_per_batch; ++k) {
                int r = enqueue_iteration(ctx, ln.pl, sc, ln.ra, bm, ln.total, ln.bound, ln.cur, ln.st);
                if (r) return r;
                ln.cur ^= 1;
            }
            // the counters as the batch's last shade left them: queue lengths of buffer `cur`, samples handed out
            PTMI_HIP(ctx, hipMemcpyAsync(ln.hc + (size_t)slot * qc_words, ln.pl.qcnt, qc_words * sizeof(unsigned long long),
                                         hipMemcpyDeviceToHost, ln.st));
            PTMI_HIP(ctx, hipEventRecord(ln.ev[slot], ln.st));
            return PTMI_OK;
        };
        for (int l = 0; l < lanes; ++l)
            if (L[l].active && (rc = enqueue_batch(L[l], 0))) return rc;
        auto any_active = [&]() { for (int l = 0; l < lanes; ++l) if (L[l].active) return true; return false; };
        for (uint64_t batch = 0; any_active(); ++batch) {
            if (batch > 100000000ull) return set_err(ctx, PTMI_ERR_HIP, "ptmi_render: wavefront loop did not terminate");
            const int slot = (int)(batch % RenderState::QC_RING), ahead = (int)((batch + 1) % RenderState::QC_RING);
            int cur_then[RenderState::MAX_LANES];           // the buffer batch `batch` left each lane's queues in
            for (int l = 0; l < lanes; ++l) {
                cur_then[l] = L[l].cur;
                if (L[l].active && (rc = enqueue_batch(L[l], ahead))) return rc;
            }
            for (int l = 0; l < lanes; ++l) {
                Lane &ln = L[l];
                if (!ln.active) continue;
                PTMI_HIP(ctx, hipEventSynchronize(ln.ev[slot]));
                const unsigned long long *hc = ln.hc + (size_t)slot * qc_words;
                const unsigned long long left = hc[(QC_EXT0 + cur_then[l]) * QC_STRIDE] + hc[(QC_SHP0 + cur_then[l]) * QC_STRIDE] +
                                                hc[(QC_SHA0 + cur_then[l]) * QC_STRIDE];
                const unsigned long long started = std::min<unsigned long long>(hc[QC_NEXT * QC_STRIDE], ln.total);
                if (left == 0 && started >= ln.total) ln.active = false;
                // every sample is either not started, alive in one of the three queues, or done: an upper bound of what
                // any later iteration can hold
                if (ctx->render_shrink_grids) ln.bound = std::min<uint64_t>(ln.bound, left + (ln.total - started));
            }
        }
        for (int l = 1; l < lanes; ++l) {
            PTMI_HIP(ctx, hipEventRecord(rs->ev_join[l], rs->lane_stream[l]));
            PTMI_HIP(ctx, hipStreamWaitEvent(s, rs->ev_join[l], 0));
        }
        {   // integrator self-checks (ERR_*): any bit set means a path record was corrupt
            unsigned long long hcnt[CNT_WORDS];
            int rcc = ptmi::read_counters(ctx, hcnt, s);
            if (rcc) return rcc;
            const unsigned long long e = hcnt[CNT_ERROR];
            if (e) {
                for (int sh = 0; sh < CNT_SHARDS; ++sh)
                    (void)hipMemsetAsync(ctx->d_counters + (size_t)sh * CNT_WORDS + CNT_ERROR, 0, sizeof e, s);
                return set_err(ctx, PTMI_ERR_HIP, "ptmi_render: integrator self-check failed, code " + std::to_string(e));
            }
        }
        if (out_samples_host) {
            PTMI_HIP(ctx, hipMemcpyAsync(out_samples_host, rs->d_out, ra.total * 12, hipMemcpyDeviceToHost, s));
            PTMI_HIP(ctx, hipStreamSynchronize(s));
        } else {
            timed(ctx, PROF_ACCUMULATE, s, [&] {
                if (d_lum)
                    hipLaunchKernelGGL(accumulate_kernel<true>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, rs->d_out,
                                       (uint32_t)npix, chunk, d_accum, d_counts, ra, rows_mode ? prm->width : 0, d_lum);
                else
                    hipLaunchKernelGGL(accumulate_kernel<false>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, rs->d_out,
                                       (uint32_t)npix, chunk, d_accum, d_counts, ra, rows_mode ? prm->width : 0, (float *)nullptr); });
            PTMI_HIP(ctx, hipGetLastError());
        }
        if (ctx->profiling) {
            PTMI_HIP(ctx, hipStreamSynchronize(s));
            prof_collect(ctx);
        }
    }
    return PTMI_OK;
}

extern "C" int ptmi_render_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0,
                               int32_t w, int32_t h, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                               uint32_t *counts, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render_dev: no BVH");
    if (!accum_rgb || !counts) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_dev: null buffer");
    int rc = render_core(ctx, camera, params, x0, y0, w, h, spp_begin, spp_count, accum_rgb, counts, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    PTMI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return PTMI_OK;
}

extern "C" int ptmi_render_rows_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t row_block,
                                    int32_t shard, int32_t nshards, uint32_t spp_begin, uint32_t spp_count,
                                    float *accum_rgb, uint32_t *counts, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_rows_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render_rows_dev: no BVH");
    if (!accum_rgb || !counts || !params) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_rows_dev: null argument");
    if (row_block < 1 || nshards < 1 || shard < 0 || shard >= nshards)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_rows_dev: need row_block >= 1 and 0 <= shard < nshards");
    // rows of this shard: blocks shard, shard + nshards, ... of row_block rows each (the last block may be short)
    int64_t rows = 0;
    for (int64_t b = shard; b * row_block < params->height; b += nshards)
        rows += std::min<int64_t>(row_block, params->height - b * row_block);
    if (rows == 0) return PTMI_OK;
    if ((int64_t)row_block * nshards > 0x7FFFFFFF) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_rows_dev: row_block too large");
    int rc = render_core(ctx, camera, params, 0, shard * row_block, params->width, (int)rows, spp_begin, spp_count, accum_rgb,
                         counts, nullptr, (hipStream_t)stream, row_block, row_block * nshards);
    if (rc) return rc;
    PTMI_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return PTMI_OK;
}

static int render_tiles(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t ntiles, const int32_t *tiles_xywh,
                        uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts, float *d_lum, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render_tiles_dev: no BVH");
    RenderState *rs = ctx->render;
    if (!rs || !rs->have_scene) return set_err(ctx, PTMI_ERR_NO_SCENE, "ptmi_render_tiles_dev: no scene uploaded");
    if (!accum_rgb || !counts || !params || (ntiles && !tiles_xywh))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: null argument");
    if (ntiles == 0) return PTMI_OK;
    std::vector<uint32_t> first(ntiles + 1);
    uint64_t total = 0;
    for (uint32_t k = 0; k < ntiles; ++k) {
        const int32_t *t = tiles_xywh + 4 * (size_t)k;
        if (t[2] <= 0 || t[3] <= 0 || t[0] < 0 || t[1] < 0 || (int64_t)t[0] + t[2] > params->width || (int64_t)t[1] + t[3] > params->height)
            return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: tile " + std::to_string(k) + " is empty or outside the image");
        first[k] = (uint32_t)total;
        total += (uint64_t)t[2] * (uint64_t)t[3];
        if (total > (uint64_t)params->width * (uint64_t)params->height)
            return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: the tiles cover more pixels than the image has (overlap)");
    }
    first[ntiles] = (uint32_t)total;
    {   // disjointness (accumulate_kernel's read-modify-write of a pixel is only race-free when one thread owns it): a
        // bitmap of the image, one bit per pixel, filled tile by tile -- linear in the pixels covered, whatever the
        // number of tiles (an adaptive pass hands down hundreds of thousands of 1-row runs)
        const size_t words_per_row = ((size_t)params->width + 63) / 64;
        std::vector<uint64_t> covered(words_per_row * (size_t)params->height, 0);
        for (uint32_t k = 0; k < ntiles; ++k) {
            const int32_t *t = tiles_xywh + 4 * (size_t)k;
            const size_t w0 = (size_t)t[0] / 64, w1 = (size_t)(t[0] + t[2] - 1) / 64;
            for (int32_t y = t[1]; y < t[1] + t[3]; ++y) {
                uint64_t *row = covered.data() + (size_t)y * words_per_row;
                for (size_t wd = w0; wd <= w1; ++wd) {
                    const int lo = wd == w0 ? t[0] % 64 : 0, hi = wd == w1 ? (t[0] + t[2] - 1) % 64 : 63;   // bits lo..hi of this word
                    const uint64_t m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
                    if (row[wd] & m)
                        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: tile " + std::to_string(k) + " overlaps an earlier tile in row " +
                                                              std::to_string(y));
                    row[wd] |= m;
                }
            }
        }
    }
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b_first = ((size_t)(ntiles + 1) * 4 + 15) / 16 * 16, b_rect = (size_t)ntiles * 16;
    if (rs->tiles_bytes < b_first + b_rect) {
        if (rs->d_tiles) { (void)hipFree(rs->d_tiles); rs->d_tiles = nullptr; rs->tiles_bytes = 0; }
        PTMI_HIP(ctx, hipMalloc((void **)&rs->d_tiles, b_first + b_rect));
        rs->tiles_bytes = b_first + b_rect;
    }
    hipStream_t s = (hipStream_t)stream;
    PTMI_HIP(ctx, hipMemcpyAsync(rs->d_tiles, first.data(), (size_t)(ntiles + 1) * 4, hipMemcpyHostToDevice, s));
    PTMI_HIP(ctx, hipMemcpyAsync(rs->d_tiles + b_first, tiles_xywh, b_rect, hipMemcpyHostToDevice, s));
    PTMI_HIP(ctx, hipStreamSynchronize(s));                 // `first` is a local
    // the region is handed to render_core as a 1-row strip of `total` pixels; region_pixel() maps them to the tiles
    if (total > 0x7FFFFFFFull) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_dev: too many pixels");
    int rc = render_core(ctx, camera, params, 0, 0, (int)total, 1, spp_begin, spp_count, accum_rgb, counts, nullptr, s, 0, 0,
                         (const uint32_t *)rs->d_tiles, (const int4 *)(rs->d_tiles + b_first), ntiles, nullptr, 0, d_lum);
    if (rc) return rc;
    PTMI_HIP(ctx, hipStreamSynchronize(s));
    return PTMI_OK;
}
extern "C" int ptmi_render_tiles_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t ntiles,
                                     const int32_t *tiles_xywh, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                                     uint32_t *counts, void *stream) {
    return render_tiles(ctx, camera, params, ntiles, tiles_xywh, spp_begin, spp_count, accum_rgb, counts, nullptr, stream);
}
extern "C" int ptmi_render_tiles_stats_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t ntiles,
                                           const int32_t *tiles_xywh, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                                           uint32_t *counts, float *pixel_stats, void *stream) {
    if (ctx && !pixel_stats) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_tiles_stats_dev: pixel_stats is null");
    return render_tiles(ctx, camera, params, ntiles, tiles_xywh, spp_begin, spp_count, accum_rgb, counts, pixel_stats, stream);
}

// The reference's adaptive worker loop (main.go:837-897 around Tile.GetNoisiestPixel, tile.go:34-66) as TWO passes over
// the frame -- see ptmi.h.
extern "C" int ptmi_render_adaptive_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t samples_per_pixel,
                                        uint32_t max_samples, float *accum_rgb, uint32_t *counts, float *pixel_stats, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_adaptive_dev: ctx is null");
    if (!params || !accum_rgb || !counts || !pixel_stats) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_adaptive_dev: null argument");
    if (samples_per_pixel < 1 || max_samples < 1 || max_samples > (1u << 24) || params->width <= 0 || params->height <= 0)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_adaptive_dev: need samples_per_pixel >= 1, 1 <= max_samples <= 2^24, a non-empty image");
    // SampleCount after each pick of a pixel that never converges: a pick adds samples_per_pixel samples but stops once
    // SampleCount > max_samples (main.go:861-883).  `tested` = the first of those counts above 32 that is not the last:
    // the count at which the contrast test (tile.go:51) first decides anything; `cap` = where such a pixel stops.
    uint32_t cap = 0, tested = 0;
    while (cap < max_samples) {
        if (cap > 32 && !tested) tested = cap;
        cap += std::min(samples_per_pixel, max_samples + 1 - cap);
    }
    const int32_t whole[4] = {0, 0, params->width, params->height};
    int rc = render_tiles(ctx, camera, params, 1, whole, 0, tested ? tested : cap, accum_rgb, counts, pixel_stats, stream);
    if (rc || !tested) return rc;
    RenderState *rs = ctx->render;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)params->width * (size_t)params->height;
    if (!rs || !rs->d_out || rs->out_bytes < npix) return set_err(ctx, PTMI_ERR_HIP, "ptmi_render_adaptive_dev: no scratch buffer after the first pass");
    // who goes on: the mask is computed on the device (into the idle per-sample output buffer) and read back; the runs of
    // pixels that go on become 1-row tiles.  A pixel's contrast only grows with more samples, so one look is enough.
    uint8_t *d_mask = (uint8_t *)rs->d_out;
    hipLaunchKernelGGL(adaptive_mask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, pixel_stats, (uint32_t)npix, d_mask);
    PTMI_HIP(ctx, hipGetLastError());
    std::vector<uint8_t> mask(npix);
    PTMI_HIP(ctx, hipMemcpyAsync(mask.data(), d_mask, npix, hipMemcpyDeviceToHost, s));
    PTMI_HIP(ctx, hipStreamSynchronize(s));
    std::vector<int32_t> runs;
    for (int32_t y = 0; y < params->height; ++y) {
        const uint8_t *row = mask.data() + (size_t)y * params->width;
        for (int32_t x = 0; x < params->width;) {
            if (!row[x]) { ++x; continue; }
            const int32_t x0 = x;
            while (x < params->width && row[x]) ++x;
            runs.insert(runs.end(), {x0, y, x - x0, 1});
        }
    }
    if (runs.empty()) return PTMI_OK;
    return render_tiles(ctx, camera, params, (uint32_t)(runs.size() / 4), runs.data(), tested, cap - tested, accum_rgb, counts, pixel_stats, stream);
}

extern "C" int ptmi_render_frames_dev(ptmi_ctx *ctx, const PtCamera *cameras, uint32_t nframes, const PtParams *params, uint32_t spp_begin,
                                      uint32_t spp_count, float *accum_rgb, uint32_t *counts, void *stream) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_frames_dev: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render_frames_dev: no BVH");
    RenderState *rs = ctx->render;
    if (!rs || !rs->have_scene) return set_err(ctx, PTMI_ERR_NO_SCENE, "ptmi_render_frames_dev: no scene uploaded");
    if (!accum_rgb || !counts || !params || (nframes && !cameras)) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_frames_dev: null argument");
    if (nframes == 0) return PTMI_OK;
    if (params->width <= 0 || params->height <= 0) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_frames_dev: empty image");
    if ((uint64_t)params->height * nframes > 0x7FFFFFFFull || (uint64_t)params->width * params->height * nframes * 3 >= 0xFFFFFFFFull)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_frames_dev: too many pixels in one call (width x height x frames x 3 < 2^32)");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    if (rs->cams_n < nframes) {
        if (rs->d_cams) { (void)hipFree(rs->d_cams); rs->d_cams = nullptr; rs->cams_n = 0; }
        PTMI_HIP(ctx, hipMalloc((void **)&rs->d_cams, (size_t)nframes * sizeof(PtCamera)));
        rs->cams_n = nframes;
    }
    hipStream_t s = (hipStream_t)stream;
    PTMI_HIP(ctx, hipMemcpyAsync(rs->d_cams, cameras, (size_t)nframes * sizeof(PtCamera), hipMemcpyHostToDevice, s));
    PTMI_HIP(ctx, hipStreamSynchronize(s));                 // the caller's array is borrowed for the call only
    // the frames are handed to render_core as one region of nframes x height rows; fresh_head() maps a row to its frame
    int rc = render_core(ctx, cameras, params, 0, 0, params->width, (int)(params->height * nframes), spp_begin, spp_count, accum_rgb, counts,
                         nullptr, s, 0, 0, nullptr, nullptr, 0, rs->d_cams, params->height);
    if (rc) return rc;
    PTMI_HIP(ctx, hipStreamSynchronize(s));
    return PTMI_OK;
}

extern "C" int ptmi_render(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0, int32_t w,
                           int32_t h, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render: no BVH");
    if (!accum_rgb || !counts) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: null buffer");
    RenderState *rs = ctx->render;
    if (!rs || !rs->have_scene) return set_err(ctx, PTMI_ERR_NO_SCENE, "ptmi_render: no scene uploaded");
    if (w <= 0 || h <= 0) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render: empty rectangle");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)w * h;
    if (rs->accum_pix < npix) {
        if (rs->d_accum) (void)hipFree(rs->d_accum);
        if (rs->d_counts) (void)hipFree(rs->d_counts);
        rs->d_accum = nullptr; rs->d_counts = nullptr; rs->accum_pix = 0;
        PTMI_HIP(ctx, hipMalloc((void **)&rs->d_accum, npix * 12));
        PTMI_HIP(ctx, hipMalloc((void **)&rs->d_counts, npix * 4));
        rs->accum_pix = npix;
    }
    hipStream_t s = ctx->stream;
    PTMI_HIP(ctx, hipMemcpyAsync(rs->d_accum, accum_rgb, npix * 12, hipMemcpyHostToDevice, s));
    PTMI_HIP(ctx, hipMemcpyAsync(rs->d_counts, counts, npix * 4, hipMemcpyHostToDevice, s));
    int rc = render_core(ctx, camera, params, x0, y0, w, h, spp_begin, spp_count, rs->d_accum, rs->d_counts, nullptr, s);
    if (rc) return rc;
    PTMI_HIP(ctx, hipMemcpyAsync(accum_rgb, rs->d_accum, npix * 12, hipMemcpyDeviceToHost, s));
    PTMI_HIP(ctx, hipMemcpyAsync(counts, rs->d_counts, npix * 4, hipMemcpyDeviceToHost, s));
    PTMI_HIP(ctx, hipStreamSynchronize(s));
    return PTMI_OK;
}

extern "C" int ptmi_render_samples(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0,
                                   int32_t w, int32_t h, uint32_t spp_begin, uint32_t spp_count, float *out_rgb) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_render_samples: ctx is null");
    if (!ctx->have_bvh) return set_err(ctx, PTMI_ERR_NO_BVH, "ptmi_render_samples: no BVH");
    if (!out_rgb) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_samples: null buffer");
    if (w <= 0 || h <= 0) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_samples: empty rectangle");
    if ((uint64_t)w * h * spp_count > (256ull << 20))
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_samples: more than 2^28 samples in one call");
    return render_core(ctx, camera, params, x0, y0, w, h, spp_begin, spp_count, nullptr, nullptr, out_rgb, ctx->stream);
}

// vec3.go:95-106 + main.go:886-896
extern "C" int ptmi_to_rgba(ptmi_ctx *ctx, const float *accum_rgb, const uint32_t *counts, int32_t width, int32_t height,
                            uint8_t *rgba) {
    if (!accum_rgb || !counts || !rgba || width <= 0 || height <= 0)
        return set_err(ctx, PTMI_ERR_INVALID, "ptmi_to_rgba: bad argument");
    auto gomax = [](float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); };
    auto gomin = [](float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); };
    for (int Y = 0; Y < height; ++Y)
        for (int X = 0; X < width; ++X) {
            const size_t i = (size_t)Y * width + X;
            const int ox = width - X;                           // main.go:896
            if (ox < 0 || ox >= width || counts[i] == 0) continue;
            const float n = (float)counts[i];
            uint8_t *o = rgba + 4 * ((size_t)Y * width + ox);
            for (int k = 0; k < 3; ++k) {
                float c = accum_rgb[3 * i + k] / n;             // main.go:886-888
                float v = gomax(0.0f, gomin(1.0f, std::sqrt(gomax(0.0f, c)))) * 255.0f;
                o[k] = (v == v) ? (uint8_t)v : 0;
            }
            o[3] = 255;
        }
    return PTMI_OK;
}

extern "C" int ptmi_set_pool_limit(ptmi_ctx *ctx, uint64_t max_paths, uint64_t max_bytes) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_set_pool_limit: ctx is null");
    if (max_paths) ctx->pool_paths = (int)std::max<uint64_t>(1u << 16, std::min<uint64_t>(max_paths, 1u << 28));
    if (max_bytes) ctx->pool_bytes_max = std::max<uint64_t>(max_bytes, 64ull << 20);
    RenderState *rs = ctx->render;
    if (rs && rs->pool_mem && (rs->pool_bytes > ctx->pool_bytes_max || (uint64_t)rs->pool_cap * rs->pool_lanes > (uint64_t)ctx->pool_paths)) {
        PTMI_HIP(ctx, hipSetDevice(ctx->device));      // the pool in place is over the new limit: give it back now
        PTMI_HIP(ctx, hipDeviceSynchronize());
        (void)hipFree(rs->pool_mem);
        rs->pool_mem = nullptr; rs->pool_bytes = 0; rs->pool_cap = 0;
    }
    return PTMI_OK;
}

extern "C" int ptmi_pool_info(ptmi_ctx *ctx, uint64_t *paths, uint64_t *bytes) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_pool_info: ctx is null");
    const RenderState *rs = ctx->render;
    if (paths) *paths = rs && rs->pool_mem ? (uint64_t)rs->pool_cap * rs->pool_lanes : 0;
    if (bytes) *bytes = rs && rs->pool_mem ? (uint64_t)rs->pool_bytes : 0;
    return PTMI_OK;
}

// ---- ptmi_selftest_math: sqrt_mid / rcp_mid (shade_device.hpp) against sqrtf / the division on every float of their ranges
namespace {
__global__ __launch_bounds__(256) void selftest_math_kernel(uint32_t sq_first, uint32_t sq_count, uint32_t rc_first, uint32_t rc_count,
                                                            unsigned long long *bad) {
    uint32_t wrong = 0;
    const uint32_t stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = t0; k < sq_count; k += stride) {
        const float x = u2f(sq_first + k);
        wrong += f2u(sqrt_mid(x)) != f2u(sqrtf(x));
    }
    for (uint32_t k = t0; k < rc_count; k += stride) {
        const float x = u2f(rc_first + k);
        wrong += f2u(rcp_mid(x)) != f2u(1.0f / x);
    }
    for (int off = 32; off > 0; off >>= 1) wrong += __shfl_xor(wrong, off);
    if ((threadIdx.x & 63) == 0 && wrong) atomicAdd(bad, (unsigned long long)wrong);
}
}  // namespace

extern "C" int ptmi_selftest_math(ptmi_ctx *ctx, uint64_t *mismatches) {
    if (!ctx || !mismatches) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_selftest_math: null argument");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d_bad = nullptr, h_bad = 0;
    PTMI_HIP(ctx, hipMalloc((void **)&d_bad, sizeof(*d_bad)));
    int rc = PTMI_OK;
    do {
        if (hipMemset(d_bad, 0, sizeof(*d_bad)) != hipSuccess) { rc = PTMI_ERR_HIP; break; }
        // the square root on (MID_LO, MID_HI) and a float either side; the reciprocal on the cubes' range, (2^-60, 2^60)
        float sq_lo = MID_LO, sq_hi = MID_HI, rc_lo = 0x1p-60f, rc_hi = 0x1p60f;
        uint32_t a, b, c, d;
        memcpy(&a, &sq_lo, 4); memcpy(&b, &sq_hi, 4); memcpy(&c, &rc_lo, 4); memcpy(&d, &rc_hi, 4);
        hipLaunchKernelGGL(selftest_math_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, a, b - a + 1, c, d - c + 1, d_bad);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(&h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost) != hipSuccess) { rc = PTMI_ERR_HIP; break; }
    } while (false);
    (void)hipFree(d_bad);
    if (rc) return set_err(ctx, rc, "ptmi_selftest_math: HIP call failed");
    *mismatches = (uint64_t)h_bad;
    return PTMI_OK;
}

extern "C" int ptmi_set_profiling(ptmi_ctx *ctx, int enabled) {
    if (!ctx) return set_err(nullptr, PTMI_ERR_INVALID, "ptmi_set_profiling: ctx is null");
    PTMI_HIP(ctx, hipSetDevice(ctx->device));
    PTMI_HIP(ctx, hipDeviceSynchronize());
    prof_collect(ctx);
    ctx->profiling = enabled != 0;
    for (double &v : ctx->prof_ms) v = 0.0;
    ctx->prof_iterations = ctx->prof_launches = 0;
    return PTMI_OK;
}

extern "C" int ptmi_render_profile(ptmi_ctx *ctx, PtRenderProfile *out) {
    if (!ctx || !out) return set_err(ctx, PTMI_ERR_INVALID, "ptmi_render_profile: null argument");
    out->prepare_ms = ctx->prof_ms[PROF_PREPARE]; out->generate_ms = ctx->prof_ms[PROF_GENERATE];
    out->extend_ms = ctx->prof_ms[PROF_EXTEND]; out->connect_ms = ctx->prof_ms[PROF_CONNECT];
    out->shade_ms = ctx->prof_ms[PROF_SHADE]; out->accumulate_ms = ctx->prof_ms[PROF_ACCUMULATE];
    out->iterations = ctx->prof_iterations; out->launches = ctx->prof_launches;
    return PTMI_OK;
}
