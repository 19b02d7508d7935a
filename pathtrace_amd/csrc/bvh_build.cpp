// bvh_build.cpp -- bit-identical binned restatement of the reference's SAH-like builder.
//
// Reference algorithm (bvh.go:148-246): for every node, for axis in {x,y,z}, for i in 0..128,
//   mid = l + (r-l)*float32(i)/128; partition the node's triangles by centroid[axis] < mid into
//   two FauxBoxes (bounds by Go min/max, counts); cost = 1/8 + nL*areaL + nR*areaR; keep the
//   strictly smallest cost (axis-major, i-minor); leaf if bestCost >= the parent's bestCost.
// That is 387 full passes over the node's triangles.  Here each axis takes ONE pass:
//   * the 129 mids are non-decreasing in i (checked; literal fallback otherwise), so a triangle
//     goes left for every i >= k where k = first i with centroid < mid_i  ("bin" k in 0..129);
//   * FauxBox.Grow is a pure min/max fold, so the left box of candidate i is the merge of bins
//     0..i and the right box the merge of bins i+1..129 -- the same bits in any merge order;
//   * a candidate whose bin is empty has the same partition as its predecessor, hence the same
//     cost, and can never win under the strict '<' -- it is skipped.
// Costs, bounds and tie-breaks therefore match the reference bit for bit (tests/test_bvh_build.py
// checks this against the literal CPU oracle).
//
//
// Parallel form (same bits, any thread count): nodes with many triangles are evaluated
// and partitioned by all threads -- per-thread bin boxes merged afterwards (min/max folds and counts
// do not depend on the order), a stable parallel partition -- and every subtree below PAR_NODE
// triangles is an independent task that builds into its own arrays, spliced back in pre-order.
//
// Build flags: -ffp-contract=off (no FMA fusion; Go/amd64 GOAMD64=v1 never fuses).
#include "bvh_build.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

namespace ptmi {
namespace {

constexpr float MAXF = std::numeric_limits<float>::max();   // math.MaxFloat32

// Go builtin min/max for floats: NaN-propagating, -0 < +0.
inline float gomin(float a, float b) {
    if (a != a || b != b) return std::numeric_limits<float>::quiet_NaN();
    if (a == 0.0f && b == 0.0f) return (std::signbit(a) || std::signbit(b)) ? -0.0f : 0.0f;
    return a < b ? a : b;
}
inline float gomax(float a, float b) {
    if (a != a || b != b) return std::numeric_limits<float>::quiet_NaN();
    if (a == 0.0f && b == 0.0f) return (std::signbit(a) && std::signbit(b)) ? -0.0f : 0.0f;
    return a > b ? a : b;
}

struct TriInfo {       // the fields of BVHTriangle (bvh.go:8-17) the builder reads
    float cen[3];
    float mn[3], mx[3];
};

struct FBox {          // FauxBox (bvh.go:38-44); lo = {X1,Y1,Z1}, hi = {X2,Y2,Z2}
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int64_t count = 0;
};

inline void grow(FBox &b, const TriInfo &t) {          // bvh.go:46-63
    if (b.count == 0) {
        for (int a = 0; a < 3; ++a) { b.lo[a] = t.mn[a]; b.hi[a] = t.mx[a]; }
    } else {
        for (int a = 0; a < 3; ++a) { b.lo[a] = gomin(b.lo[a], t.mn[a]); b.hi[a] = gomax(b.hi[a], t.mx[a]); }
    }
    b.count += 1;
}
inline void merge(FBox &b, const FBox &o) {
    if (o.count == 0) return;
    if (b.count == 0) { b = o; return; }
    for (int a = 0; a < 3; ++a) { b.lo[a] = gomin(b.lo[a], o.lo[a]); b.hi[a] = gomax(b.hi[a], o.hi[a]); }
    b.count += o.count;
}
inline float area(const FBox &b) {                     // bvh.go:64-70
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * ((dx * dy + dy * dz) + dx * dz);
}
inline float split_cost(const FBox &L, const FBox &R) { // bvh.go:201
    return (0.125f + (float)L.count * area(L)) + (float)R.count * area(R);
}

struct Builder {
    const TriInfo *tris = nullptr;      // shared, read-only
    uint32_t *ids = nullptr, *tmp = nullptr;   // shared; a builder only touches its own [begin, begin+n)
    uint8_t kbuf[64];                   // per-triangle bin of the current axis (0..129), small nodes only
    LinearBVH *out = nullptr;
    int threshold = 4;

    void make_leaf(size_t node, size_t begin, size_t n) {
        LinNode &nd = out->nodes[node];
        nd.is_leaf = 1;
        nd.tri_offset = (uint32_t)out->tri_ids.size();
        nd.tri_count = (uint32_t)n;
        out->tri_ids.insert(out->tri_ids.end(), ids + begin, ids + begin + n);
    }

    // Evaluate the 129 candidates of one axis; update best (strict <, in order of i).
    void eval_axis(size_t begin, size_t n, int axis, float l, float r, float &best_cost, int &best_axis,
                   float &best_split) {
        float mids[129];
        bool monotone = true;
        for (int i = 0; i <= 128; ++i) {
            mids[i] = l + ((r - l) * (float)i) / (float)128;     // bvh.go:172
            if (i > 0 && !(mids[i] >= mids[i - 1])) monotone = false;
        }
        if (!monotone || mids[0] != mids[0]) {
            // literal fallback (NaN / overflowing bounds): one scan per candidate
            for (int i = 0; i <= 128; ++i) {
                FBox L, R;
                for (size_t k = 0; k < n; ++k) {
                    const TriInfo &t = tris[ids[begin + k]];
                    if (t.cen[axis] < mids[i]) grow(L, t); else grow(R, t);
                }
                float c = split_cost(L, R);
                if (c < best_cost) { best_cost = c; best_axis = axis; best_split = mids[i]; }
            }
            return;
        }
        // bin of each triangle: first i with centroid < mid_i (129 = never left)
        bool nonempty[130];
        std::memset(nonempty, 0, sizeof nonempty);
        if (n <= 48) {
            // small node: no bin boxes, re-scan the (few) triangles per distinct candidate
            uint8_t *ks = kbuf;
            for (size_t k = 0; k < n; ++k) {
                float c = tris[ids[begin + k]].cen[axis];
                int b = (int)(std::upper_bound(mids, mids + 129, c) - mids);
                ks[k] = (uint8_t)b;
                nonempty[b] = true;
            }
            for (int i = 0; i <= 128; ++i) {
                if (i > 0 && !nonempty[i]) continue;
                FBox L, R;
                for (size_t k = 0; k < n; ++k) {
                    const TriInfo &t = tris[ids[begin + k]];
                    if ((int)ks[k] <= i) grow(L, t); else grow(R, t);
                }
                float c = split_cost(L, R);
                if (c < best_cost) { best_cost = c; best_axis = axis; best_split = mids[i]; }
            }
            return;
        }
        FBox bins[130];
        for (size_t k = 0; k < n; ++k) {
            const TriInfo &t = tris[ids[begin + k]];
            int b = (int)(std::upper_bound(mids, mids + 129, t.cen[axis]) - mids);
            grow(bins[b], t);
        }
        FBox suffix[130];            // suffix[i] = merge of bins i..129
        suffix[129] = bins[129];
        for (int i = 128; i >= 0; --i) { suffix[i] = bins[i]; merge(suffix[i], suffix[i + 1]); }
        FBox L;
        for (int i = 0; i <= 128; ++i) {
            merge(L, bins[i]);
            if (i > 0 && bins[i].count == 0) continue;
            float c = split_cost(L, suffix[i + 1]);
            if (c < best_cost) { best_cost = c; best_axis = axis; best_split = mids[i]; }
        }
    }

    void build(size_t begin, size_t n, const float lo[3], const float hi[3], int depth, float parent_cost) {
        size_t node = out->nodes.size();
        LinNode nd;
        std::memset(&nd, 0, sizeof nd);
        for (int a = 0; a < 3; ++a) { nd.minb[a] = lo[a]; nd.maxb[a] = hi[a]; }
        out->nodes.push_back(nd);

        if ((int64_t)n <= (int64_t)threshold || depth <= 0) { make_leaf(node, begin, n); return; }   // bvh.go:153

        float best_cost = MAXF, best_split = -1.0f;
        int best_axis = -1;
        for (int axis = 0; axis < 3; ++axis) eval_axis(begin, n, axis, lo[axis], hi[axis], best_cost, best_axis, best_split);

        if (best_cost >= parent_cost) { make_leaf(node, begin, n); return; }                          // bvh.go:210

        // getSplit (bvh.go:231-246): stable partition, Box.Grow bounds
        FBox L, R;
        size_t nl = 0, nr = 0;
        for (size_t k = 0; k < n; ++k) {
            uint32_t id = ids[begin + k];
            const TriInfo &t = tris[id];
            if (t.cen[best_axis] < best_split) { ids[begin + nl++] = id; grow(L, t); }
            else { tmp[begin + nr++] = id; grow(R, t); }
        }
        std::copy(tmp + begin, tmp + begin + nr, ids + begin + nl);

        build(begin, nl, L.lo, L.hi, depth - 1, best_cost);
        out->nodes[node].second_child = (uint32_t)out->nodes.size();                                  // linear_bvh.go:102
        build(begin + nl, nr, R.lo, R.hi, depth - 1, best_cost);
    }
};

constexpr size_t PAR_NODE = 1u << 17;     // nodes with at least this many triangles are split by all threads
constexpr size_t TASK_NODE = 1u << 14;    // subtrees below this many triangles are independent tasks

// Run fn(k0, k1, thread) over [0, n) cut into `nthreads` contiguous chunks (chunk t = thread t, so results
// that are merged in thread order are merged in triangle order).
template <typename Fn>
void parallel_for(int nthreads, size_t n, Fn fn) {
    if (nthreads <= 1 || n < 4096) { fn((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        const size_t k0 = n * (size_t)t / (size_t)nthreads, k1 = n * (size_t)(t + 1) / (size_t)nthreads;
        th.emplace_back([=]() { fn(k0, k1, t); });
    }
    for (std::thread &x : th) x.join();
}

struct ParallelBuilder {
    const TriInfo *tris = nullptr;
    uint32_t *ids = nullptr, *tmp = nullptr;
    int threshold = 4, nthreads = 1;

    struct TopNode {
        float lo[3], hi[3];
        int kind;                  // 0 inner, 1 leaf, 2 task
        int left = -1, right = -1; // inner
        size_t begin = 0, n = 0;   // leaf / task
        int depth = 0; float parent_cost = 0;   // task
        int task = -1;
    };
    std::vector<TopNode> top_nodes;
    std::vector<int> tasks;                    // top node numbers of kind 2
    std::vector<LinearBVH> task_out;

    // One axis of bvh.go:167-207 for a big node: every thread bins its chunk, the bin boxes are merged.
    // Returns false when the mids are not monotone (NaN / overflowing bounds): the caller falls back.
    bool eval_axis_par(int nthreads, size_t begin, size_t n, int axis, float l, float r, float &best_cost, int &best_axis, float &best_split) {
        float mids[129];
        for (int i = 0; i <= 128; ++i) {
            mids[i] = l + ((r - l) * (float)i) / (float)128;     // bvh.go:172
            if (i > 0 && !(mids[i] >= mids[i - 1])) return false;
        }
        if (mids[0] != mids[0]) return false;
        std::vector<FBox> tb((size_t)nthreads * 130);
        parallel_for(nthreads, n, [&](size_t k0, size_t k1, int t) {
            FBox *bins = tb.data() + (size_t)t * 130;
            for (size_t k = k0; k < k1; ++k) {
                const TriInfo &ti = tris[ids[begin + k]];
                int b = (int)(std::upper_bound(mids, mids + 129, ti.cen[axis]) - mids);
                grow(bins[b], ti);
            }
        });
        FBox bins[130];
        for (int t = 0; t < nthreads; ++t)
            for (int b = 0; b < 130; ++b) merge(bins[b], tb[(size_t)t * 130 + b]);
        FBox suffix[130];
        suffix[129] = bins[129];
        for (int i = 128; i >= 0; --i) { suffix[i] = bins[i]; merge(suffix[i], suffix[i + 1]); }
        FBox L;
        for (int i = 0; i <= 128; ++i) {
            merge(L, bins[i]);
            if (i > 0 && bins[i].count == 0) continue;
            float c = split_cost(L, suffix[i + 1]);
            if (c < best_cost) { best_cost = c; best_axis = axis; best_split = mids[i]; }
        }
        return true;
    }

    int add(const TopNode &t) { top_nodes.push_back(t); return (int)top_nodes.size() - 1; }

    int top(size_t begin, size_t n, const float lo[3], const float hi[3], int depth, float parent_cost) {
        TopNode t;
        for (int a = 0; a < 3; ++a) { t.lo[a] = lo[a]; t.hi[a] = hi[a]; }
        t.begin = begin; t.n = n;
        if ((int64_t)n <= (int64_t)threshold || depth <= 0) { t.kind = 1; return add(t); }       // bvh.go:153
        const int nthreads = n >= PAR_NODE ? this->nthreads : 1;     // mid-sized nodes: split inline, children become tasks
        if (n < TASK_NODE) {
            t.kind = 2; t.depth = depth; t.parent_cost = parent_cost; t.task = (int)tasks.size();
            const int id = add(t);
            tasks.push_back(id);
            return id;
        }
        float best_cost = MAXF, best_split = -1.0f;
        int best_axis = -1;
        for (int axis = 0; axis < 3; ++axis) {
            if (!eval_axis_par(nthreads, begin, n, axis, lo[axis], hi[axis], best_cost, best_axis, best_split)) {
                Builder b;                                        // literal fallback of the serial builder
                b.tris = tris; b.ids = ids; b.tmp = tmp; b.threshold = threshold;
                b.eval_axis(begin, n, axis, lo[axis], hi[axis], best_cost, best_axis, best_split);
            }
        }
        if (best_cost >= parent_cost) { t.kind = 1; return add(t); }                                 // bvh.go:210
        // getSplit (bvh.go:231-246): stable partition by all threads
        std::vector<size_t> nl_t(nthreads, 0), nr_t(nthreads, 0);
        std::vector<FBox> lb(nthreads), rb(nthreads);
        parallel_for(nthreads, n, [&](size_t k0, size_t k1, int tt) {
            size_t nl = 0;
            for (size_t k = k0; k < k1; ++k) nl += tris[ids[begin + k]].cen[best_axis] < best_split ? 1 : 0;
            nl_t[tt] = nl; nr_t[tt] = (k1 - k0) - nl;
        });
        size_t nl = 0, nr = 0;
        std::vector<size_t> loff(nthreads), roff(nthreads);
        for (int tt = 0; tt < nthreads; ++tt) { loff[tt] = nl; roff[tt] = nr; nl += nl_t[tt]; nr += nr_t[tt]; }
        parallel_for(nthreads, n, [&](size_t k0, size_t k1, int tt) {
            size_t pl = begin + loff[tt], pr = begin + nl + roff[tt];
            FBox L, R;
            for (size_t k = k0; k < k1; ++k) {
                const uint32_t id = ids[begin + k];
                const TriInfo &ti = tris[id];
                if (ti.cen[best_axis] < best_split) { tmp[pl++] = id; grow(L, ti); }
                else { tmp[pr++] = id; grow(R, ti); }
            }
            lb[tt] = L; rb[tt] = R;
        });
        parallel_for(nthreads, n, [&](size_t k0, size_t k1, int) { std::copy(tmp + begin + k0, tmp + begin + k1, ids + begin + k0); });
        FBox L, R;
        for (int tt = 0; tt < nthreads; ++tt) { merge(L, lb[tt]); merge(R, rb[tt]); }
        t.kind = 0;
        const int id = add(t);
        const int l = top(begin, nl, L.lo, L.hi, depth - 1, best_cost);
        const int r = top(begin + nl, nr, R.lo, R.hi, depth - 1, best_cost);
        top_nodes[id].left = l; top_nodes[id].right = r;
        return id;
    }

    void run_tasks() {
        task_out.resize(tasks.size());
        std::vector<int> order(tasks.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return top_nodes[tasks[a]].n > top_nodes[tasks[b]].n; });
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= order.size()) break;
                const TopNode &t = top_nodes[tasks[order[i]]];
                Builder b;
                b.tris = tris; b.ids = ids; b.tmp = tmp; b.threshold = threshold; b.out = &task_out[order[i]];
                b.out->nodes.reserve(t.n / 2 + 16);
                b.out->tri_ids.reserve(t.n);
                b.build(t.begin, t.n, t.lo, t.hi, t.depth, t.parent_cost);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
        for (std::thread &x : th) x.join();
    }

    // pre-order emission (linear_bvh.go:77-106)
    void emit(int id, LinearBVH &out) {
        const TopNode &t = top_nodes[id];
        if (t.kind == 2) {
            const LinearBVH &sub = task_out[t.task];
            const uint32_t nbase = (uint32_t)out.nodes.size(), tbase = (uint32_t)out.tri_ids.size();
            for (LinNode n : sub.nodes) {
                if (n.is_leaf) n.tri_offset += tbase; else n.second_child += nbase;
                out.nodes.push_back(n);
            }
            out.tri_ids.insert(out.tri_ids.end(), sub.tri_ids.begin(), sub.tri_ids.end());
            return;
        }
        LinNode nd;
        std::memset(&nd, 0, sizeof nd);
        for (int a = 0; a < 3; ++a) { nd.minb[a] = t.lo[a]; nd.maxb[a] = t.hi[a]; }
        const size_t idx = out.nodes.size();
        out.nodes.push_back(nd);
        if (t.kind == 1) {
            LinNode &l = out.nodes[idx];
            l.is_leaf = 1; l.tri_offset = (uint32_t)out.tri_ids.size(); l.tri_count = (uint32_t)t.n;
            out.tri_ids.insert(out.tri_ids.end(), ids + t.begin, ids + t.begin + t.n);
            return;
        }
        emit(t.left, out);
        out.nodes[idx].second_child = (uint32_t)out.nodes.size();                                      // linear_bvh.go:102
        emit(t.right, out);
    }
};

}  // namespace

int build_linear_bvh(const float *verts, uint64_t nverts, const uint32_t *tri_idx, uint64_t ntris,
                     const float bounds[6], int threshold, int depth, LinearBVH &out, const char **msg) {
    if ((!verts && nverts) || (!tri_idx && ntris) || !bounds) { *msg = "null input array"; return -1; }
    if (ntris >= 0x7FFFFFF0ull) { *msg = "too many triangles"; return -1; }
    // The reference's traversal stack has 64 entries (linear_bvh.go:117,203); a deeper tree
    // would make it panic.  depth+1 levels need at most depth+1 entries.
    if (depth > 62) { *msg = "depth > 62 would overflow the 64-entry traversal stack"; return -5; }
    for (uint64_t i = 0; i < 3 * ntris; ++i)
        if (tri_idx[i] >= nverts) { *msg = "triangle index out of range"; return -1; }

    out.nodes.clear();
    out.tri_ids.clear();
    std::vector<TriInfo> tris(ntris);
    std::vector<uint32_t> ids(ntris), tmp(ntris);
    int nthreads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("PTMI_BUILD_THREADS")) nthreads = std::max(1, std::min(64, atoi(e)));
    if (ntris < 2 * PAR_NODE) nthreads = 1;
    parallel_for(nthreads, ntris, [&](size_t k0, size_t k1, int) {
        for (size_t k = k0; k < k1; ++k) {          // bvh.go:250-272
            const float *a = verts + 3ull * tri_idx[3 * k], *bb = verts + 3ull * tri_idx[3 * k + 1],
                        *c = verts + 3ull * tri_idx[3 * k + 2];
            TriInfo &t = tris[k];
            for (int x = 0; x < 3; ++x) {
                t.cen[x] = ((a[x] + bb[x]) + c[x]) * (float)(1.0 / 3.0);
                t.mn[x] = gomin(gomin(a[x], bb[x]), c[x]);
                t.mx[x] = gomax(gomax(a[x], bb[x]), c[x]);
            }
            ids[k] = (uint32_t)k;
        }
    });
    const float lo[3] = {bounds[0], bounds[2], bounds[4]}, hi[3] = {bounds[1], bounds[3], bounds[5]};
    if (nthreads == 1) {
        Builder b;
        b.threshold = threshold; b.out = &out; b.tris = tris.data(); b.ids = ids.data(); b.tmp = tmp.data();
        out.nodes.reserve(ntris / 2 + 16);
        out.tri_ids.reserve(ntris);
        b.build(0, ntris, lo, hi, depth, MAXF);
        return 0;
    }
    ParallelBuilder pb;
    pb.tris = tris.data(); pb.ids = ids.data(); pb.tmp = tmp.data(); pb.threshold = threshold; pb.nthreads = nthreads;
    const int root = pb.top(0, ntris, lo, hi, depth, MAXF);
    pb.run_tasks();
    out.nodes.reserve(ntris / 2 + 16);
    out.tri_ids.reserve(ntris);
    pb.emit(root, out);
    return 0;
}

int validate_linear_bvh(const LinearBVH &bvh, uint64_t ntris, const char **msg) {
    const size_t nn = bvh.nodes.size();
    if (nn == 0) { *msg = "empty node array"; return -1; }
    if (nn >= 0x7FFFFFF0ull || bvh.tri_ids.size() >= 0x7FFFFFF0ull) { *msg = "tree too large"; return -1; }
    for (size_t i = 0; i < nn; ++i) {
        const LinNode &n = bvh.nodes[i];
        if (n.is_leaf) {
            if ((uint64_t)n.tri_offset + n.tri_count > bvh.tri_ids.size()) { *msg = "leaf range out of bounds"; return -1; }
        } else {
            // every inner node must have two children (the builder never makes one-child nodes)
            if (i + 1 >= nn || n.second_child <= i + 1 || n.second_child >= nn) { *msg = "bad child offset"; return -1; }
        }
    }
    for (uint32_t id : bvh.tri_ids)
        if (id >= ntris) { *msg = "triangle id out of range"; return -1; }
    // depth check by iterative walk
    std::vector<std::pair<uint32_t, int>> st;
    st.push_back({0u, 1});
    size_t visited = 0;
    while (!st.empty()) {
        auto [p, d] = st.back();
        st.pop_back();
        if (++visited > nn) { *msg = "node graph is not a tree"; return -1; }
        if (d > 63) { *msg = "tree deeper than the 64-entry traversal stack"; return -5; }
        const LinNode &n = bvh.nodes[p];
        if (!n.is_leaf) { st.push_back({n.second_child, d + 1}); st.push_back({p + 1, d + 1}); }
    }
    return 0;
}

BvhStats linear_bvh_stats(const LinearBVH &bvh) {
    BvhStats s;
    s.max_depth = -1; s.max_tris = -1; s.min_tris = std::numeric_limits<int64_t>::max();
    if (bvh.nodes.empty()) return s;
    std::vector<std::pair<uint32_t, int>> st;
    st.push_back({0u, 1});                       // GetStats(1): main.go:809
    while (!st.empty()) {
        auto [p, d] = st.back();
        st.pop_back();
        const LinNode &n = bvh.nodes[p];
        if (n.is_leaf) {
            s.max_depth = std::max<int64_t>(s.max_depth, d);
            s.min_tris = std::min<int64_t>(s.min_tris, n.tri_count);
            s.max_tris = std::max<int64_t>(s.max_tris, n.tri_count);
            s.total_leafs += 1; s.total_nodes += 1;     // bvh.go:123: a leaf counts as one node, inner nodes as none
            s.total_triangles += n.tri_count;
        } else {
            st.push_back({n.second_child, d + 1});
            st.push_back({p + 1, d + 1});
        }
    }
    return s;
}

int pack_bvh(const LinearBVH &bvh, const float *verts, const uint32_t *tri_idx, PackedBVH &out, const char **msg) {
    const size_t nn = bvh.nodes.size();
    out.nodes.clear(); out.tris.clear();
    out.bounds_finite = true;
    // triangle records in leaf order
    // + one degenerate record (all zero: det = 0 or NaN, never a hit in either triangle test) that empty
    // leaves reference, so the kernels need no empty-leaf case
    out.tris.assign(bvh.tri_ids.size() + 1, PackedTri{});
    out.dummy_tri = (uint32_t)bvh.tri_ids.size();
    for (size_t i = 0; i < nn; ++i) {
        const LinNode &n = bvh.nodes[i];
        if (!n.is_leaf) continue;
        for (uint32_t k = 0; k < n.tri_count; ++k) {
            uint32_t id = bvh.tri_ids[n.tri_offset + k];
            PackedTri &t = out.tris[n.tri_offset + k];
            const float *a = verts + 3ull * tri_idx[3ull * id], *b = verts + 3ull * tri_idx[3ull * id + 1],
                        *c = verts + 3ull * tri_idx[3ull * id + 2];
            t.BCx[0] = b[0]; t.BCx[1] = c[0]; t.BCy[0] = b[1]; t.BCy[1] = c[1]; t.BCz[0] = b[2]; t.BCz[1] = c[2];
            t.Axy[0] = a[0]; t.Axy[1] = a[1]; t.Az = a[2];
            t.prim = id; t.nleft = n.tri_count - 1 - k; t.pad = 0;
        }
    }
    // inner nodes numbered in pre-order
    // inner nodes numbered in pre-order (a first child sits right after its parent, half the time in the
    // same 128-byte line)
    std::vector<uint32_t> inner_index(nn, 0);
    uint32_t ninner = 0;
    for (size_t i = 0; i < nn; ++i) if (!bvh.nodes[i].is_leaf) inner_index[i] = ninner++;
    auto ref_of = [&](size_t i) -> uint32_t {
        const LinNode &n = bvh.nodes[i];
        if (!n.is_leaf) return inner_index[i];
        if (n.tri_count == 0) return REF_LEAF_BIT | out.dummy_tri;
        return REF_LEAF_BIT | n.tri_offset;
    };
    out.nodes.resize(ninner);
    for (size_t i = 0; i < nn; ++i) {
        const LinNode &n = bvh.nodes[i];
        if (n.is_leaf) continue;
        PackedNode &p = out.nodes[inner_index[i]];
        const LinNode &c0 = bvh.nodes[i + 1], &c1 = bvh.nodes[n.second_child];
        float *axis[3] = {p.x, p.y, p.z};
        for (int x = 0; x < 3; ++x) {
            axis[x][0] = c0.minb[x]; axis[x][1] = c0.maxb[x];
            axis[x][2] = c1.minb[x]; axis[x][3] = c1.maxb[x];
            for (int k = 0; k < 4; ++k)
                if (!std::isfinite(axis[x][k])) out.bounds_finite = false;
        }
        p.ref0 = ref_of(i + 1); p.ref1 = ref_of(n.second_child);
        p.pad0 = p.pad1 = 0;
    }
    out.root_ref = ref_of(0);
    out.max_depth = (int)linear_bvh_stats(bvh).max_depth;
    if (out.max_depth > 63) { *msg = "tree deeper than the traversal stack"; return -5; }
    return 0;
}

}  // namespace ptmi
