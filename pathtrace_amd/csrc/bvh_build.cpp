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
// Build flags: -ffp-contract=off (no FMA fusion; Go/amd64 GOAMD64=v1 never fuses).
#include "bvh_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

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
    std::vector<TriInfo> tris;
    std::vector<uint32_t> ids, tmp;
    std::vector<uint8_t> kbuf;          // per-triangle bin of the current axis (0..129)
    LinearBVH *out = nullptr;
    int threshold = 4;

    void make_leaf(size_t node, size_t begin, size_t n) {
        LinNode &nd = out->nodes[node];
        nd.is_leaf = 1;
        nd.tri_offset = (uint32_t)out->tri_ids.size();
        nd.tri_count = (uint32_t)n;
        out->tri_ids.insert(out->tri_ids.end(), ids.begin() + begin, ids.begin() + begin + n);
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
            uint8_t *ks = kbuf.data();
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
        std::copy(tmp.begin() + begin, tmp.begin() + begin + nr, ids.begin() + begin + nl);

        build(begin, nl, L.lo, L.hi, depth - 1, best_cost);
        out->nodes[node].second_child = (uint32_t)out->nodes.size();                                  // linear_bvh.go:102
        build(begin + nl, nr, R.lo, R.hi, depth - 1, best_cost);
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

    Builder b;
    b.threshold = threshold;
    b.out = &out;
    out.nodes.clear();
    out.tri_ids.clear();
    b.tris.resize(ntris);
    b.ids.resize(ntris);
    b.tmp.resize(ntris);
    b.kbuf.resize(64);
    for (uint64_t k = 0; k < ntris; ++k) {          // bvh.go:250-272
        const float *a = verts + 3ull * tri_idx[3 * k], *bb = verts + 3ull * tri_idx[3 * k + 1],
                    *c = verts + 3ull * tri_idx[3 * k + 2];
        TriInfo &t = b.tris[k];
        for (int x = 0; x < 3; ++x) {
            t.cen[x] = ((a[x] + bb[x]) + c[x]) * (float)(1.0 / 3.0);
            t.mn[x] = gomin(gomin(a[x], bb[x]), c[x]);
            t.mx[x] = gomax(gomax(a[x], bb[x]), c[x]);
        }
        b.ids[k] = (uint32_t)k;
    }
    out.nodes.reserve(ntris / 2 + 16);
    out.tri_ids.reserve(ntris);
    const float lo[3] = {bounds[0], bounds[2], bounds[4]}, hi[3] = {bounds[1], bounds[3], bounds[5]};
    b.build(0, ntris, lo, hi, depth, MAXF);
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
