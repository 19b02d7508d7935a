// bvh_build.hpp -- host-side BVH builder and GPU packing for libptmi.
//
// Produces a tree that is bit-identical (topology, bounds, leaf order) to the reference's
// BuildBVH + ConstructLinearBVH (bvh.go:148-275, linear_bvh.go:77-112), using a binned
// one-pass-per-axis evaluation of the 387 candidate planes instead of 387 re-scans.
#pragma once
#include <cstdint>
#include <vector>

namespace ptmi {

// linear_bvh.go:11-21
struct LinNode {
    float minb[3], maxb[3];
    uint8_t is_leaf;
    uint32_t tri_offset, tri_count, second_child;
};

// linear_bvh.go:72-75 (Triangles as triangle numbers = tri.Index/3)
struct LinearBVH {
    std::vector<LinNode> nodes;
    std::vector<uint32_t> tri_ids;
};

struct BvhStats {   // bvh.go:109-146
    int64_t max_depth = 0, max_tris = 0, min_tris = 0, total_leafs = 0, total_nodes = 0, total_triangles = 0;
};

// 64-byte packed inner node: both children's boxes + child references, one axis per 16-byte
// quarter (min,max of child 0, min,max of child 1).
//   ref: bit 31 set  -> leaf; low 31 bits = offset of its first triangle record (an empty
//                       leaf points at the shared degenerate record PackedBVH::dummy_tri)
//        bit 31 clear -> index of a packed inner node
struct alignas(64) PackedNode {
    float x[4];        // c0min.x c0max.x c1min.x c1max.x
    float y[4];
    float z[4];
    uint32_t ref0, ref1;
    uint32_t pad0, pad1;
};
static_assert(sizeof(PackedNode) == 64, "packed node must be one 64-byte fetch");

// 48-byte triangle record, stored in leaf (DFS) order.  A,B,C verbatim (edges are NOT
// precomputed so that edge1 = B - A is evaluated exactly as utils.go:70-71 does).
// Layout pairs B with C per axis: (B.x,C.x) - (A.x,A.x) = (edge1.x, edge2.x) is one packed subtract.
struct alignas(16) PackedTri {
    float BCx[2], BCy[2];          // B.x C.x B.y C.y
    float BCz[2], Axy[2];          // B.z C.z A.x A.y
    float Az;
    uint32_t prim;      // tri.Index/3
    uint32_t nleft;     // triangles that follow this one in the same leaf
    uint32_t pad;
};
static_assert(sizeof(PackedTri) == 48, "packed triangle must be 48 bytes");

constexpr uint32_t REF_LEAF_BIT = 0x80000000u;
constexpr uint32_t REF_EMPTY_LEAF = 0xFFFFFFFFu;

struct PackedBVH {
    std::vector<PackedNode> nodes;
    std::vector<PackedTri> tris;
    uint32_t root_ref = REF_EMPTY_LEAF;
    uint32_t dummy_tri = 0;         // index of the degenerate record (A=B=C=0, never hit) empty leaves point at
    int max_depth = 0;              // levels, root = 1
    bool bounds_finite = true;      // no NaN / Inf in any child box
};

// Returns 0 or a negative error; msg set on error.
int build_linear_bvh(const float *verts, uint64_t nverts, const uint32_t *tri_idx, uint64_t ntris,
                     const float bounds[6], int threshold, int depth, LinearBVH &out, const char **msg);

int validate_linear_bvh(const LinearBVH &bvh, uint64_t ntris, const char **msg);
BvhStats linear_bvh_stats(const LinearBVH &bvh);
int pack_bvh(const LinearBVH &bvh, const float *verts, const uint32_t *tri_idx, PackedBVH &out, const char **msg);

}  // namespace ptmi
