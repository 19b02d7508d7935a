// trav_device.hpp -- gfx950 device code for BVH traversal (closest-hit "extend", any-hit "connect").
//
// Results must equal the reference's Go traversal bit for bit (hit id) -- linear_bvh.go:23-70,
// 116-250 and utils.go:63-125 -- so every float expression below keeps the reference's operand
// order, is compiled with -ffp-contract=off (no FMA), uses IEEE division, and reproduces Go's
// NaN-propagating min/max.  The tree keeps the reference's binary topology and near-first
// order: only the memory layout changes (both child boxes hoisted into one 64-byte node).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptmi {

constexpr float PT_INF = 3.40282346638528859811704183484516925440e+38f;   // math.MaxFloat32 (linear_bvh.go:9)
constexpr uint32_t D_REF_LEAF_BIT = 0x80000000u;
constexpr uint32_t D_REF_EMPTY_LEAF = 0xFFFFFFFFu;

struct DevBVH {
    const float4 *nodes;   // 4 x float4 per packed inner node
    const float4 *tris;    // 3 x float4 per packed triangle
    uint32_t root_ref;
    uint32_t bounds_finite;   // every child box of every packed node is finite (enables slab_fast)
    uint32_t dummy_tri;       // record empty leaves point at (degenerate, never hit); only the counters care
    // union of the non-empty child boxes of all packed nodes (every leaf that holds a triangle lies inside) and a
    // margin that covers the rounding of the slab test for origins up to SEG_OUTSIDE_MAX_ORIGIN; 0 = not usable
    float tight_min[3], tight_max[3], tight_margin;
    uint32_t tight_valid;
};
constexpr float SEG_OUTSIDE_MAX_ORIGIN = 1.0e5f;

// True when the segment [o, o + d*step] (|d| <= ~1) lies outside the tree's tight box by more than the margin on some
// axis.  Then the slab test (linear_bvh.go:23-70) fails for every box inside the tight box -- on that axis both plane
// distances are negative, or both exceed `step` by far more than their rounding -- so the reference's walk pops the
// root's two children and ends without testing a triangle: the query is a miss, and the walk can be skipped.
__device__ __forceinline__ bool segment_outside_tree(const DevBVH &bvh, float ox, float oy, float oz, float dx, float dy,
                                                     float dz, float step) {
    if (!bvh.tight_valid) return false;
    const float ex = ox + dx * step, ey = oy + dy * step, ez = oz + dz * step;
    const float m = bvh.tight_margin;
    const bool ok = fabsf(ox) <= SEG_OUTSIDE_MAX_ORIGIN && fabsf(oy) <= SEG_OUTSIDE_MAX_ORIGIN && fabsf(oz) <= SEG_OUTSIDE_MAX_ORIGIN &&
                    fabsf(ex) <= SEG_OUTSIDE_MAX_ORIGIN && fabsf(ey) <= SEG_OUTSIDE_MAX_ORIGIN && fabsf(ez) <= SEG_OUTSIDE_MAX_ORIGIN;
    const bool out = fminf(ox, ex) > bvh.tight_max[0] + m || fmaxf(ox, ex) < bvh.tight_min[0] - m ||
                     fminf(oy, ey) > bvh.tight_max[1] + m || fmaxf(oy, ey) < bvh.tight_min[1] - m ||
                     fminf(oz, ez) > bvh.tight_max[2] + m || fmaxf(oz, ez) < bvh.tight_min[2] - m;
    return ok && out;          // a NaN anywhere makes `ok` or `out` false: the walk runs
}

// The same answer from the origin alone: when o is outside the tight box by more than margin + step (x 1.001: a
// normalised direction's components exceed 1 by rounding at most) on some axis, every segment of that length from o is
// outside by more than the margin there, whatever its direction -- segment_outside_tree would say yes.  The curved
// march asks this first and normalises the geodesic's direction only for the steps that come near the tree.
__device__ __forceinline__ bool point_far_from_tree(const DevBVH &bvh, float ox, float oy, float oz, float step) {
    if (!bvh.tight_valid) return false;
    const float m = bvh.tight_margin + step * 1.001f, lim = SEG_OUTSIDE_MAX_ORIGIN - step * 1.001f;
    const bool ok = fabsf(ox) <= lim && fabsf(oy) <= lim && fabsf(oz) <= lim && step >= 0.0f;
    const bool out = ox > bvh.tight_max[0] + m || ox < bvh.tight_min[0] - m || oy > bvh.tight_max[1] + m || oy < bvh.tight_min[1] - m ||
                     oz > bvh.tight_max[2] + m || oz < bvh.tight_min[2] - m;
    return ok && out;          // a NaN anywhere makes `ok` or `out` false
}

// Packed node quarters: q0 = c0min.x c0max.x c1min.x c1max.x | q1 = same for y | q2 = same for z |
// q3 = ref0 ref1 - -.  Child k's box is (q0[2k], q1[2k], q2[2k]) - (q0[2k+1], q1[2k+1], q2[2k+1]).
// Packed triangle: r0 = B.x C.x B.y C.y | r1 = B.z C.z A.x A.y | r2 = A.z prim nleft -.

// Go builtin max/min (NaN if either operand is NaN).  The sign of a zero result is not
// reproduced: slab distances are only ever compared, never output.
__device__ __forceinline__ float gomaxf(float a, float b) {
    float m = fmaxf(a, b);
    return (a != a || b != b) ? __builtin_nanf("") : m;
}
__device__ __forceinline__ float gominf(float a, float b) {
    float m = fminf(a, b);
    return (a != a || b != b) ? __builtin_nanf("") : m;
}

// linear_bvh.go:23-70 with inverseDirection hoisted out (it only depends on the ray).
__device__ __forceinline__ float slab(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                      float ox, float oy, float oz, float ix, float iy, float iz, float step) {
    float t1 = (minx - ox) * ix, t2 = (maxx - ox) * ix;
    float tmin = t1 > t2 ? t2 : t1;
    float tmax = t1 > t2 ? t1 : t2;
    t1 = (miny - oy) * iy; t2 = (maxy - oy) * iy;
    float a = t1 > t2 ? t2 : t1, b = t1 > t2 ? t1 : t2;
    tmin = gomaxf(tmin, a); tmax = gominf(tmax, b);
    t1 = (minz - oz) * iz; t2 = (maxz - oz) * iz;
    a = t1 > t2 ? t2 : t1; b = t1 > t2 ? t1 : t2;
    tmin = gomaxf(tmin, a); tmax = gominf(tmax, b);
    if (tmin > gominf(tmax, step) || tmax < 0.0f) return PT_INF;
    return gomaxf(0.0f, tmin);
}

// utils.go:63-125.  Returns true and t on a hit.
__device__ __forceinline__ bool segment_triangle(float ox, float oy, float oz, float dx, float dy, float dz,
                                                 float step, float ax, float ay, float az, float bx, float by,
                                                 float bz, float cx, float cy, float cz, float &t_out) {
    const float EPS = (float)1e-6;
    const float ONE_PLUS_EPS = (float)1.000001;
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    float det = (e1x * px + e1y * py) + e1z * pz;
    if (fabsf(det) < EPS) return false;
    float inv = 1.0f / det;
    float tx = ox - ax, ty = oy - ay, tz = oz - az;
    float u = ((tx * px + ty * py) + tz * pz) * inv;
    if (u < -EPS || u > ONE_PLUS_EPS) return false;
    float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    float v = ((dx * qx + dy * qy) + dz * qz) * inv;
    if (v < -EPS || u + v > ONE_PLUS_EPS) return false;
    float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    if (t <= EPS || t > step) return false;
    t_out = t;
    return true;
}

// linear_bvh.go:173-200
__device__ __forceinline__ bool shadow_triangle(float ox, float oy, float oz, float dx, float dy, float dz,
                                                float tmax, float ax, float ay, float az, float bx, float by,
                                                float bz, float cx, float cy, float cz) {
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
    float a = (e1x * hx + e1y * hy) + e1z * hz;
    if (fabsf(a) < (float)0.00001) return false;
    float sx = ox - ax, sy = oy - ay, sz = oz - az;
    float u = ((sx * hx + sy * hy) + sz * hz) / a;
    if (u < 0.0f || u > 1.0f) return false;
    float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    float v = ((dx * qx + dy * qy) + dz * qz) / a;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = ((e2x * qx + e2y * qy) + e2z * qz) / a;
    return t > (float)0.00001 && t < tmax;
}

struct TravCounters {
    uint32_t inner = 0, tri = 0, leaf = 0;
};

// Per-lane traversal stack in LDS, laid out [entry][lane of the workgroup]: a lane's bank
// depends only on its lane id, so pushes/pops at different depths never conflict.
template <int BLOCK>
struct LdsStack {
    uint32_t *base;   // &smem[threadIdx.x]
    int sp = 0;
    __device__ __forceinline__ void push(uint32_t v) { base[sp * BLOCK] = v; ++sp; }
    __device__ __forceinline__ uint32_t pop() { --sp; return base[sp * BLOCK]; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};

__device__ __forceinline__ float slab_fast(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                           float ox, float oy, float oz, float ix, float iy, float iz, float step);
__device__ __forceinline__ bool finitef(float x) { return fabsf(x) <= PT_INF; }   // false for NaN and +-Inf
// true when the ray must take the NaN-exact slab (see slab_fast below)
__device__ __forceinline__ bool ray_needs_exact(const DevBVH &bvh, float ox, float oy, float oz, float ix, float iy, float iz,
                                                float step) {
    return !(finitef(ix) && finitef(iy) && finitef(iz) && finitef(ox) && finitef(oy) && finitef(oz) && finitef(step)) ||
           !bvh.bounds_finite;
}
template <bool EXACT>
__device__ __forceinline__ float slab_sel(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                          float ox, float oy, float oz, float ix, float iy, float iz, float step) {
    return EXACT ? slab(minx, miny, minz, maxx, maxy, maxz, ox, oy, oz, ix, iy, iz, step)
                 : slab_fast(minx, miny, minz, maxx, maxy, maxz, ox, oy, oz, ix, iy, iz, step);
}

constexpr uint32_t D_REF_DONE = 0xFFFFFFFEu;     // traversal finished (has the leaf bit, never a real leaf)
// The one-shot walks (integrator) descend while-while: the inner-node loop runs until every lane of the wave is at a
// leaf or done, then the leaves are tested together.  Against the if-if form (one node or one leaf per iteration,
// PTMI_WHILE_WHILE=0) connect is 9 % faster and extend equal; postponing the first leaf to descend further
// (speculative while-while, legal for any-hit) was 5 % slower again.
#ifndef PTMI_WHILE_WHILE
#define PTMI_WHILE_WHILE 1
#endif

// CheckIntersection (linear_bvh.go:116-170).  prim = -1 on a miss.
template <int BLOCK, bool COUNT, bool EXACT>
__device__ __forceinline__ void closest_hit_impl(const DevBVH &bvh, uint32_t *stack_base, float ox, float oy, float oz,
                                                 float dx, float dy, float dz, float ix, float iy, float iz, float step,
                                                 int &prim_out, float &t_out, TravCounters &cnt) {
    LdsStack<BLOCK> st{stack_base};
    float best_t = step;
    int best = -1;
    uint32_t cur = bvh.root_ref;
#if PTMI_WHILE_WHILE
    for (;;) {
        while (!(cur & D_REF_LEAF_BIT)) {           // descend until every lane of the wave is at a leaf (or done)
            if (COUNT) cnt.inner++;
            const float4 *p = bvh.nodes + (size_t)cur * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float d1 = slab_sel<EXACT>(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, ox, oy, oz, ix, iy, iz, best_t);
            float d2 = slab_sel<EXACT>(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, ox, oy, oz, ix, iy, iz, best_t);
            uint32_t i = __float_as_uint(q3.x), j = __float_as_uint(q3.y);
            if (d1 > d2) { uint32_t ti = i; i = j; j = ti; float td = d1; d1 = d2; d2 = td; }   // :148-151
            if (d1 == PT_INF) cur = st.empty() ? D_REF_DONE : st.pop();                          // :152-154
            else { if (d2 < best_t) st.push(j); cur = i; }                                       // :156-160
        }
        if (cur == D_REF_DONE) break;
        {
            if (COUNT) cnt.leaf++;
            const uint32_t first = cur & 0x7FFFFFFFu;
            const float4 *p = bvh.tris + (size_t)first * 3;
            for (;;) {
                const float4 r0 = p[0], r1 = p[1], r2 = p[2];
                if (COUNT && first != bvh.dummy_tri) cnt.tri++;
                float t;
                bool hit = segment_triangle(ox, oy, oz, dx, dy, dz, best_t, r1.z, r1.w, r2.x, r0.x, r0.z, r1.x,
                                            r0.y, r0.w, r1.y, t);
                if (hit && t < best_t && t > 0.0f) { best_t = t; best = (int)__float_as_uint(r2.y); }   // :133-136
                if (__float_as_uint(r2.z) == 0u) break;
                p += 3;
            }
        }
        cur = st.empty() ? D_REF_DONE : st.pop();
    }
#else
    for (;;) {
        if (!(cur & D_REF_LEAF_BIT)) {
            if (COUNT) cnt.inner++;
            const float4 *p = bvh.nodes + (size_t)cur * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float d1 = slab_sel<EXACT>(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, ox, oy, oz, ix, iy, iz, best_t);
            float d2 = slab_sel<EXACT>(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, ox, oy, oz, ix, iy, iz, best_t);
            uint32_t i = __float_as_uint(q3.x), j = __float_as_uint(q3.y);
            if (d1 > d2) { uint32_t ti = i; i = j; j = ti; float td = d1; d1 = d2; d2 = td; }   // :148-151
            if (d1 == PT_INF) {                                                                  // :152-154
                if (st.empty()) break;
                cur = st.pop();
                continue;
            }
            if (d2 < best_t) st.push(j);                                                         // :156-159
            cur = i;                                                                             // :160 (pushed, popped next)
            continue;
        }
        {
            if (COUNT) cnt.leaf++;
            const uint32_t first = cur & 0x7FFFFFFFu;
            const float4 *p = bvh.tris + (size_t)first * 3;
            for (;;) {
                const float4 r0 = p[0], r1 = p[1], r2 = p[2];
                if (COUNT && first != bvh.dummy_tri) cnt.tri++;
                float t;
                bool hit = segment_triangle(ox, oy, oz, dx, dy, dz, best_t, r1.z, r1.w, r2.x, r0.x, r0.z, r1.x,
                                            r0.y, r0.w, r1.y, t);
                if (hit && t < best_t && t > 0.0f) { best_t = t; best = (int)__float_as_uint(r2.y); }   // :133-136
                if (__float_as_uint(r2.z) == 0u) break;
                p += 3;
            }
        }
        if (st.empty()) break;
        cur = st.pop();
    }
#endif
    prim_out = best;
    t_out = best >= 0 ? best_t : 0.0f;
}
// The walk takes the cheaper NaN-free slab when no active lane of the wave needs the exact one.
template <int BLOCK, bool COUNT>
__device__ __forceinline__ void closest_hit(const DevBVH &bvh, uint32_t *stack_base, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float step, int &prim_out, float &t_out,
                                            TravCounters &cnt) {
    // Inverse() (vec3.go:118-124); the reference recomputes it per box with the same result
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    if (__any(ray_needs_exact(bvh, ox, oy, oz, ix, iy, iz, step)))
        closest_hit_impl<BLOCK, COUNT, true>(bvh, stack_base, ox, oy, oz, dx, dy, dz, ix, iy, iz, step, prim_out, t_out, cnt);
    else
        closest_hit_impl<BLOCK, COUNT, false>(bvh, stack_base, ox, oy, oz, dx, dy, dz, ix, iy, iz, step, prim_out, t_out, cnt);
}

// ---- step-wise form of the same walk (used by the persistent kernel) --------------------------------
//
// slab_fast: linear_bvh.go:23-70 for rays whose slab products cannot be NaN.  A product (bound - o) * inv
// is NaN only for 0*inf, inf*0 or NaN inputs; with finite bounds (checked at upload), finite origin, finite
// step and finite 1/d every product is finite or +-inf, and on non-NaN values Go's min/max and the
// `if t1 > t2 swap` reduce to plain min/max (the sign of a zero distance is never observable).  Rays that
// do not qualify (a zero / denormal direction component, non-finite input) take the exact `slab`.
__device__ __forceinline__ float slab_fast(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                           float ox, float oy, float oz, float ix, float iy, float iz, float step) {
    float t1 = (minx - ox) * ix, t2 = (maxx - ox) * ix;
    float lo = fminf(t1, t2), hi = fmaxf(t1, t2);
    t1 = (miny - oy) * iy; t2 = (maxy - oy) * iy;
    lo = fmaxf(lo, fminf(t1, t2)); hi = fminf(hi, fmaxf(t1, t2));
    t1 = (minz - oz) * iz; t2 = (maxz - oz) * iz;
    lo = fmaxf(lo, fminf(t1, t2)); hi = fminf(hi, fmaxf(t1, t2));
    const bool miss = lo > fminf(hi, step) || hi < 0.0f;
    return miss ? PT_INF : fmaxf(0.0f, lo);
}

// Both children of a packed node at once.  (Packed-f32 forms of the twelve products were measured: on
// gfx950 v_pk_mul_f32 issues at half the rate of v_mul_f32, so they buy nothing -- tools/ubench_pk.hip.)
__device__ __forceinline__ void slab_fast2(const float4 q0, const float4 q1, const float4 q2, float ox, float oy,
                                           float oz, float ix, float iy, float iz, float step, float &d1, float &d2) {
    d1 = slab_fast(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, ox, oy, oz, ix, iy, iz, step);
    d2 = slab_fast(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, ox, oy, oz, ix, iy, iz, step);
}


struct RayTrav {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    float best_t;
    int best;
    uint32_t best_aux;   // the fourth word of the best triangle's record (AUX walks only: the integrator keeps the material class there)
    uint32_t cur;
    int sp;
    bool exact;          // needs the NaN-exact slab
};

__device__ __forceinline__ void trav_init(RayTrav &r, const DevBVH &bvh, float ox, float oy, float oz, float dx, float dy,
                                          float dz, float step) {
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;        // Inverse() (vec3.go:118-124)
    r.best_t = step; r.best = -1; r.best_aux = 0u; r.cur = bvh.root_ref; r.sp = 0;
    r.exact = ray_needs_exact(bvh, ox, oy, oz, r.ix, r.iy, r.iz, step);
}

// ---- single-step forms for the vote-scheduled kernel -------------------------------------------------
//
// A lane is in one of three states, encoded in r.cur: at an inner node (leaf bit clear), at a triangle
// record (leaf bit set; low bits = record index, advanced one record per step) or DONE.  Each wave
// iteration runs ONE step type for the lanes in that state, chosen by majority, so no lane waits for a
// neighbour's whole leaf or whole descent.  Both steps are branch-free apart from the LDS stack access.

// pop the next reference (or DONE) -- linear_bvh.go:124-126
template <int BLOCK>
__device__ __forceinline__ uint32_t trav_pop(uint32_t *stack_base, int &sp) {
    if (sp == 0) return D_REF_DONE;
    --sp;
    return stack_base[sp * BLOCK];
}

// linear_bvh.go:138-162 for one inner node; r.cur must be an inner reference
template <int BLOCK, bool COUNT, bool EXACT>
__device__ __forceinline__ void vote_inner_step(const DevBVH &bvh, uint32_t *stack_base, RayTrav &r, TravCounters &cnt) {
    if (COUNT) cnt.inner++;
    const float4 *p = bvh.nodes + (size_t)r.cur * 4;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    float d1, d2;
    if (EXACT) {
        d1 = slab(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.best_t);
        d2 = slab(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.best_t);
    } else {
        slab_fast2(q0, q1, q2, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, r.best_t, d1, d2);
    }
    const uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
    const bool sw = d1 > d2;                                        // :148-151
    const uint32_t near = sw ? r1 : r0, far = sw ? r0 : r1;
    const float dn = sw ? d2 : d1, df = sw ? d1 : d2;
    if (dn == PT_INF) {                                             // :152-154
        r.cur = trav_pop<BLOCK>(stack_base, r.sp);
    } else {
        if (df < r.best_t) { stack_base[r.sp * BLOCK] = far; ++r.sp; }   // :156-159
        r.cur = near;                                               // :160
    }
}

// One triangle of a leaf (linear_bvh.go:129-137 + utils.go:63-125), branch-free: both cross products and all four
// dot products are evaluated up front, each operation as the source line writes it, and the early returns of
// utils.go become one predicate (a NaN fails each `<`/`>` exactly as in the reference).  Scalar f32 on purpose:
// v_pk_mul/add_f32 issue at half rate on gfx950 (tools/ubench_pk.hip) and need extra moves.
template <int BLOCK, bool COUNT, bool AUX = false>
__device__ __forceinline__ void vote_tri_step(const DevBVH &bvh, uint32_t *stack_base, RayTrav &r, TravCounters &cnt) {
    const float EPS = (float)1e-6;
    const float ONE_PLUS_EPS = (float)1.000001;
    const uint32_t rec = r.cur & 0x7FFFFFFFu;
    const float4 *p = bvh.tris + (size_t)rec * 3;
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    const float ax = r1.z, ay = r1.w, az = r2.x;
    const float e1x = r0.x - ax, e1y = r0.z - ay, e1z = r1.x - az;      // edge1 = B - A    utils.go:70
    const float e2x = r0.y - ax, e2y = r0.w - ay, e2z = r1.y - az;      // edge2 = C - A    utils.go:71
    const float px = r.dy * e2z - r.dz * e2y, py = r.dz * e2x - r.dx * e2z, pz = r.dx * e2y - r.dy * e2x;   // :77-81
    const float det = (e1x * px + e1y * py) + e1z * pz;                 // :83
    const float inv = 1.0f / det;                                       // :88
    const float tx = r.ox - ax, ty = r.oy - ay, tz = r.oz - az;         // :91
    const float u = ((tx * px + ty * py) + tz * pz) * inv;              // :93
    const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;               // :103-107
    const float v = ((r.dx * qx + r.dy * qy) + r.dz * qz) * inv;        // :108
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;           // :116
    bool hit = !(fabsf(det) < EPS);                                          // :84-86
    hit = hit && !(u < -EPS || u > ONE_PLUS_EPS);                            // :96-98
    hit = hit && !(v < -EPS || u + v > ONE_PLUS_EPS);                        // :111-113
    hit = hit && !(t <= EPS || t > r.best_t);                                // :118-120 (stepSize = best_t, linear_bvh.go:132)
    if (COUNT) { if (rec != bvh.dummy_tri) cnt.tri++; }
    if (hit && t < r.best_t && t > 0.0f) {                                                        // linear_bvh.go:133-136
        r.best_t = t; r.best = (int)__float_as_uint(r2.y);
        if (AUX) r.best_aux = __float_as_uint(r2.w);
    }
    if (__float_as_uint(r2.z) == 0u) {                  // last record of the leaf
        if (COUNT) cnt.leaf++;
        r.cur = trav_pop<BLOCK>(stack_base, r.sp);
    } else {
        r.cur = r.cur + 1u;
    }
}

// ---- any-hit forms of the two steps (QuickCheckIntersection, linear_bvh.go:202-250) -------------------
// r.best_t holds the (constant) range, r.best becomes 1 when an occluder is found and the walk stops.

// linear_bvh.go:218-246 for one inner node.  FAR: when the segment enters both children, take the FAR one first.  The
// occlusion bit cannot depend on the order (the range is constant and boxes are pruned one by one, :226-245), so the
// walk is free to choose it, and shadow rays find their occluder sooner from the far end: they start ON a surface, whose
// neighbourhood fills the near boxes and hardly ever occludes (the ray is offset from it and leaves it behind), while
// an occluded ray walks every near box before it reaches what blocks it -- by the counters a third of C3's shadow rays
// are occluded and make half of all node visits, 24 per ray against 11 for a free one (tools/probe_occluded.py).
// Far first: connect -11 % on C3, -10 % on the 10.4 M-triangle scene; the longer overlap first -5 / -7 %, the box that
// holds the origin last -9 / -6 %, the smaller box first +4 / -2 %.  The reference's order stays for the batch
// kernels' counting variant, whose counters are compared with the oracle's.
template <int BLOCK, bool COUNT, bool EXACT, bool FAR = false>
__device__ __forceinline__ void vote_inner_step_any(const DevBVH &bvh, uint32_t *stack_base, RayTrav &r, TravCounters &cnt) {
    if (COUNT) cnt.inner++;
    const float4 *p = bvh.nodes + (size_t)r.cur * 4;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const float step = r.best_t;
    float d1, d2;
    if (EXACT) {
        d1 = slab(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, step);
        d2 = slab(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, step);
    } else {
        slab_fast2(q0, q1, q2, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz, step, d1, d2);
    }
    const uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
    const bool h1 = d1 < step, h2 = d2 < step;
    if (h1 && h2) {                                                 // :226-238 far child pushed first, near on top
        const bool first0 = FAR ? !(d1 < d2) : d1 < d2;
        stack_base[r.sp * BLOCK] = first0 ? r1 : r0; ++r.sp;
        r.cur = first0 ? r0 : r1;
    } else if (h1 || h2) {                                          // :239-244
        r.cur = h1 ? r0 : r1;
    } else {
        r.cur = trav_pop<BLOCK>(stack_base, r.sp);
    }
}

// One triangle of a leaf with FastIntersectShadowTriangle (linear_bvh.go:173-200): the early returns become one
// predicate evaluated in the source's order (a NaN fails each compare as it does there; the three quotients
// are only looked at when |a| >= 1e-5).
template <int BLOCK, bool COUNT>
__device__ __forceinline__ void vote_tri_step_any(const DevBVH &bvh, uint32_t *stack_base, RayTrav &r, TravCounters &cnt) {
    const uint32_t rec = r.cur & 0x7FFFFFFFu;
    const float4 *p = bvh.tris + (size_t)rec * 3;
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    const float ax = r1.z, ay = r1.w, az = r2.x;
    const float e1x = r0.x - ax, e1y = r0.z - ay, e1z = r1.x - az;
    const float e2x = r0.y - ax, e2y = r0.w - ay, e2z = r1.y - az;
    const float hx = r.dy * e2z - r.dz * e2y, hy = r.dz * e2x - r.dx * e2z, hz = r.dx * e2y - r.dy * e2x;
    const float a = (e1x * hx + e1y * hy) + e1z * hz;
    const float sx = r.ox - ax, sy = r.oy - ay, sz = r.oz - az;
    const float u = ((sx * hx + sy * hy) + sz * hz) / a;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = ((r.dx * qx + r.dy * qy) + r.dz * qz) / a;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) / a;
    bool hit = !(fabsf(a) < (float)0.00001);
    hit = hit && !(u < 0.0f || u > 1.0f);
    hit = hit && !(v < 0.0f || u + v > 1.0f);
    hit = hit && (t > (float)0.00001 && t < r.best_t);
    if (COUNT) { if (rec != bvh.dummy_tri) cnt.tri++; }
    const bool last = __float_as_uint(r2.z) == 0u;
    if (COUNT) { if (last || hit) cnt.leaf++; }
    if (hit) { r.best = 1; r.sp = 0; r.cur = D_REF_DONE; }
    else if (last) r.cur = trav_pop<BLOCK>(stack_base, r.sp);
    else r.cur = r.cur + 1u;
}

// ---- the vote-scheduled walk of a persistent wave -------------------------------------------------------
// Shared by the batch kernels (ptmi_api.hip) and the integrator's extend / connect (ptmi_render.hip); they differ only
// in where a ray comes from (`fetch(idx, ray, tag)`) and where its answer goes (`store(tag, r)`).
//
// Every iteration the wave counts its lanes at an inner node (I), at a triangle record (T) and idle with a ray in
// reserve (R), and runs ONE step type for the lanes in that state: refill when `refill_min` lanes are idle (or nothing
// else is left), otherwise the majority of I / T (I wins when nI + tri_bias >= nT).  A waiting state's population only
// grows until it is served, so no lane starves.
//
// Rays are dealt in chunks of `chunk_rays` from one global cursor, in batch order: at any moment the whole chip works
// on neighbouring rays, which keeps the touched part of the tree in L2 (dealing chunks statically per wave was measured
// 25 % slower for that reason).  The cursor is only touched once per chunk and two chunks ahead of use, and each lane
// keeps its NEXT ray prefetched in registers, loaded while the current one is walked, so neither the atomic nor the
// ray fetch is ever waited for.  The loop ends when the cursor has passed n and every lane has drained and stored its
// result.  ANY selects QuickCheckIntersection's walk (r.best = 1 when occluded), otherwise CheckIntersection's.
// Returns the number of rays this lane walked.  The workgroup is one wave.
struct NextRay { float ox, oy, oz, dx, dy, dz, tmax; };

template <int BLOCK, bool ANY, bool COUNT, typename Tag, bool AUX = false, bool FAR = false, typename Fetch, typename Store>
__device__ __forceinline__ uint32_t vote_walk(const DevBVH &bvh, uint32_t *stack_base, uint64_t n, unsigned long long *cursor,
                                              int refill_min, int tri_bias, uint32_t chunk_rays, TravCounters &cnt,
                                              Fetch fetch, Store store) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    RayTrav r;
    r.cur = D_REF_DONE; r.exact = false; r.sp = 0; r.best = ANY ? 0 : -1; r.best_t = 0.0f; r.best_aux = 0u;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
    Tag my_tag{}, next_tag{};
    NextRay nx = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool pending = false, have_next = false;
    uint32_t nrays = 0;
    // chunk pipeline (wave-uniform): `chunk` is being handed out, `chunk1` is ready, `inflight` (lane 0) is on its way
    auto grab = [&]() -> unsigned long long {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(cursor, (unsigned long long)chunk_rays);
        return v;
    };
    auto uniform = [&](unsigned long long v) -> uint64_t {
        return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
               (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    };
    uint64_t chunk = uniform(grab()), chunk1 = uniform(grab());
    unsigned long long inflight = grab();
    uint32_t used = 0;                      // rays of `chunk` already reserved
    auto reserve = [&](uint32_t k) {        // k-th ray counted from the start of `chunk` -> prefetch registers
        const uint64_t idx = k < chunk_rays ? chunk + k : chunk1 + (k - chunk_rays);
        have_next = idx < n;
        if (have_next) fetch(idx, nx, next_tag);
    };
    auto advance = [&](uint32_t taken) {
        used += taken;
        if (used >= chunk_rays) {
            used -= chunk_rays;
            chunk = chunk1;
            chunk1 = uniform(inflight);
            inflight = grab();
        }
    };
    reserve(lane);
    advance(64);
    for (;;) {
        const bool at_inner = !(r.cur & D_REF_LEAF_BIT);
        const bool at_tri = (r.cur & D_REF_LEAF_BIT) && r.cur != D_REF_DONE;
        const bool idle = r.cur == D_REF_DONE;
        const int nI = __popcll(__ballot(at_inner)), nT = __popcll(__ballot(at_tri));
        const unsigned long long takers = __ballot(idle && have_next);
        const int nR = __popcll(takers);
        if (nI + nT == 0 || nR >= refill_min) {
            // ---- R step: store finished results, start the prefetched rays, reserve the next ones ----
            if (idle && pending) { store(my_tag, r); pending = false; }
            if (nR == 0) break;             // nothing left to start, nothing in flight: the wave is done
            if (idle && have_next) {
                my_tag = next_tag; pending = true; ++nrays;
                trav_init(r, bvh, nx.ox, nx.oy, nx.oz, nx.dx, nx.dy, nx.dz, nx.tmax);
                if (ANY) r.best = 0;
                reserve(used + (uint32_t)__popcll(takers & lt_mask));
            }
            advance((uint32_t)nR);
            continue;
        }
        if (nI > 0 && (nT == 0 || nI + tri_bias >= nT)) {
            if (__any(at_inner && r.exact)) {
                if (at_inner) { if (ANY) vote_inner_step_any<BLOCK, COUNT, true, FAR>(bvh, stack_base, r, cnt); else vote_inner_step<BLOCK, COUNT, true>(bvh, stack_base, r, cnt); }
            } else {
                if (at_inner) { if (ANY) vote_inner_step_any<BLOCK, COUNT, false, FAR>(bvh, stack_base, r, cnt); else vote_inner_step<BLOCK, COUNT, false>(bvh, stack_base, r, cnt); }
            }
        } else {
            if (at_tri) { if (ANY) vote_tri_step_any<BLOCK, COUNT>(bvh, stack_base, r, cnt); else vote_tri_step<BLOCK, COUNT, AUX>(bvh, stack_base, r, cnt); }
        }
    }
    return nrays;
}

// QuickCheckIntersection (linear_bvh.go:202-250); FAR as in vote_inner_step_any
template <int BLOCK, bool COUNT, bool EXACT, bool FAR>
__device__ __forceinline__ bool any_hit_impl(const DevBVH &bvh, uint32_t *stack_base, float ox, float oy, float oz,
                                             float dx, float dy, float dz, float ix, float iy, float iz, float step,
                                             TravCounters &cnt) {
    LdsStack<BLOCK> st{stack_base};
    uint32_t cur = bvh.root_ref;
#if PTMI_WHILE_WHILE
    for (;;) {
        while (!(cur & D_REF_LEAF_BIT)) {
            if (COUNT) cnt.inner++;
            const float4 *p = bvh.nodes + (size_t)cur * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float d1 = slab_sel<EXACT>(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, ox, oy, oz, ix, iy, iz, step);
            float d2 = slab_sel<EXACT>(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, ox, oy, oz, ix, iy, iz, step);
            uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
            bool h1 = d1 < step, h2 = d2 < step;
            if (h1 && h2) {                              // :226-238 push far first, near on top
                if (FAR ? !(d1 < d2) : d1 < d2) { st.push(r1); cur = r0; } else { st.push(r0); cur = r1; }
            } else if (h1) cur = r0;                     // :239-241
            else if (h2) cur = r1;                       // :242-244
            else cur = st.empty() ? D_REF_DONE : st.pop();
        }
        if (cur == D_REF_DONE) return false;
        {
            if (COUNT) cnt.leaf++;
            const uint32_t first = cur & 0x7FFFFFFFu;
            const float4 *p = bvh.tris + (size_t)first * 3;
            for (;;) {
                const float4 r0 = p[0], r1 = p[1], r2 = p[2];
                if (COUNT && first != bvh.dummy_tri) cnt.tri++;
                if (shadow_triangle(ox, oy, oz, dx, dy, dz, step, r1.z, r1.w, r2.x, r0.x, r0.z, r1.x, r0.y, r0.w,
                                    r1.y))
                    return true;
                if (__float_as_uint(r2.z) == 0u) break;
                p += 3;
            }
        }
        cur = st.empty() ? D_REF_DONE : st.pop();
    }
#else
    for (;;) {
        if (!(cur & D_REF_LEAF_BIT)) {
            if (COUNT) cnt.inner++;
            const float4 *p = bvh.nodes + (size_t)cur * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            float d1 = slab_sel<EXACT>(q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, ox, oy, oz, ix, iy, iz, step);
            float d2 = slab_sel<EXACT>(q0.z, q1.z, q2.z, q0.w, q1.w, q2.w, ox, oy, oz, ix, iy, iz, step);
            uint32_t r0 = __float_as_uint(q3.x), r1 = __float_as_uint(q3.y);
            bool h1 = d1 < step, h2 = d2 < step;
            if (h1 && h2) {                              // :226-238 push far first, near on top
                if (FAR ? !(d1 < d2) : d1 < d2) { st.push(r1); cur = r0; } else { st.push(r0); cur = r1; }
                continue;
            } else if (h1) { cur = r0; continue; }       // :239-241
            else if (h2) { cur = r1; continue; }         // :242-244
            if (st.empty()) return false;
            cur = st.pop();
            continue;
        }
        {
            if (COUNT) cnt.leaf++;
            const uint32_t first = cur & 0x7FFFFFFFu;
            const float4 *p = bvh.tris + (size_t)first * 3;
            for (;;) {
                const float4 r0 = p[0], r1 = p[1], r2 = p[2];
                if (COUNT && first != bvh.dummy_tri) cnt.tri++;
                if (shadow_triangle(ox, oy, oz, dx, dy, dz, step, r1.z, r1.w, r2.x, r0.x, r0.z, r1.x, r0.y, r0.w,
                                    r1.y))
                    return true;
                if (__float_as_uint(r2.z) == 0u) break;
                p += 3;
            }
        }
        if (st.empty()) return false;
        cur = st.pop();
    }
#endif
}
template <int BLOCK, bool COUNT, bool FAR = false>
__device__ __forceinline__ bool any_hit(const DevBVH &bvh, uint32_t *stack_base, float ox, float oy, float oz,
                                        float dx, float dy, float dz, float step, TravCounters &cnt) {
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    if (__any(ray_needs_exact(bvh, ox, oy, oz, ix, iy, iz, step)))
        return any_hit_impl<BLOCK, COUNT, true, FAR>(bvh, stack_base, ox, oy, oz, dx, dy, dz, ix, iy, iz, step, cnt);
    return any_hit_impl<BLOCK, COUNT, false, FAR>(bvh, stack_base, ox, oy, oz, dx, dy, dz, ix, iy, iz, step, cnt);
}

}  // namespace ptmi
