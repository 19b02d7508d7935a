/*
 * oracle/orc_bvh.c -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
 * PARITY UNPINNED (no reference fixtures exist).
 *
 * Literal CPU restatement of the reference's BVH path:
 *   BuildBVH / buildBVH / getSplit / FauxBox / Box.Grow     bvh.go:38-107,148-275
 *   convert / ConstructLinearBVH                            linear_bvh.go:77-112
 *   intersectAABB                                           linear_bvh.go:23-70
 *   IntersectSegmentTriangle                                utils.go:63-125
 *   CheckIntersection                                       linear_bvh.go:116-170
 *   FastIntersectShadowTriangle                             linear_bvh.go:173-200
 *   QuickCheckIntersection                                  linear_bvh.go:202-250
 *
 * The builder is the literal O(387*N) per node algorithm (every candidate plane
 * re-scans every triangle); the only liberty taken is that the 387 candidate
 * costs of one node may be evaluated by several threads -- they are then
 * compared in the reference's order (axis-major, i-minor, strict <).
 */
#include "orc_bvh.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ---------------------------------------------------------------- builder */

/* bvh.go:38-44 */
typedef struct { float x1, x2, y1, y2, z1, z2; int64_t count; } fauxbox;

/* bvh.go:46-63 (FauxBox.Grow) == bvh.go:82-99 (Box.Grow) for the bounds */
static inline void faux_grow(fauxbox *b, const orc_tri *t) {
    if (b->count == 0) {
        b->x1 = t->minx; b->x2 = t->maxx;
        b->y1 = t->miny; b->y2 = t->maxy;
        b->z1 = t->minz; b->z2 = t->maxz;
    } else {
        b->x1 = gominf(b->x1, t->minx); b->x2 = gomaxf(b->x2, t->maxx);
        b->y1 = gominf(b->y1, t->miny); b->y2 = gomaxf(b->y2, t->maxy);
        b->z1 = gominf(b->z1, t->minz); b->z2 = gomaxf(b->z2, t->maxz);
    }
    b->count += 1;
}
/* bvh.go:64-70 */
static inline float faux_area(const fauxbox *b) {
    float dx = b->x2 - b->x1, dy = b->y2 - b->y1, dz = b->z2 - b->z1;
    return 2.0f * ((dx * dy + dy * dz) + dx * dz);
}

/* bvh.go:72-80 pointer tree */
typedef struct box {
    float x1, x2, y1, y2, z1, z2;
    int is_leaf;
    uint32_t *tris; uint64_t ntris;
    struct box *child[2];
} box;

static inline float tri_centroid_axis(const orc_tri *t, int axis) {
    return axis == 0 ? t->centroid.x : (axis == 1 ? t->centroid.y : t->centroid.z);
}

/* cost of one candidate plane: bvh.go:172-201 */
static float candidate_cost(const orc_tri *T, const uint32_t *ids, uint64_t n, int axis, float mid) {
    fauxbox L, R;
    memset(&L, 0, sizeof L); memset(&R, 0, sizeof R);
    for (uint64_t k = 0; k < n; ++k) {
        const orc_tri *t = &T[ids[k]];
        if (tri_centroid_axis(t, axis) < mid) faux_grow(&L, t); else faux_grow(&R, t);
    }
    /* 1.0/8 + float32(nL)*areaL + float32(nR)*areaR, left-associated */
    return (0.125f + (float)L.count * faux_area(&L)) + (float)R.count * faux_area(&R);
}

static box *build_rec(const orc_tri *T, uint32_t *ids, uint64_t n,
                      float x1, float x2, float y1, float y2, float z1, float z2,
                      int threshold, int depth, float parent_cost) {
    box *b = (box *)calloc(1, sizeof(box));
    b->x1 = x1; b->x2 = x2; b->y1 = y1; b->y2 = y2; b->z1 = z1; b->z2 = z2;

    /* bvh.go:153-158 */
    if ((int64_t)n <= (int64_t)threshold || depth <= 0) {
        b->is_leaf = 1; b->tris = ids; b->ntris = n;
        return b;
    }
    const float starts[3] = { x1, y1, z1 }, ends[3] = { x2, y2, z2 };
    float costs[3][129], mids[3][129];
    /* bvh.go:167-172   mid := l + (r-l)*float32(i)/float32(n)  ==  l + (((r-l)*i)/128) */
    for (int axis = 0; axis < 3; ++axis)
        for (int i = 0; i <= 128; ++i) {
            float l = starts[axis], r = ends[axis];
            mids[axis][i] = l + ((r - l) * (float)i) / (float)128;
        }
    if (n >= 4096) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < 387; ++c)
            costs[c / 129][c % 129] = candidate_cost(T, ids, n, c / 129, mids[c / 129][c % 129]);
    } else {
        for (int c = 0; c < 387; ++c)
            costs[c / 129][c % 129] = candidate_cost(T, ids, n, c / 129, mids[c / 129][c % 129]);
    }
    /* bvh.go:163-165,202-206 */
    float best_cost = ORC_INF; int best_axis = -1; float best_split = -1.0f;
    for (int axis = 0; axis < 3; ++axis)
        for (int i = 0; i <= 128; ++i)
            if (costs[axis][i] < best_cost) {
                best_cost = costs[axis][i]; best_axis = axis; best_split = mids[axis][i];
            }
    /* bvh.go:210-215 */
    if (best_cost >= parent_cost) {
        b->is_leaf = 1; b->tris = ids; b->ntris = n;
        return b;
    }
    /* bvh.go:231-246 getSplit: stable partition + Box.Grow bounds */
    uint32_t *lids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *rids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    fauxbox L, R; memset(&L, 0, sizeof L); memset(&R, 0, sizeof R);
    for (uint64_t k = 0; k < n; ++k) {
        const orc_tri *t = &T[ids[k]];
        if (tri_centroid_axis(t, best_axis) < best_split) { lids[L.count] = ids[k]; faux_grow(&L, t); }
        else { rids[R.count] = ids[k]; faux_grow(&R, t); }
    }
    free(ids);
    /* bvh.go:219-220 */
    b->child[0] = build_rec(T, lids, (uint64_t)L.count, L.x1, L.x2, L.y1, L.y2, L.z1, L.z2,
                            threshold, depth - 1, best_cost);
    b->child[1] = build_rec(T, rids, (uint64_t)R.count, R.x1, R.x2, R.y1, R.y2, R.z1, R.z2,
                            threshold, depth - 1, best_cost);
    return b;
}

static void count_rec(const box *b, uint64_t *nn, uint64_t *nt) {
    *nn += 1;
    if (b->is_leaf) { *nt += b->ntris; return; }
    count_rec(b->child[0], nn, nt); count_rec(b->child[1], nn, nt);
}

/* linear_bvh.go:77-106 */
static void convert_rec(box *root, orc_bvh *o) {
    orc_node node; memset(&node, 0, sizeof node);
    node.is_leaf = (uint8_t)root->is_leaf;
    node.minb.x = root->x1; node.minb.y = root->y1; node.minb.z = root->z1;
    node.maxb.x = root->x2; node.maxb.y = root->y2; node.maxb.z = root->z2;
    if (root->is_leaf) {
        node.tri_offset = (uint32_t)o->n_tri_ids;
        memcpy(o->tri_ids + o->n_tri_ids, root->tris, sizeof(uint32_t) * root->ntris);
        o->n_tri_ids += root->ntris;
        node.tri_count = (uint32_t)root->ntris;
    }
    uint64_t ptr = o->n_nodes;
    o->nodes[o->n_nodes++] = node;
    if (!root->is_leaf) {
        convert_rec(root->child[0], o);
        o->nodes[ptr].second_child = (uint32_t)o->n_nodes;
        convert_rec(root->child[1], o);
    }
    free(root->tris);
    free(root);
}

/* bvh.go:248-275 triangle records */
static orc_tri *make_tris(const float *verts, const uint32_t *idx, uint64_t ntris) {
    orc_tri *T = (orc_tri *)malloc(sizeof(orc_tri) * (ntris ? ntris : 1));
    for (uint64_t k = 0; k < ntris; ++k) {
        const float *pa = verts + 3 * (uint64_t)idx[3 * k], *pb = verts + 3 * (uint64_t)idx[3 * k + 1],
                    *pc = verts + 3 * (uint64_t)idx[3 * k + 2];
        v3 a = { pa[0], pa[1], pa[2] }, b = { pb[0], pb[1], pb[2] }, c = { pc[0], pc[1], pc[2] };
        orc_tri *t = &T[k];
        t->A = a; t->B = b; t->C = c;
        /* a.Add(b).Add(c).Scale(1.0 / 3.0): the constant is float32(1/3) */
        t->centroid = v3_scale(v3_add(v3_add(a, b), c), (float)(1.0 / 3.0));
        t->index = (int64_t)(3 * k);
        t->minx = gominf(gominf(a.x, b.x), c.x); t->maxx = gomaxf(gomaxf(a.x, b.x), c.x);
        t->miny = gominf(gominf(a.y, b.y), c.y); t->maxy = gomaxf(gomaxf(a.y, b.y), c.y);
        t->minz = gominf(gominf(a.z, b.z), c.z); t->maxz = gomaxf(gomaxf(a.z, b.z), c.z);
    }
    return T;
}

/* BuildBVH (bvh.go:248) followed by ConstructLinearBVH (linear_bvh.go:108) */
orc_bvh *orc_build_bvh(const float *verts, uint64_t nverts, const uint32_t *idx, uint64_t ntris,
                       const float bounds[6], int threshold, int depth) {
    (void)nverts;
    orc_bvh *o = (orc_bvh *)calloc(1, sizeof(orc_bvh));
    o->tris = make_tris(verts, idx, ntris); o->n_tris = ntris;
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (ntris ? ntris : 1));
    for (uint64_t k = 0; k < ntris; ++k) ids[k] = (uint32_t)k;
    box *root = build_rec(o->tris, ids, ntris, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4],
                          bounds[5], threshold, depth, ORC_INF);
    uint64_t nn = 0, nt = 0; count_rec(root, &nn, &nt);
    o->nodes = (orc_node *)malloc(sizeof(orc_node) * nn);
    o->tri_ids = (uint32_t *)malloc(sizeof(uint32_t) * (nt ? nt : 1));
    convert_rec(root, o);
    return o;
}

/* Wrap a tree that is already in the reference's LinearBVH form (arrays as exported
 * by orc_bvh_export or by the product's ptmi_get_linear_bvh).  Used when the literal
 * builder would take too long (>= 1 M triangles). */
orc_bvh *orc_bvh_import(const float *verts, uint64_t nverts, const uint32_t *idx, uint64_t ntris,
                        uint64_t n_nodes, const float *node_bounds, const uint8_t *is_leaf,
                        const uint32_t *tri_off, const uint32_t *tri_cnt, const uint32_t *second,
                        uint64_t n_tri_ids, const uint32_t *tri_ids) {
    (void)nverts;
    orc_bvh *o = (orc_bvh *)calloc(1, sizeof(orc_bvh));
    o->tris = make_tris(verts, idx, ntris); o->n_tris = ntris;
    o->nodes = (orc_node *)malloc(sizeof(orc_node) * (n_nodes ? n_nodes : 1)); o->n_nodes = n_nodes;
    for (uint64_t i = 0; i < n_nodes; ++i) {
        orc_node *n = &o->nodes[i];
        n->minb.x = node_bounds[6 * i + 0]; n->minb.y = node_bounds[6 * i + 1]; n->minb.z = node_bounds[6 * i + 2];
        n->maxb.x = node_bounds[6 * i + 3]; n->maxb.y = node_bounds[6 * i + 4]; n->maxb.z = node_bounds[6 * i + 5];
        n->is_leaf = is_leaf[i]; n->tri_offset = tri_off[i]; n->tri_count = tri_cnt[i];
        n->second_child = second[i];
    }
    o->tri_ids = (uint32_t *)malloc(sizeof(uint32_t) * (n_tri_ids ? n_tri_ids : 1));
    memcpy(o->tri_ids, tri_ids, sizeof(uint32_t) * n_tri_ids); o->n_tri_ids = n_tri_ids;
    return o;
}

void orc_free_bvh(orc_bvh *o) {
    if (!o) return;
    free(o->nodes); free(o->tri_ids); free(o->tris); free(o);
}
uint64_t orc_bvh_num_nodes(const orc_bvh *o) { return o->n_nodes; }
uint64_t orc_bvh_num_tri_ids(const orc_bvh *o) { return o->n_tri_ids; }

/* node_bounds: 6 floats per node (min xyz, max xyz) */
void orc_bvh_export(const orc_bvh *o, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_off,
                    uint32_t *tri_cnt, uint32_t *second, uint32_t *tri_ids) {
    for (uint64_t i = 0; i < o->n_nodes; ++i) {
        const orc_node *n = &o->nodes[i];
        node_bounds[6 * i + 0] = n->minb.x; node_bounds[6 * i + 1] = n->minb.y; node_bounds[6 * i + 2] = n->minb.z;
        node_bounds[6 * i + 3] = n->maxb.x; node_bounds[6 * i + 4] = n->maxb.y; node_bounds[6 * i + 5] = n->maxb.z;
        is_leaf[i] = n->is_leaf; tri_off[i] = n->tri_offset; tri_cnt[i] = n->tri_count; second[i] = n->second_child;
    }
    memcpy(tri_ids, o->tri_ids, sizeof(uint32_t) * o->n_tri_ids);
}

/* -------------------------------------------------------------- traversal */

/* linear_bvh.go:23-70 */
static inline float intersect_aabb(const orc_node *l, v3 o, v3 d, float step) {
    v3 inv = v3_inverse(d);
    v3 lb = v3_cmul(v3_sub(l->minb, o), inv);
    v3 rb = v3_cmul(v3_sub(l->maxb, o), inv);
    float t1 = lb.x, t2 = rb.x, tmp;
    if (t1 > t2) { tmp = t1; t1 = t2; t2 = tmp; }
    float tmin = t1, tmax = t2;
    t1 = lb.y; t2 = rb.y;
    if (t1 > t2) { tmp = t1; t1 = t2; t2 = tmp; }
    tmin = gomaxf(tmin, t1);
    tmax = gominf(tmax, t2);
    t1 = lb.z; t2 = rb.z;
    if (t1 > t2) { tmp = t1; t1 = t2; t2 = tmp; }
    tmin = gomaxf(tmin, t1);
    tmax = gominf(tmax, t2);
    if (tmin > gominf(tmax, step) || tmax < 0.0f) return ORC_INF;
    return gomaxf(0.0f, tmin);
}

/* utils.go:63-125 */
static inline int intersect_segment_triangle(v3 o, v3 d, float step, v3 A, v3 B, v3 C, float *t_out) {
    const float EPS = (float)1e-6;
    const float ONE_PLUS_EPS = (float)1.000001;      /* untyped const 1.0+EPSILON -> float32 */
    v3 e1 = v3_sub(B, A), e2 = v3_sub(C, A);
    v3 pvec = { d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x };
    float det = v3_dot(e1, pvec);
    if (fabsf(det) < EPS) return 0;
    float inv = 1.0f / det;
    v3 tvec = v3_sub(o, A);
    float u = v3_dot(tvec, pvec) * inv;
    if (u < -EPS || u > ONE_PLUS_EPS) return 0;
    v3 qvec = { tvec.y * e1.z - tvec.z * e1.y, tvec.z * e1.x - tvec.x * e1.z, tvec.x * e1.y - tvec.y * e1.x };
    float v = v3_dot(d, qvec) * inv;
    if (v < -EPS || u + v > ONE_PLUS_EPS) return 0;
    float t = v3_dot(e2, qvec) * inv;
    if (t <= EPS || t > step) return 0;
    *t_out = t;
    return 1;
}

/* linear_bvh.go:173-200 */
static inline int fast_shadow_triangle(v3 o, v3 d, float tmax, v3 v0, v3 v1, v3 v2) {
    v3 e1 = v3_sub(v1, v0), e2 = v3_sub(v2, v0);
    v3 h = v3_cross(d, e2);
    float a = (e1.x * h.x + e1.y * h.y) + e1.z * h.z;
    if (fabsf(a) < (float)0.00001) return 0;
    v3 s = v3_sub(o, v0);
    float u = ((s.x * h.x + s.y * h.y) + s.z * h.z) / a;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3_cross(s, e1);
    float v = ((d.x * q.x + d.y * q.y) + d.z * q.z) / a;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = ((e2.x * q.x + e2.y * q.y) + e2.z * q.z) / a;
    return t > (float)0.00001 && t < tmax;
}

/* linear_bvh.go:116-170 */
int orc_check_intersection(const orc_bvh *b, v3 o, v3 d, float step, float *t_out, int64_t *tri_id,
                           orc_counters *c) {
    uint32_t stack[64];
    stack[0] = 0;
    int nptr = 1;
    float best_t = step;
    int64_t best = -1;
    while (nptr > 0) {
        uint32_t ptr = stack[nptr - 1];
        const orc_node *node = &b->nodes[ptr];
        nptr--;
        if (node->is_leaf) {
            if (c) c->leaf_visits++;
            for (uint32_t i = 0; i < node->tri_count; ++i) {
                uint32_t id = b->tri_ids[node->tri_offset + i];
                const orc_tri *tri = &b->tris[id];
                float t;
                if (c) c->tri_tests++;
                int hit = intersect_segment_triangle(o, d, best_t, tri->A, tri->B, tri->C, &t);
                if (hit && t < best_t && t > 0.0f) { best_t = t; best = (int64_t)id; }
            }
        } else {
            if (c) { c->inner_visits++; c->box_tests++; }
            float d1 = intersect_aabb(&b->nodes[ptr + 1], o, d, best_t);
            if (node->second_child == 0) {
                if (d1 < best_t) stack[nptr++] = ptr + 1;
            } else {
                if (c) c->box_tests++;
                float d2 = intersect_aabb(&b->nodes[node->second_child], o, d, best_t);
                uint32_t i = ptr + 1, j = node->second_child;
                if (d1 > d2) { uint32_t ti = i; i = j; j = ti; float td = d1; d1 = d2; d2 = td; }
                if (d1 == ORC_INF) continue;
                if (d2 < best_t) stack[nptr++] = j;
                stack[nptr++] = i;
            }
        }
    }
    if (best < 0) { *t_out = 0.0f; *tri_id = -1; return 0; }
    *t_out = best_t; *tri_id = best;
    return 1;
}

/* linear_bvh.go:202-250 */
int orc_quick_check_intersection(const orc_bvh *b, v3 o, v3 d, float step, orc_counters *c) {
    uint32_t stack[64];
    stack[0] = 0;
    int nptr = 1;
    while (nptr > 0) {
        uint32_t ptr = stack[nptr - 1];
        const orc_node *node = &b->nodes[ptr];
        nptr--;
        if (node->is_leaf) {
            if (c) c->leaf_visits++;
            for (uint32_t i = 0; i < node->tri_count; ++i) {
                const orc_tri *tri = &b->tris[b->tri_ids[node->tri_offset + i]];
                if (c) c->tri_tests++;
                if (fast_shadow_triangle(o, d, step, tri->A, tri->B, tri->C)) return 1;
            }
        } else {
            if (c) { c->inner_visits++; c->box_tests += 2; }
            float d1 = intersect_aabb(&b->nodes[ptr + 1], o, d, step);
            float d2 = intersect_aabb(&b->nodes[node->second_child], o, d, step);
            if (d1 < step && d2 < step) {
                if (d1 < d2) { stack[nptr++] = node->second_child; stack[nptr++] = ptr + 1; }
                else { stack[nptr++] = ptr + 1; stack[nptr++] = node->second_child; }
            } else if (d1 < step) {
                stack[nptr++] = ptr + 1;
            } else if (d2 < step) {
                stack[nptr++] = node->second_child;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------- batch entry points */

/* o, d: n x 3 floats; tmax: n floats.  prim: tri.Index/3 or -1; t: best_t or 0.
 * counters (may be NULL): 4 x uint64 summed over the batch. */
void orc_closest_hit_batch(const orc_bvh *b, const float *o, const float *d, const float *tmax, uint64_t n,
                           int32_t *prim, float *t, uint64_t *counters, int nthreads) {
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1024) reduction(+ : c0, c1, c2, c3)
    for (int64_t r = 0; r < (int64_t)n; ++r) {
        v3 ro = { o[3 * r], o[3 * r + 1], o[3 * r + 2] }, rd = { d[3 * r], d[3 * r + 1], d[3 * r + 2] };
        orc_counters c; memset(&c, 0, sizeof c);
        float tt; int64_t id;
        orc_check_intersection(b, ro, rd, tmax[r], &tt, &id, counters ? &c : NULL);
        prim[r] = (int32_t)id; t[r] = tt;
        c0 += c.inner_visits; c1 += c.box_tests; c2 += c.tri_tests; c3 += c.leaf_visits;
    }
    if (counters) { counters[0] = c0; counters[1] = c1; counters[2] = c2; counters[3] = c3; }
}

void orc_any_hit_batch(const orc_bvh *b, const float *o, const float *d, const float *tmax, uint64_t n,
                       uint8_t *occluded, uint64_t *counters, int nthreads) {
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1024) reduction(+ : c0, c1, c2, c3)
    for (int64_t r = 0; r < (int64_t)n; ++r) {
        v3 ro = { o[3 * r], o[3 * r + 1], o[3 * r + 2] }, rd = { d[3 * r], d[3 * r + 1], d[3 * r + 2] };
        orc_counters c; memset(&c, 0, sizeof c);
        occluded[r] = (uint8_t)orc_quick_check_intersection(b, ro, rd, tmax[r], counters ? &c : NULL);
        c0 += c.inner_visits; c1 += c.box_tests; c2 += c.tri_tests; c3 += c.leaf_visits;
    }
    if (counters) { counters[0] = c0; counters[1] = c1; counters[2] = c2; counters[3] = c3; }
}

/* Single-function probes used by the unit tests (edge cases of utils.go:63 / linear_bvh.go:23,173). */
int orc_probe_segment_triangle(const float *o, const float *d, float step, const float *abc, float *t) {
    v3 ro = { o[0], o[1], o[2] }, rd = { d[0], d[1], d[2] };
    v3 A = { abc[0], abc[1], abc[2] }, B = { abc[3], abc[4], abc[5] }, C = { abc[6], abc[7], abc[8] };
    float tt = 0.0f;
    int h = intersect_segment_triangle(ro, rd, step, A, B, C, &tt);
    *t = h ? tt : 0.0f;
    return h;
}
int orc_probe_shadow_triangle(const float *o, const float *d, float tmax, const float *abc) {
    v3 ro = { o[0], o[1], o[2] }, rd = { d[0], d[1], d[2] };
    v3 A = { abc[0], abc[1], abc[2] }, B = { abc[3], abc[4], abc[5] }, C = { abc[6], abc[7], abc[8] };
    return fast_shadow_triangle(ro, rd, tmax, A, B, C);
}
float orc_probe_aabb(const float *minb, const float *maxb, const float *o, const float *d, float step) {
    orc_node n; memset(&n, 0, sizeof n);
    n.minb.x = minb[0]; n.minb.y = minb[1]; n.minb.z = minb[2];
    n.maxb.x = maxb[0]; n.maxb.y = maxb[1]; n.maxb.z = maxb[2];
    v3 ro = { o[0], o[1], o[2] }, rd = { d[0], d[1], d[2] };
    return intersect_aabb(&n, ro, rd, step);
}
