/*
 * oracle/orc_bvh.h -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
 * PARITY UNPINNED (no reference fixtures exist).
 *
 * CPU restatement of the reference's BVH builder, flattener and the two
 * traversals.  Reference: bvh.go:8-107,148-275; linear_bvh.go:9-250;
 * utils.go:63-125.
 */
#ifndef ORC_BVH_H
#define ORC_BVH_H

#include "orc_common.h"

/* bvh.go:8-17 (only the fields the hot path reads) */
typedef struct {
    v3 A, B, C;
    v3 centroid;
    int64_t index;              /* Index = 3 * triangle number (bvh.go:257) */
    float minx, maxx, miny, maxy, minz, maxz;
} orc_tri;

/* linear_bvh.go:11-21 */
typedef struct {
    v3 minb, maxb;
    uint8_t is_leaf;
    uint32_t tri_offset, tri_count, second_child;
} orc_node;

/* linear_bvh.go:72-75; Triangles []*BVHTriangle becomes an index list */
typedef struct {
    orc_node *nodes;   uint64_t n_nodes;
    uint32_t *tri_ids; uint64_t n_tri_ids;   /* position -> triangle number */
    orc_tri  *tris;    uint64_t n_tris;      /* triangle number -> record   */
} orc_bvh;

typedef struct {
    uint64_t inner_visits;   /* inner nodes popped                      */
    uint64_t box_tests;      /* intersectAABB calls                     */
    uint64_t tri_tests;      /* triangle tests                          */
    uint64_t leaf_visits;    /* leaves popped                           */
} orc_counters;

/* linear_bvh.go:116-170.  Returns 1 on hit; *t = best_t, *tri_id = tri.Index/3 */
int orc_check_intersection(const orc_bvh *b, v3 o, v3 d, float step, float *t, int64_t *tri_id,
                           orc_counters *c);
/* linear_bvh.go:202-250 */
int orc_quick_check_intersection(const orc_bvh *b, v3 o, v3 d, float step, orc_counters *c);

#endif
