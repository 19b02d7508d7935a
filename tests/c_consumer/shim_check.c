/* A plain-C99 consumer of include/ptmi.h: the call sequence the cgo shim of INTEGRATION.md makes, from the
 * language cgo compiles the header with.  `shim_check host` needs no GPU (header is valid C, struct sizes are the
 * documented ones, host builder and error paths work); `shim_check gpu` runs build -> closest hit -> any hit ->
 * scene upload -> one small render on device 0 and checks hand-derived answers.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptmi.h"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); ++failures; } \
    } while (0)

/* a unit quad in the plane z = 0 made of two triangles, and a second quad above it at z = 2 */
static const float verts[8 * 3] = {0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0, 0, 0, 2, 1, 0, 2, 1, 1, 2, 0, 1, 2};
static const uint32_t tris[4 * 3] = {0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7};
static const float bounds[6] = {-1000, 1000, -1000, 1000, -1000, 1000};   /* main.go:804 */

static void host_checks(void) {
    ptmi_host_bvh *hb = NULL;
    uint64_t n_nodes = 0, n_ids = 0;
    PtBvhStats st;
    CHECK(sizeof(PtMaterial) == 64);
    CHECK(sizeof(PtLight) == 48);
    CHECK(sizeof(PtSky) == 64);
    CHECK(sizeof(PtBlackHole) == 32);
    CHECK(sizeof(PtCamera) == 52);
    CHECK(sizeof(PtParams) == 32);
    CHECK(sizeof(PtCounters) == 112);
    CHECK(strstr(ptmi_version(), "ptmi") != NULL);
    CHECK(ptmi_create(0, NULL) == PTMI_ERR_INVALID);
    CHECK(strlen(ptmi_last_error(NULL)) > 0);
    CHECK(ptmi_host_bvh_build(verts, 8, tris, 4, bounds, 4, 42, &hb) == PTMI_OK);
    CHECK(hb != NULL);
    CHECK(ptmi_host_bvh_sizes(hb, &n_nodes, &n_ids) == PTMI_OK);
    CHECK(n_nodes >= 1 && n_ids == 4);
    CHECK(ptmi_host_bvh_stats(hb, &st) == PTMI_OK);
    CHECK(st.total_triangles == 4 && st.linear_nodes == (int64_t)n_nodes);
    ptmi_host_bvh_free(hb);
    hb = NULL;
    {   /* a triangle index beyond nverts is an error code, not a crash (the reference would panic) */
        const uint32_t bad[3] = {0, 1, 99};
        CHECK(ptmi_host_bvh_build(verts, 8, bad, 1, bounds, 4, 42, &hb) == PTMI_ERR_INVALID);
    }
}

static void gpu_checks(void) {
    ptmi_ctx *ctx = NULL;
    int rc = ptmi_create(0, &ctx);
    if (rc != PTMI_OK) { fprintf(stderr, "ptmi_create: %d %s\n", rc, ptmi_last_error(NULL)); ++failures; return; }
    {   /* queries before a tree exists */
        float o[3] = {0, 0, 1}, d[3] = {0, 0, -1}, tm = 10, t = -1;
        int32_t p = 7;
        CHECK(ptmi_closest_hit(ctx, o, d, &tm, 1, &p, &t) == PTMI_ERR_NO_BVH);
    }
    CHECK(ptmi_build_bvh(ctx, verts, 8, tris, 4, bounds, 4, 42) == PTMI_OK);
    {
        /* ray 0: straight down from z = 1 onto triangle 0 (x > y half of the lower quad): t = 1 exactly
         * ray 1: straight down from z = 3 -- hits the upper quad first (triangle 3: x < y half), t = 1
         * ray 2: same as ray 1 but range 0.5: miss (prim -1, t 0: the reference returns (false, 0, nil))
         * ray 3: beside the quads: miss */
        const float o[4 * 3] = {0.75f, 0.25f, 1, 0.25f, 0.75f, 3, 0.25f, 0.75f, 3, 5, 5, 3};
        const float d[4 * 3] = {0, 0, -1, 0, 0, -1, 0, 0, -1, 0, 0, -1};
        const float tmax[4] = {1000, 1000, 0.5f, 1000};
        int32_t prim[4] = {9, 9, 9, 9};
        float t[4] = {9, 9, 9, 9};
        uint8_t occ[4] = {9, 9, 9, 9};
        CHECK(ptmi_closest_hit(ctx, o, d, tmax, 4, prim, t) == PTMI_OK);
        CHECK(prim[0] == 0 && t[0] == 1.0f);
        CHECK(prim[1] == 3 && t[1] == 1.0f);
        CHECK(prim[2] == -1 && t[2] == 0.0f);
        CHECK(prim[3] == -1 && t[3] == 0.0f);
        CHECK(ptmi_any_hit(ctx, o, d, tmax, 4, occ) == PTMI_OK);
        CHECK(occ[0] == 1 && occ[1] == 1 && occ[2] == 0 && occ[3] == 0);
        CHECK(ptmi_closest_hit(ctx, o, d, tmax, 0, prim, t) == PTMI_OK);          /* an empty batch is fine */
        CHECK(ptmi_closest_hit(ctx, NULL, d, tmax, 4, prim, t) == PTMI_ERR_INVALID);
    }
    {   /* scene: one grey diffuse material, solid white sky, no lights -> a small render */
        PtMaterial m;
        PtSky sky;
        PtCamera cam;
        PtParams prm;
        uint32_t tri_mat[4] = {0, 0, 0, 0};
        float normals[4 * 9];
        float accum[8 * 8 * 3];
        uint32_t counts[8 * 8];
        int i, lit = 0;
        memset(&m, 0, sizeof m);
        m.kd[0] = m.kd[1] = m.kd[2] = 0.5f;
        m.ni = 1.0f;
        memset(&sky, 0, sizeof sky);
        sky.kind = PTMI_SKY_SOLID;
        sky.horizon[0] = sky.horizon[1] = sky.horizon[2] = 1.0f;
        sky.intensity = 1.0f;
        for (i = 0; i < 12; ++i) { normals[3 * i] = 0; normals[3 * i + 1] = 0; normals[3 * i + 2] = 1; }
        CHECK(ptmi_render(ctx, &cam, &prm, 0, 0, 8, 8, 0, 1, accum, counts) == PTMI_ERR_NO_SCENE);
        CHECK(ptmi_upload_scene(ctx, &m, 1, tri_mat, normals, NULL, 0, &sky, NULL, NULL) == PTMI_OK);
        memset(&cam, 0, sizeof cam);
        cam.position[0] = 0.5f; cam.position[1] = 0.5f; cam.position[2] = 6.0f;
        cam.forward[2] = -1.0f; cam.right[0] = 1.0f; cam.up[1] = 1.0f;
        cam.frustum_distance = 1.0f;
        memset(&prm, 0, sizeof prm);
        prm.width = 8; prm.height = 8; prm.bounces = 2; prm.scatter_rays = 1; prm.max_steps = 1;
        prm.ambient = 0.0f; prm.step_size = 1000.0f; prm.seed = 3;
        memset(accum, 0, sizeof accum);
        memset(counts, 0, sizeof counts);
        rc = ptmi_render(ctx, &cam, &prm, 0, 0, 8, 8, 0, 4, accum, counts);
        if (rc != PTMI_OK) fprintf(stderr, "ptmi_render: %d %s\n", rc, ptmi_last_error(ctx));
        CHECK(rc == PTMI_OK);
        for (i = 0; i < 64; ++i) {
            CHECK(counts[i] == 4);
            CHECK(isfinite(accum[3 * i]) && accum[3 * i] >= 0.0f);
            if (accum[3 * i] > 0.0f) ++lit;
        }
        CHECK(lit > 0);
        /* a second call continues the running sums (Pixel.AddSample): counts go 4 -> 6 */
        CHECK(ptmi_render(ctx, &cam, &prm, 0, 0, 8, 8, 4, 2, accum, counts) == PTMI_OK);
        CHECK(counts[0] == 6 && counts[63] == 6);
        CHECK(ptmi_render(ctx, &cam, &prm, 4, 4, 8, 8, 0, 1, accum, counts) == PTMI_ERR_INVALID);   /* outside the image */
        {   /* the path pool: in place after a render, given back when the limit drops under it, same sums afterwards */
            uint64_t paths = 0, bytes = 0;
            float accum2[64 * 3] = {0};
            uint32_t counts2[64] = {0};
            CHECK(ptmi_pool_info(ctx, &paths, &bytes) == PTMI_OK && paths >= 64 && bytes > 0);
            CHECK(ptmi_set_pool_limit(NULL, 0, 0) == PTMI_ERR_INVALID);
            CHECK(ptmi_set_pool_limit(ctx, 1, 1) == PTMI_OK);                 /* clamped to the smallest pool */
            CHECK(ptmi_pool_info(ctx, &paths, &bytes) == PTMI_OK);
            CHECK(ptmi_render(ctx, &cam, &prm, 0, 0, 8, 8, 0, 6, accum2, counts2) == PTMI_OK);
            CHECK(counts2[0] == 6 && memcmp(accum, accum2, sizeof accum2) == 0);
        }
        {   /* the march's short square root / reciprocal against the compiler's, every float of their ranges */
            uint64_t bad = 1;
            CHECK(ptmi_selftest_math(NULL, &bad) == PTMI_ERR_INVALID && ptmi_selftest_math(ctx, NULL) == PTMI_ERR_INVALID);
            CHECK(ptmi_selftest_math(ctx, &bad) == PTMI_OK && bad == 0);
        }
        /* ptmi_render_tiles_stats_dev takes device buffers (the Python tests drive it); its argument checks from here */
        CHECK(ptmi_render_tiles_stats_dev(ctx, &cam, &prm, 0, NULL, 0, 1, accum, counts, NULL, NULL) == PTMI_ERR_INVALID);
        CHECK(ptmi_render_adaptive_dev(ctx, &cam, &prm, 0, 64, accum, counts, accum, NULL) == PTMI_ERR_INVALID);      /* batch of 0 */
        CHECK(ptmi_render_adaptive_dev(ctx, &cam, &prm, 32, 64, accum, counts, NULL, NULL) == PTMI_ERR_INVALID);
    }
    ptmi_destroy(ctx);
}

int main(int argc, char **argv) {
    host_checks();
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) gpu_checks();
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("shim_check %s ok\n", argc > 1 ? argv[1] : "host");
    return 0;
}
