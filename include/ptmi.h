/*
 * ptmi.h -- C ABI of the MI355X-native path-tracing backend (libptmi.so).
 *
 * This is the drop-in boundary for the hot path of 0Xero7/pathtrace: the seam between the
 * reference's tile scheduler (main.go:832-899, tile.go, renderer.go) and its integrator /
 * BVH traversal (tracer.go, linear_bvh.go, utils.go).  Each entry point cites the reference
 * interface it replaces.  INTEGRATION.md shows the cgo shim a maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; all entry points return 0 on success and a negative
 *     PTMI_ERR_* code on failure; the text of the last error is ptmi_last_error(ctx)
 *     (ctx may be NULL for errors raised before a context exists).  Nothing aborts or throws
 *     across this boundary (the reference panics instead: skybox.go:55-62, obj_loader.go:35-45).
 *   - caller-owned buffers are only borrowed for the duration of a call;
 *     "upload"/"build" calls copy what they need into GPU memory owned by the context.
 *   - one ptmi_ctx per GPU; calls on one ctx must be serialised by the caller; different
 *     contexts may be driven from different OS threads.  Every entry point selects the
 *     context's device itself (hipSetDevice), so Go's goroutine->thread migration is harmless.
 *   - entry points ending in _dev take DEVICE pointers on the context's GPU and enqueue on
 *     `stream` (a hipStream_t passed as void*, NULL = default stream) without synchronising.
 *     The batch queries (ptmi_closest_hit_dev, ptmi_any_hit_dev) may be issued on different
 *     streams of one context and overlap on the device: every launch has its own ray cursor
 *     (a ring of 64, so at most 64 such launches of one context may be in flight at once).
 *   - a new tree (ptmi_build_bvh, ptmi_upload_bvh, ptmi_upload_host_bvh) waits for the device
 *     and invalidates the uploaded scene: call ptmi_upload_scene again before rendering.
 *   - vectors are packed xyz float32 triples (the reference's Vec3, vec3.go:9-11).
 */
#ifndef PTMI_H
#define PTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptmi_ctx ptmi_ctx;

enum {
    PTMI_OK = 0,
    PTMI_ERR_INVALID = -1,     /* bad argument (null pointer, index out of range, ...)        */
    PTMI_ERR_NO_DEVICE = -2,   /* no usable GPU / HIP runtime failure at create               */
    PTMI_ERR_HIP = -3,         /* a HIP call failed; text in ptmi_last_error                   */
    PTMI_ERR_NO_BVH = -4,      /* query before ptmi_build_bvh / ptmi_upload_bvh               */
    PTMI_ERR_DEPTH = -5,       /* tree deeper than the traversal stack (reference: 64 entries) */
    PTMI_ERR_NO_SCENE = -6,    /* ptmi_render before ptmi_upload_scene                         */
    PTMI_ERR_OOM = -7,
    PTMI_ERR_COMM = -8         /* RCCL unavailable or a collective failed; text in ptmi_last_error */
};

/* ------------------------------------------------------------------ context */

/* No reference counterpart (the reference is one process with shared memory); creates the
 * per-GPU state that stands in for the globals of main.go. */
int ptmi_create(int device, ptmi_ctx **out);
void ptmi_destroy(ptmi_ctx *ctx);
const char *ptmi_last_error(const ptmi_ctx *ctx);
const char *ptmi_version(void);

/* ------------------------------------------------------------------ BVH */

/* Replaces BuildBVH (bvh.go:248-275) + ConstructLinearBVH (linear_bvh.go:108-112) as called at
 * main.go:804-810.  verts_xyz: nverts*3 floats; tri_idx: ntris*3 vertex indices (the reference's
 * `tris []int`); bounds = {x1,x2,y1,y2,z1,z2} root box (the reference passes +-1000);
 * threshold / depth as bvh.go:148 (reference: 4, 42).  The tree produced is bit-identical in
 * topology, bounds and leaf order to the reference's, then packed for the GPU and uploaded. */
int ptmi_build_bvh(ptmi_ctx *ctx, const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx,
                   uint64_t ntris, const float bounds[6], int threshold, int depth);

/* Host-only form of the builder: the same code ptmi_build_bvh runs, without touching a GPU
 * (start-up work in the reference too: main.go:803-810 runs before any worker starts).  The
 * handle keeps copies of the vertex / index arrays; ptmi_upload_host_bvh packs and uploads it. */
typedef struct ptmi_host_bvh ptmi_host_bvh;
int ptmi_host_bvh_build(const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx, uint64_t ntris,
                        const float bounds[6], int threshold, int depth, ptmi_host_bvh **out);
void ptmi_host_bvh_free(ptmi_host_bvh *bvh);
int ptmi_host_bvh_sizes(const ptmi_host_bvh *bvh, uint64_t *n_nodes, uint64_t *n_tri_ids);
int ptmi_host_bvh_get(const ptmi_host_bvh *bvh, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_offset,
                      uint32_t *tri_count, uint32_t *second_child, uint32_t *tri_ids);

/* Upload a tree that is already in the reference's LinearBVH form (linear_bvh.go:11-21,72-75):
 * node_bounds = 6 floats per node (min xyz, max xyz); tri_ids = the Triangles slice as triangle
 * numbers (tri.Index/3).  Replaces nothing in the reference; lets a Go host keep its own builder. */
int ptmi_upload_bvh(ptmi_ctx *ctx, const float *verts_xyz, uint64_t nverts, const uint32_t *tri_idx,
                    uint64_t ntris, uint64_t n_nodes, const float *node_bounds, const uint8_t *is_leaf,
                    const uint32_t *tri_offset, const uint32_t *tri_count, const uint32_t *second_child,
                    uint64_t n_tri_ids, const uint32_t *tri_ids);

/* Read the tree back in the reference's LinearBVH form (sizes first, then the arrays). */
int ptmi_bvh_sizes(ptmi_ctx *ctx, uint64_t *n_nodes, uint64_t *n_tri_ids);
int ptmi_get_linear_bvh(ptmi_ctx *ctx, float *node_bounds, uint8_t *is_leaf, uint32_t *tri_offset,
                        uint32_t *tri_count, uint32_t *second_child, uint32_t *tri_ids);

/* Replaces Box.GetStats(1) (bvh.go:109-146) as printed at main.go:809.  total_nodes follows the
 * reference (it counts leaves); linear_nodes is len(LinearBVH.Nodes). */
typedef struct {
    int64_t max_depth, max_tris, min_tris, total_leafs, total_nodes, total_triangles;
    int64_t linear_nodes, packed_inner_nodes;
    double build_ms;
} PtBvhStats;
int ptmi_bvh_stats(ptmi_ctx *ctx, PtBvhStats *out);
int ptmi_host_bvh_stats(const ptmi_host_bvh *bvh, PtBvhStats *out);
int ptmi_upload_host_bvh(ptmi_ctx *ctx, const ptmi_host_bvh *bvh);

/* ------------------------------------------------------------------ ray queries */

/* Replaces (*LinearBVH).CheckIntersection (linear_bvh.go:116-170) for a batch of n rays.
 * o_xyz, d_xyz: n*3 floats; tmax: n floats (the reference's stepSize); outputs prim_id[n] =
 * tri.Index/3 of the hit triangle or -1, t[n] = best_t or 0 on a miss (reference returns
 * (false, 0, nil)).  Host buffers; synchronous. */
int ptmi_closest_hit(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                     int32_t *prim_id, float *t);
/* Same, device buffers, asynchronous on `stream`. */
int ptmi_closest_hit_dev(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax,
                         uint64_t n, int32_t *prim_id, float *t, void *stream);

/* No counterpart in the reference.  ptmi_closest_hit_dev for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same kernels, same
 * results bit for bit; the one piece of scratch the persistent kernel needs (its ray cursor) lives in a 256-byte
 * __device__ array of the library instead of the context's hipMalloc'd block, and the call makes no hipMalloc /
 * hipFree: one hipMemsetAsync of that array and one kernel launch on `stream`.  n may be at most
 * PTMI_NOALLOC_MAX_RAYS; a larger n returns PTMI_ERR_INVALID.  Every call on a GPU uses that GPU's one array
 * (whatever the context or stream): calls that overlap on one GPU race -- order them on one stream or with events. */
#define PTMI_NOALLOC_MAX_RAYS (1ull << 32)
int ptmi_closest_hit_dev_noalloc(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax,
                                 uint64_t n, int32_t *prim_id, float *t, void *stream);

/* Replaces (*LinearBVH).QuickCheckIntersection (linear_bvh.go:202-250): occluded[n] = 1 if any
 * triangle is hit in (1e-5, tmax). */
int ptmi_any_hit(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                 uint8_t *occluded);
int ptmi_any_hit_dev(ptmi_ctx *ctx, const float *o_xyz, const float *d_xyz, const float *tmax, uint64_t n,
                     uint8_t *occluded, void *stream);

/* Work counters.  The reference only counts diffuse shading events (raysTraced, tracer.go:12,666);
 * the traversal counters here define the algorithmic bytes of the roofline.
 * Counting is off by default (it selects instrumented kernel variants: of the batch kernels and of the
 * integrator's extend / connect). */
typedef struct {
    uint64_t closest_queries, any_queries;
    uint64_t inner_visits;     /* inner nodes popped  (64 B each)  */
    uint64_t tri_tests;        /* triangles tested    (48 B each)  */
    uint64_t leaf_visits;
    uint64_t diffuse_events;   /* == raysTraced (tracer.go:666)    */
    uint64_t samples;          /* primary samples completed        */
    /* the any-hit walks' share of inner_visits / tri_tests (the closest-hit share is the rest) */
    uint64_t any_inner_visits, any_tri_tests;
    /* integrator bookkeeping for the render leg's roofline (always counted): queue entries shaded by the kernel for
     * returning calls / fresh diffuse hits / everything else, and parked frames read back */
    uint64_t shade_ret_visits, shade_diff_visits, shade_general_visits, frames_read;
    uint64_t extend_entries;   /* pending TraceRay calls traced (a marching call makes many closest-hit queries) */
} PtCounters;
int ptmi_set_counting(ptmi_ctx *ctx, int enabled);
int ptmi_reset_counters(ptmi_ctx *ctx);
int ptmi_counters(ptmi_ctx *ctx, PtCounters *out);   /* synchronises the device */

/* ------------------------------------------------------------------ scene + integrator */

/* Material record the integrator reads (obj.go:63-79).  The reference detects glass and the
 * accretion disk by material NAME (tracer.go:49 prefix "Glass", :95 == "AccretionDisk"); the host
 * flattens that to flags.  64 bytes. */
enum { PTMI_MAT_GLASS = 1, PTMI_MAT_ACCRETION = 2 };
typedef struct {
    float kd[3];        /* Diffuse   */
    float ks[3];        /* Specular  */
    float ke[3];        /* Emissive  */
    float ni;           /* Refraction */
    float ns;           /* Shininess */
    uint32_t flags;
    uint32_t diffuse_map;   /* Material.DiffuseImage (obj.go:77): 1-based texture number of ptmi_upload_textures, 0 = nil */
    uint32_t bump_map;      /* Material.BumpImage    (obj.go:78): same */
    uint32_t pad[2];
} PtMaterial;

/* lights.go:11-15 (Sun: vec = Direction, used un-normalised as the reference does) and :32-35
 * (PointLight: vec = Position).  48 bytes. */
enum { PTMI_LIGHT_SUN = 0, PTMI_LIGHT_POINT = 1 };
typedef struct {
    int32_t kind;
    float color[3];
    float vec[3];
    float intensity;
    float pad[4];
} PtLight;

/* skybox.go: nil / SolidColorSkybox (colour in `horizon`) / GradientSkybox / ImageSkybox (equirect
 * RGB16 image passed beside the struct).  64 bytes. */
enum { PTMI_SKY_NONE = 0, PTMI_SKY_SOLID = 1, PTMI_SKY_GRADIENT = 2, PTMI_SKY_IMAGE = 3 };
typedef struct {
    int32_t kind;
    float horizon[3], zenith[3], ground[3];
    float intensity;
    int32_t image_width, image_height;
    float pad[3];
} PtSky;

/* BlackHole.go:10-14, 166-170 (only BlackHoles[0] is used by the reference: tracer.go:28-31). */
typedef struct {
    float position[3];
    float rs;
    int32_t has_disk;
    float disk_inner, disk_outer;
    float pad;
} PtBlackHole;

/* camera.go:9-15 */
typedef struct {
    float position[3], forward[3], right[3], up[3];
    float frustum_distance;
} PtCamera;

/* The literals of main.go:441-452 that TraceRay receives positionally (main.go:875), plus the seed
 * of the counter-based RNG that replaces Go's unseeded global math/rand. */
typedef struct {
    int32_t width, height;      /* full image size: pixel (X,Y) -> ray as main.go:862-874 */
    int32_t bounces, scatter_rays, max_steps;
    float ambient, step_size;
    uint32_t seed;
} PtParams;

/* Replaces the scene-side arguments of TraceRay (vnmu *VNMU, scene *Scene: tracer.go:15,
 * vnmu.go:8-13, scene.go:3-9), flattened once (the Go pointer graphs cannot cross cgo).  Needs a
 * BVH (the vertex / index arrays given to ptmi_build_bvh are reused).  corner_normals: 3 per
 * triangle, in triangle order (vnmu.Normals); tri_material: one per triangle (vnmu.Materials,
 * as an index); emissive triangles are derived as renderer.go:311-321 does.  sky_image: RGB16
 * rows (image_height x image_width x 3) or NULL; blackhole: NULL when the scene has none. */
int ptmi_upload_scene(ptmi_ctx *ctx, const PtMaterial *materials, uint64_t nmaterials, const uint32_t *tri_material,
                      const float *corner_normals, const PtLight *lights, uint64_t nlights, const PtSky *sky,
                      const uint16_t *sky_image, const PtBlackHole *blackhole);

/* Replaces vnmu.UVs (vnmu.go:12) and the CachedImage table LoadObj builds (obj_loader.go:29-57,
 * renderer.go:88-120) for SampleDiffuseMap / SampleBumpMap / TransformNormalToWorldSpace
 * (renderer.go:122-169, :199-291; used by HandleDiffuseMaterial, tracer.go:468-514).  Call after
 * ptmi_upload_scene.  uvs: 6 floats per triangle (u,v of corners A,B,C, already flipped v = 1-v as
 * obj_loader.go:92 does); pixels: ntextures images back to back, row-major, each texel packed
 * R | G<<8 | B<<16 | A<<24 exactly as CacheImage packs it (renderer.go:103-104).  A NaN or
 * out-of-range texel coordinate is clamped into the image (the reference would index out of the
 * pixel slice).  ntextures = 0 removes all textures. */
int ptmi_upload_textures(ptmi_ctx *ctx, const float *uvs, uint32_t ntextures, const int32_t *widths,
                         const int32_t *heights, const uint32_t *pixels);

/* Replaces the body of the `calculate` worker (main.go:832-899) for one tile: for every pixel of the
 * rectangle (x0,y0,w,h) trace samples [spp_begin, spp_begin+spp_count) -- camera ray (main.go:862-874),
 * TraceRay (tracer.go:15) -- and ADD them, in sample order, into accum_rgb[w*h*3] and counts[w*h]:
 * the R,G,B,SampleCount sums of Pixel.AddSample (renderer.go:29,59-61).  Host buffers; synchronous. */
int ptmi_render(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0, int32_t w,
                int32_t h, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts);
/* Same with device buffers; work is enqueued on `stream`, but the call itself waits for the
 * wavefront loop's queue counters, so it returns when the rectangle is done. */
int ptmi_render_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0, int32_t w,
                    int32_t h, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts,
                    void *stream);
/* One GPU's share of a frame in ONE call: the rows y with (y / row_block) % nshards == shard, full width, added
 * into FULL-FRAME device buffers accum_rgb[height*width*3] / counts[height*width] (other rows untouched).  The
 * reference deals 64x64 tiles to goroutines (main.go:454-491, :935-937); a GPU wants its whole share at once --
 * every call ends by draining its own path pool, 10-15 iterations on fewer and fewer paths -- so the shard is an
 * interleave of row blocks, and a sum of the shards' buffers (RCCL reduce) is the frame. */
int ptmi_render_rows_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t row_block,
                         int32_t shard, int32_t nshards, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                         uint32_t *counts, void *stream);
/* Several tiles of tile.go's split in ONE call (the reference starts a goroutine per tile, main.go:935-937; a shim
 * collects them): tiles_xywh[4*k..] = (x0, y0, w, h) of tile k, disjoint, all rendered in one wavefront pass and added
 * into FULL-FRAME device buffers as above.  The sums are those of one ptmi_render_dev call per tile, bit for bit. */
int ptmi_render_tiles_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t ntiles,
                          const int32_t *tiles_xywh, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                          uint32_t *counts, void *stream);
/* ptmi_render_tiles_dev that also keeps the rest of Pixel.AddSample's state (renderer.go:25-82) in
 * pixel_stats[(y*width + x)*PTMI_PIXEL_STATS + k] (device, full frame, zeroed by the caller before a pixel's first sample):
 * k = 0, 1 MinLuminance, MaxLuminance (:63-78; luminance = 0.2126 R + 0.7152 G + 0.0722 B of a sample, :84-86; the first
 * sample a pixel ever gets sets both, later ones widen the range), 2-4 Mean, 5-7 M2 (:33-51), 8 Variance (:53-57), every
 * operation as the source writes it, in sample order.  Pixel.Contrast (:81) = max - min is what Tile.GetNoisiestPixel
 * (tile.go:34-66) ranks and stops pixels by.  That picker hands out samplesPerPixel samples at a time to one pixel of a
 * tile, but whether a pixel is picked AGAIN depends on that pixel alone (its count against maxSamplesPerPixel, and
 * count > 32 && contrast < 0.01, :39, :51), and its contrast only grows, so the finished image takes two calls: all
 * pixels up to the first count above 32, then the host reads the contrast, lists the runs of pixels at or above 0.01 as
 * 1-row tiles, and calls this again for their remaining samples (pathtrace_amd/adaptive.py; INTEGRATION.md has the Go
 * text). */
#define PTMI_PIXEL_STATS 9
int ptmi_render_tiles_stats_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t ntiles,
                                const int32_t *tiles_xywh, uint32_t spp_begin, uint32_t spp_count, float *accum_rgb,
                                uint32_t *counts, float *pixel_stats, void *stream);
/* The whole adaptive frame in one call: what the reference's worker loop (main.go:837-897) around
 * Tile.GetNoisiestPixel(maxSamplesPerPixel) (tile.go:34-66) leaves in the Pixels when every tile reports "completed
 * rendering", with samples_per_pixel = samplesPerPixel and max_samples = maxSamplesPerPixel.  Two passes as described
 * above: every pixel up to the first SampleCount above 32, then the pixels whose Contrast is not under 0.01 up to the
 * count at which the picker drops them (max_samples, or the first count above it when it is not a multiple of the batch:
 * main.go:880).  accum_rgb / counts / pixel_stats are full-frame device buffers as for ptmi_render_tiles_stats_dev,
 * ZEROED by the caller (a fresh frame).  The order in which the picker would have visited the pixels is not reproduced
 * (it does not change the result); tests/test_adaptive.py holds the frame against the sequential loop bit for bit. */
int ptmi_render_adaptive_dev(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, uint32_t samples_per_pixel,
                             uint32_t max_samples, float *accum_rgb, uint32_t *counts, float *pixel_stats, void *stream);
/* Several frames of one scene in ONE call -- the orbit of the reference's frame loop (main.go:903-937: when all tile
 * workers are done save orbit/frame<j>.png, theta += 1 degree, `scene.Camera.SphericalAround(Vec3{}, 1500, theta, phi)`
 * (camera.go:17), start the workers again), nframes cameras at a time: frame f is rendered through
 * cameras[f] (host array, borrowed for the call) into accum_rgb[f][height][width][3] / counts[f][height][width] (device,
 * contiguous), samples [spp_begin, spp_begin + spp_count) of every pixel added in sample order.  The sums of frame f are
 * those of ptmi_render_dev with cameras[f] into its slice, bit for bit.  What the call buys: a call ends by draining its
 * path pool (iterations on fewer and fewer paths, a third of a 1080p frame at 32 spp); here the samples of the next
 * frames fill the pool while the first ones drain, so the drain is paid once per call instead of once per frame.
 * width x height x nframes x 3 must stay below 2^32. */
int ptmi_render_frames_dev(ptmi_ctx *ctx, const PtCamera *cameras, uint32_t nframes, const PtParams *params, uint32_t spp_begin,
                           uint32_t spp_count, float *accum_rgb, uint32_t *counts, void *stream);
/* Per-sample radiance without accumulation: out[(pixel*spp_count + s)*3], host buffer. */
int ptmi_render_samples(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t x0, int32_t y0,
                        int32_t w, int32_t h, uint32_t spp_begin, uint32_t spp_count, float *out_rgb);

/* Replaces Vec3.ToRGBA (vec3.go:95-106) + the write-back of main.go:886-896: mean = sum/count,
 * sqrt-gamma, clamp, truncating uint8, written mirrored to column width - X (X = 0 is dropped and
 * column 0 never written, as in the reference).  rgba: width*height*4 bytes, host, caller-initialised. */
int ptmi_to_rgba(ptmi_ctx *ctx, const float *accum_rgb, const uint32_t *counts, int32_t width, int32_t height,
                 uint8_t *rgba);

/* No counterpart in the reference (its working set is the goroutines' stacks).  The integrator keeps its paths in a
 * pool of device memory that is sized for the call -- as many paths as the call has samples, up to max_paths paths and
 * max_bytes bytes (about 1 KB per path with bounces = 2 and two lights) -- and kept for the next call.  Every wavefront
 * iteration has a cost that does not depend on the number of paths in it, so the defaults are generous (2^27 paths,
 * 160 GiB of the 288 GB: C3 at 256 spp runs 11 % faster than with a quarter of that); a host that shares the GPU
 * sets its own limits here (0 keeps a value; at least 2^16 paths and 64 MiB).  The image does not depend on the pool's
 * size.  A pool already over the new limit is given back at once (the call synchronises the device).  When device
 * memory is short a render call halves its pool until it fits, so the limit is a ceiling, not a reservation.
 * ptmi_pool_info reports the pool in place (0, 0 before the first render call). */
int ptmi_set_pool_limit(ptmi_ctx *ctx, uint64_t max_paths, uint64_t max_bytes);
int ptmi_pool_info(ptmi_ctx *ctx, uint64_t *paths, uint64_t *bytes);

/* No counterpart in the reference.  The geodesic march (BlackHole.go:74-110) takes one float32 square root and one
 * reciprocal per acceleration, 12 000 of each per sample; away from the ends of the exponent range the library computes
 * them with shorter instruction sequences than the compiler's general ones.  This call compares the two on EVERY float
 * of the ranges the short forms are used on (0.7 G square roots, 1.0 G reciprocals; a few milliseconds) and returns the
 * number that differ in any bit: 0 on a correct installation.  For integrators who want that shown on their hardware. */
int ptmi_selftest_math(ptmi_ctx *ctx, uint64_t *mismatches);

/* Per-kernel timing of the render calls -- the counterpart of the reference's runtime/pprof CPU profile around its
 * workers (main.go:797-822).  While enabled, every kernel launch of the wavefront loop is bracketed by a pair of HIP
 * events on its stream, a pass runs as ONE loop (so that the kernels do not overlap and their times add up to the
 * wall time), and the totals accumulate until they are reset.  Off by default: the events cost a few per cent. */
typedef struct {
    double prepare_ms;     /* loop bookkeeping kernels                                  */
    double generate_ms;    /* camera rays                      (main.go:861-874)        */
    double extend_ms;      /* closest-hit walks                (tracer.go:34-35, :226-256) */
    double connect_ms;     /* shadow-ray walks                 (QuickCheckIntersection) */
    double shade_ms;       /* TraceRay / HandleDiffuseMaterial / HandleReflectiveMaterial */
    double accumulate_ms;  /* Pixel.AddSample sums             (renderer.go:59-61)      */
    uint64_t iterations;   /* wavefront iterations enqueued (the last few of a call are empty) */
    uint64_t launches;     /* kernel launches timed                                     */
} PtRenderProfile;
int ptmi_set_profiling(ptmi_ctx *ctx, int enabled);     /* also resets the totals */
int ptmi_render_profile(ptmi_ctx *ctx, PtRenderProfile *out);

/* ------------------------------------------------------------------ multi-GPU (RCCL over xGMI) */

/* No counterpart in the reference: its tile workers are goroutines of one process that write disjoint Pixels of one
 * shared buffer (main.go:472-491 sets the tiles up, :935-937 starts one worker per tile, :878 / :896 are the
 * unsynchronised writes).  Across GPUs the same disjoint regions are put together by ONE collective per frame: every
 * GPU adds its share into a zero-initialised full-frame accumulator and an RCCL reduce (sum) puts the frame on
 * `root`; the regions are disjoint, so the reduced sums equal a single-GPU render bit for bit.  BVH and scene are
 * replicated (ptmi_build_bvh / ptmi_upload_scene on every context).  RCCL is loaded when the first of these calls
 * is made (librccl.so of the process or of the ROCm installation; PTMI_RCCL_LIB overrides); without it they return
 * PTMI_ERR_COMM and everything else keeps working.
 *
 * One communicator per context.  Either one process (or thread) per GPU: rank 0 calls ptmi_comm_unique_id, hands
 * the 128 bytes to the other ranks by whatever channel the host has, and EVERY rank calls ptmi_comm_init_rank
 * (collective).  Or one process driving all GPUs of a node (a Go host: one goroutine per GPU):
 * ptmi_comm_init_all(ctxs, n) makes ctxs[i] rank i.  Collectives are enqueued on `stream` and must be called by all
 * ranks in the same order; with several contexts in one process call them from one thread per context (a collective
 * returns once every rank has joined it). */
#define PTMI_UNIQUE_ID_BYTES 128
int ptmi_comm_unique_id(uint8_t id[PTMI_UNIQUE_ID_BYTES]);
int ptmi_comm_init_rank(ptmi_ctx *ctx, int nranks, int rank, const uint8_t id[PTMI_UNIQUE_ID_BYTES]);
int ptmi_comm_init_all(ptmi_ctx **ctxs, int n);
int ptmi_comm_rank(const ptmi_ctx *ctx, int *rank, int *nranks);     /* (0, 1) without a communicator */
int ptmi_comm_destroy(ptmi_ctx *ctx);                                /* also done by ptmi_destroy */

/* Sum of all ranks' full-frame device buffers onto `root` (ncclReduce, float32 sums + uint32 counts in one group):
 * recv_* are only written on root and may equal send_* (in place).  A context without a communicator is a world of
 * one: the call copies (or does nothing in place).  Replaces the shared-memory Pixel buffer of main.go:878. */
int ptmi_reduce_frame(ptmi_ctx *ctx, const float *send_accum, const uint32_t *send_counts, float *recv_accum,
                      uint32_t *recv_counts, int32_t width, int32_t height, int root, void *stream);

/* A whole frame over all ranks in one call per rank -- the multi-GPU form of the `calculate` workers
 * (main.go:832-899, :935-937): clears the full-frame device buffers, renders this rank's rows (ptmi_render_rows_dev
 * with shard = rank, nshards = nranks: rows y with (y / row_block) % nranks == rank) for samples
 * [spp_begin, spp_begin + spp_count), and reduces onto `root`, whose buffers then hold the frame -- bit-identical to
 * ptmi_render_dev of the whole frame into zeroed buffers on one GPU.  To continue a frame over several calls keep
 * per-rank buffers with ptmi_render_rows_dev and reduce once with ptmi_reduce_frame (out of place).  If a rank's
 * render fails it still joins the reduce (the other ranks are already waiting in it) and then returns its error: no
 * rank hangs, the frame on `root` is not valid, and the host learns so from the failing rank's return code. */
int ptmi_render_sharded(ptmi_ctx *ctx, const PtCamera *camera, const PtParams *params, int32_t row_block,
                        uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts, int root,
                        void *stream);

/* ------------------------------------------------------------------ assets (host only) */

/* Replaces LoadObj (obj_loader.go:19-111, over the vendored decoder obj.go:101-226, :405-939) and the
 * flattening of DecomposeObjects (renderer.go:293-324) for one object at `position`: parses path and
 * its material library and yields exactly the arrays ptmi_build_bvh / ptmi_upload_scene /
 * ptmi_upload_textures take.  Reference behaviour that is kept: only the first object/group of the
 * file, triangles only, vn and usemtl required, vertices * scale, normals normalised, v -> 1-v,
 * glass / accretion disk by material name, the grey default material when no .mtl parses.  Where the
 * reference panics (non-triangle face, missing vn / vt / usemtl, bad index) this returns
 * PTMI_ERR_INVALID with a message in ptmi_obj_last_error().  Texture files are NOT decoded: their
 * names come back through ptmi_obj_string and the host decodes them as LoadObj :29-57 does. */
typedef struct ptmi_obj ptmi_obj;
int ptmi_obj_load(const char *path, float scale, const float position[3], ptmi_obj **out);
void ptmi_obj_free(ptmi_obj *obj);
const char *ptmi_obj_last_error(void);
int ptmi_obj_sizes(const ptmi_obj *obj, uint64_t *nverts, uint64_t *ntris, uint64_t *nmaterials, int32_t *has_uvs,
                   uint64_t *nwarnings);
/* Any pointer may be NULL.  corner_normals: 9 floats per triangle; uvs: 6 per triangle (only when has_uvs). */
int ptmi_obj_get(const ptmi_obj *obj, float *verts_xyz, uint32_t *tri_idx, float *corner_normals,
                 uint32_t *tri_material, PtMaterial *materials, float *uvs);
/* which: 0 material name, 1 map_Kd file, 2 map_Bump file (index = material), 3 warning text (index =
 * warning number), 4 directory of the .obj.  Owned by obj; "" when absent. */
const char *ptmi_obj_string(const ptmi_obj *obj, int which, uint64_t index);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
