/*
 * oracle/orc_trace.c -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
 * PARITY UNPINNED (no reference fixtures exist).
 *
 * Recursive CPU restatement of the reference's integrator, kept in the reference's own shape:
 *   TraceRay                         tracer.go:15-264
 *   HandleDiffuseMaterial            tracer.go:443-668
 *   HandleReflectiveMaterial         tracer.go:670-722
 *   SampleGlossyReflection           tracer.go:724-747
 *   RefractiveIndexTracker           refractive_index_tracker.go:3-40
 *   InterpolateNormal, SampleTrianglePoint, TriangleArea, cosine sampling, reflect,
 *   GetRefractedRay                  utils.go:127-151, 217-307
 *   MISWeight                        renderer.go:326-336
 *   Skybox.Sample x3                 skybox.go:23-25, 36-42, 70-107
 *   Sun.Sample / PointLight.Sample   lights.go:18-54
 *   GetInitialState / GetAcceleration / Step / GetProceduralColor   BlackHole.go:24-242
 *   camera ray + Pixel.AddSample sums main.go:861-879, renderer.go:59-61
 *
 * Documented deviations (the reference is not reproducible here):
 *   - RNG: the reference draws from Go's unseeded global math/rand; here every primary sample owns
 *     a counter-based stream keyed by (seed, pixel, sample) and drawn in the reference's draw order.
 *   - sin/cos/atan/atan2/acos/pow come from libm (the reference: chewxy/math32, g3n/math32).
 *   - Perlin noise of the accretion disk: own gradient noise (go-perlin's seeded table is not
 *     reproducible offline).
 *   - textures / bump maps (renderer.go:88-291) are not restated yet.
 */
#define _POSIX_C_SOURCE 200809L
#include "orc_bvh.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI32 ((float)3.14159265358979323846)

/* ------------------------------------------------------------------ scene */

typedef struct {
    float kd[3], ks[3], ke[3];
    float ni, ns;
    uint32_t flags;          /* 1 = name has prefix "Glass" (tracer.go:49), 2 = name == "AccretionDisk" (:95) */
    uint32_t diffuse_map, bump_map;   /* 1-based texture numbers, 0 = nil (Material.DiffuseImage / BumpImage, obj.go:77-78) */
} orc_material;

/* renderer.go:88-92 CachedImage: packed R | G<<8 | B<<16 | A<<24 (renderer.go:103-104) */
typedef struct { const uint32_t *pixels; int width, height; } orc_image;

typedef struct { int kind; v3 color; v3 vec; float intensity; } orc_light;   /* kind 0 Sun, 1 PointLight */

typedef struct {
    int kind;                /* 0 nil, 1 solid, 2 gradient, 3 image */
    v3 horizon, zenith, ground;
    float intensity;
    const uint16_t *img; int img_w, img_h;   /* RGB16, row-major */
} orc_sky;

typedef struct { int present; v3 pos; float rs; int has_disk; float inner, outer; } orc_blackhole;

typedef struct {
    const orc_bvh *bvh;
    const float *verts; const uint32_t *tri_idx; uint64_t ntris;
    const float *normals;            /* 3 per triangle corner */
    const uint32_t *tri_material;
    orc_material *materials; uint64_t nmaterials;
    uint32_t *emissive; uint64_t nemissive;      /* triangle numbers (renderer.go:311-321) */
    orc_light *lights; uint64_t nlights;
    orc_sky sky;
    orc_blackhole bh;
    const float *uvs;                /* vnmu.UVs: 6 per triangle (vnmu.go:12) */
    orc_image *images; uint64_t nimages;
} orc_scene;

typedef struct {
    int width, height, bounces, scatter_rays, max_steps;
    float ambient, step_size;
    uint32_t seed;
} orc_params;

typedef struct { uint64_t closest, any, diffuse_events, samples; } orc_trace_counters;

/* ------------------------------------------------------------------ counter-based RNG */

static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
typedef struct { uint32_t key, ctr; } orc_rng;
static inline orc_rng rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
    orc_rng r;
    r.key = mix32(mix32(mix32(seed ^ 0xA511E9B3u) + pixel * 0x9E3779B9u) ^ (sample * 0x85EBCA6Bu + 0x27D4EB2Fu));
    r.ctr = 0;
    return r;
}
static inline uint32_t rng_u32(orc_rng *r) { return mix32(r->key + 0x9E3779B9u * (r->ctr++)); }
/* rand.Float32(): [0,1) */
static inline float rng_f32(orc_rng *r) { return (float)(rng_u32(r) >> 8) * (float)(1.0 / 16777216.0); }
/* rand.Intn(n) */
static inline uint32_t rng_intn(orc_rng *r, uint32_t n) { return (uint32_t)(((uint64_t)rng_u32(r) * (uint64_t)n) >> 32); }

/* ------------------------------------------------------------------ tracker (refractive_index_tracker.go) */

#define ORC_TRACKER_MAX 64
typedef struct { float hist[ORC_TRACKER_MAX]; int len; } orc_tracker;
static inline void trk_update(orc_tracker *t, float v) { if (t->len < ORC_TRACKER_MAX) t->hist[t->len++] = v; }
static inline float trk_current(const orc_tracker *t) { return t->len == 0 ? 1.0f : t->hist[t->len - 1]; }
static inline float trk_previous(const orc_tracker *t) { return t->len < 2 ? 1.0f : t->hist[t->len - 2]; }
static inline void trk_pop(orc_tracker *t) { if (t->len > 0) t->len--; }

/* ------------------------------------------------------------------ utils.go */

/* utils.go:127-151 */
static v3 interpolate_normal(v3 p, v3 a, v3 b, v3 c, v3 nA, v3 nB, v3 nC) {
    v3 v0 = v3_sub(b, a), v1 = v3_sub(c, a), v2 = v3_sub(p, a);
    float d00 = v3_dot(v0, v0), d01 = v3_dot(v0, v1), d11 = v3_dot(v1, v1), d20 = v3_dot(v2, v0), d21 = v3_dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    nA = v3_scale(nA, u); nB = v3_scale(nB, v); nC = v3_scale(nC, w);
    nA = v3_add(nA, nB); nA = v3_add(nA, nC);
    return nA;
}
/* utils.go:217-229 */
static v3 sample_triangle_point(orc_rng *r, v3 A, v3 B, v3 C) {
    float u = rng_f32(r);
    float v = rng_f32(r) * (1.0f - u);
    float w = 1.0f - u - v;
    A = v3_scale(A, w); B = v3_scale(B, v); C = v3_scale(C, u);
    A = v3_add(A, B); A = v3_add(A, C);
    return A;
}
/* utils.go:231-236 */
static float triangle_area(v3 A, v3 B, v3 C) { return 0.5f * v3_length(v3_cross(v3_sub(B, A), v3_sub(C, A))); }

/* utils.go:273-290 (and the tail of :242-271) */
static v3 cosine_sample2(orc_rng *r, v3 normal, v3 t1, v3 t2) {
    float u1 = rng_f32(r), u2 = rng_f32(r);
    float rr = sqrtf(u1);
    float theta = (float)(2.0 * 3.14159265358979323846) * u2;
    float x = rr * cosf(theta), y = rr * sinf(theta);
    float z = sqrtf(gomaxf(0.0f, 1.0f - u1));
    v3 dir = v3_scale(t1, x);
    dir = v3_add(dir, v3_scale(t2, y));
    dir = v3_add(dir, v3_scale(normal, z));
    return v3_normalize(dir);
}
/* utils.go:242-253 / tracer.go:625-634: tangent frame with the 0.999 threshold */
static void tangent_frame(v3 normal, v3 *t1, v3 *t2) {
    v3 up;
    if (fabsf(normal.y) < (float)0.999) { up.x = 0; up.y = 1; up.z = 0; } else { up.x = 1; up.y = 0; up.z = 0; }
    *t1 = v3_normalize(v3_cross(normal, up));
    *t2 = v3_cross(normal, *t1);
}
static v3 cosine_sample(orc_rng *r, v3 normal) {
    v3 t1, t2; tangent_frame(normal, &t1, &t2);
    return cosine_sample2(r, normal, t1, t2);
}
/* utils.go:292-294 */
static v3 reflect_v(v3 V, v3 N) { return v3_sub(V, v3_scale(N, 2.0f * v3_dot(V, N))); }
/* utils.go:296-307 */
static v3 get_refracted_ray(v3 inDir, v3 inNormal, float n1, float n2, int *tir) {
    v3 A = v3_normalize(inDir), N = v3_normalize(inNormal);
    float R = n1 / n2;
    float c = -v3_dot(A, N);
    float D = 1.0f - R * R * (1.0f - c * c);
    if (D < 0.0f) { *tir = 1; return reflect_v(inDir, inNormal); }
    *tir = 0;
    return v3_normalize(v3_add(v3_scale(A, R), v3_scale(N, c * R - sqrtf(D))));
}
/* renderer.go:326-336 */
static float mis_weight(float pdf1, float pdf2) {
    if (pdf1 <= 0.0f) return 0.0f;
    if (pdf2 <= 0.0f) return 1.0f;
    float p1 = pdf1 * pdf1, p2 = pdf2 * pdf2;
    return p1 / (p1 + p2);
}

/* ------------------------------------------------------------------ skybox.go */

static v3 sky_sample(const orc_sky *s, v3 dir) {
    v3 zero = { 0, 0, 0 };
    switch (s->kind) {
    case 1: return s->horizon;                                             /* skybox.go:23-25 */
    case 2: {                                                              /* skybox.go:36-42 */
        v3 upv = { 0, 1, 0 };
        float angle = v3_dot(dir, upv);
        if (angle < 0.0f) return s->ground;
        return v3_scale(v3_add(v3_scale(s->horizon, 1.0f - angle), v3_scale(s->zenith, angle)), s->intensity);
    }
    case 3: {                                                              /* skybox.go:70-107 */
        v3 d = v3_normalize(dir);
        float phi = atan2f(d.z, d.x), theta = acosf(d.y);
        float u = (phi + PI32) / (2.0f * PI32), v = theta / PI32;
        float fx = u * (float)(s->img_w - 1), fy = v * (float)(s->img_h - 1);
        int x = (fx == fx) ? (int)fx : 0, y = (fy == fy) ? (int)fy : 0;
        if (x < 0) x = 0; if (x > s->img_w - 1) x = s->img_w - 1;
        if (y < 0) y = 0; if (y > s->img_h - 1) y = s->img_h - 1;
        const uint16_t *p = s->img + 3 * ((size_t)y * s->img_w + x);
        v3 c = { (float)p[0] / 65535.0f, (float)p[1] / 65535.0f, (float)p[2] / 65535.0f };
        return v3_scale(c, s->intensity);
    }
    default: return zero;
    }
}

/* ------------------------------------------------------------------ BlackHole.go */

typedef struct { float px, py, pz, vx, vy, vz, pt, vt; } ray_state;

/* BlackHole.go:24-69 */
static ray_state get_initial_state(v3 position, v3 direction, const orc_blackhole *bh) {
    ray_state z; memset(&z, 0, sizeof z);
    v3 rel = v3_sub(position, bh->pos);
    float r = v3_length(rel);
    if (r == 0.0f) return z;
    float rs = bh->rs;
    float p_dot_v = v3_dot(rel, direction), v_dot_v = v3_dot(direction, direction);
    float A = rs / r - 1.0f;
    float B = 2.0f * rs * p_dot_v / (r * r);
    float C = v_dot_v + rs * p_dot_v * p_dot_v / (r * r * r);
    float disc = B * B - 4.0f * A * C;
    if (disc < 0.0f) return z;
    float vt = (-B + sqrtf(disc)) / (2.0f * A);
    ray_state s;
    s.pt = 0.0f; s.px = rel.x; s.py = rel.y; s.pz = rel.z; s.vt = vt; s.vx = direction.x; s.vy = direction.y; s.vz = direction.z;
    return s;
}
/* BlackHole.go:74-110 */
static void get_acceleration(const ray_state *s, const orc_blackhole *bh, float a[4]) {
    float rs = bh->rs;
    float px = s->px, py = s->py, pz = s->pz, vx = s->vx, vy = s->vy, vz = s->vz;
    float r_sq = (px * px + py * py) + pz * pz;
    if (r_sq <= rs * rs) { a[0] = a[1] = a[2] = a[3] = 0.0f; return; }
    float r = sqrtf(r_sq);
    float r_cubed = r_sq * r;
    float inv_r_cubed = 1.0f / r_cubed;
    float x_dot_v = (px * vx + py * vy) + pz * vz;
    float x_dot_v_sq = x_dot_v * x_dot_v;
    float rs_x_dot_v = rs * x_dot_v;
    float two = 2.0f * x_dot_v * inv_r_cubed;
    a[0] = -2.0f * rs_x_dot_v * inv_r_cubed;
    float factor = (1.0f - 3.0f * rs * x_dot_v_sq * inv_r_cubed) * inv_r_cubed;
    float rs_inv_factor = -rs * factor;
    a[1] = px * rs_inv_factor + rs * two * vx;
    a[2] = py * rs_inv_factor + rs * two * vy;
    a[3] = pz * rs_inv_factor + rs * two * vz;
}
typedef struct { float at, ax, ay, az, vt, vx, vy, vz; } rk4d;
/* BlackHole.go:118-127 */
static void evaluate_rk4(const ray_state *s, float dt, const rk4d *k, ray_state *o) {
    o->pt = s->pt + k->vt * dt; o->px = s->px + k->vx * dt; o->py = s->py + k->vy * dt; o->pz = s->pz + k->vz * dt;
    o->vt = s->vt + k->at * dt; o->vx = s->vx + k->ax * dt; o->vy = s->vy + k->ay * dt; o->vz = s->vz + k->az * dt;
}
/* BlackHole.go:131-163 */
static ray_state rk4_step(const ray_state *s, float h, const orc_blackhole *bh) {
    float a[4]; ray_state tmp; memset(&tmp, 0, sizeof tmp);
    get_acceleration(s, bh, a);
    rk4d k1 = { a[0], a[1], a[2], a[3], s->vt, s->vx, s->vy, s->vz };
    evaluate_rk4(s, h * 0.5f, &k1, &tmp); get_acceleration(&tmp, bh, a);
    rk4d k2 = { a[0], a[1], a[2], a[3], tmp.vt, tmp.vx, tmp.vy, tmp.vz };
    evaluate_rk4(s, h * 0.5f, &k2, &tmp); get_acceleration(&tmp, bh, a);
    rk4d k3 = { a[0], a[1], a[2], a[3], tmp.vt, tmp.vx, tmp.vy, tmp.vz };
    evaluate_rk4(s, h, &k3, &tmp); get_acceleration(&tmp, bh, a);
    rk4d k4 = { a[0], a[1], a[2], a[3], tmp.vt, tmp.vx, tmp.vy, tmp.vz };
    float h6 = h / 6.0f;
    ray_state o;
    o.pt = s->pt + h6 * (((k1.vt + 2.0f * k2.vt) + 2.0f * k3.vt) + k4.vt);
    o.px = s->px + h6 * (((k1.vx + 2.0f * k2.vx) + 2.0f * k3.vx) + k4.vx);
    o.py = s->py + h6 * (((k1.vy + 2.0f * k2.vy) + 2.0f * k3.vy) + k4.vy);
    o.pz = s->pz + h6 * (((k1.vz + 2.0f * k2.vz) + 2.0f * k3.vz) + k4.vz);
    o.vt = s->vt + h6 * (((k1.at + 2.0f * k2.at) + 2.0f * k3.at) + k4.at);
    o.vx = s->vx + h6 * (((k1.ax + 2.0f * k2.ax) + 2.0f * k3.ax) + k4.ax);
    o.vy = s->vy + h6 * (((k1.ay + 2.0f * k2.ay) + 2.0f * k3.ay) + k4.ay);
    o.vz = s->vz + h6 * (((k1.az + 2.0f * k2.az) + 2.0f * k3.az) + k4.az);
    return o;
}

/* Gradient noise standing in for go-perlin's Noise2D(alpha=2, beta=2, n=4) -- own lattice hash. */
static double grad2(int ix, int iy, double fx, double fy) {
    uint32_t h = mix32((uint32_t)ix * 0x9E3779B9u ^ mix32((uint32_t)iy * 0x85EBCA6Bu + 0x165667B1u));
    static const double g[8][2] = { { 1, 0 }, { -1, 0 }, { 0, 1 }, { 0, -1 },
                                    { 0.70710678118654752, 0.70710678118654752 }, { -0.70710678118654752, 0.70710678118654752 },
                                    { 0.70710678118654752, -0.70710678118654752 }, { -0.70710678118654752, -0.70710678118654752 } };
    return g[h & 7][0] * fx + g[h & 7][1] * fy;
}
static double noise2(double x, double y) {
    double flx = floor(x), fly = floor(y);
    int ix = (int)flx, iy = (int)fly;
    double fx = x - flx, fy = y - fly;
    double sx = fx * fx * (3.0 - 2.0 * fx), sy = fy * fy * (3.0 - 2.0 * fy);
    double a = grad2(ix, iy, fx, fy), b = grad2(ix + 1, iy, fx - 1.0, fy);
    double c = grad2(ix, iy + 1, fx, fy - 1.0), d = grad2(ix + 1, iy + 1, fx - 1.0, fy - 1.0);
    double u = a + sx * (b - a), v = c + sx * (d - c);
    return u + sy * (v - u);
}
static double perlin_noise2d(double x, double y) {          /* alpha = 2, beta = 2, n = 4 (main.go:658) */
    double scale = 1.0, sum = 0.0;
    for (int i = 0; i < 4; ++i) { sum += noise2(x, y) / scale; scale *= 2.0; x *= 2.0; y *= 2.0; }
    return sum;
}

/* BlackHole.go:174-242 */
static v3 disk_color(const orc_blackhole *bh, v3 world, v3 bhpos) {
    v3 rel = v3_sub(world, bhpos);
    float radius = sqrtf(rel.x * rel.x + rel.z * rel.z);
    float angle = atan2f(rel.z, rel.x);
    float t = (radius - bh->inner) / (bh->outer - bh->inner);
    if (t < 0.0f) t = 0.0f;
    if (t > 1.0f) t = 1.0f;
    v3 hot = { 1.0f, 1.0f, 0.85f }, mid = { 1.0f, 0.6f, 0.0f }, cool = { 0.8f, 0.1f, 0.0f };
    v3 temp;
    if (t < 0.5f) temp = v3_lerp(hot, mid, t * 2.0f); else temp = v3_lerp(mid, cool, (t - 0.5f) * 2.0f);
    float intensity = 1.0f / (t * t + 0.1f);
    temp = v3_scale(temp, intensity);
    float noiseScale = 3.0f, stretch = 2.0f;
    float nx = (radius / bh->outer) * noiseScale * stretch;
    float ny = (angle / (float)(2.0 * 3.14159265358979323846)) * noiseScale;
    float nv = (float)(perlin_noise2d((double)nx, (double)ny) + 1.0) / 2.0f;
    nv = powf(nv, 5.0f);
    return v3_scale(temp, nv);
}

/* ------------------------------------------------------------------ tracer.go */

typedef struct {
    const orc_scene *sc;
    orc_params p;
    orc_rng *rng;
    orc_tracker *trk;
    orc_trace_counters *cnt;
} trace_ctx;

static v3 trace_ray(trace_ctx *tc, v3 ro, v3 rd, int bounces, int bounceIndex, v3 lastNormal, int isSpecular, float energy);

static inline v3 vert(const orc_scene *sc, uint32_t vi) { v3 r = { sc->verts[3 * (size_t)vi], sc->verts[3 * (size_t)vi + 1], sc->verts[3 * (size_t)vi + 2] }; return r; }
static inline v3 nrm(const orc_scene *sc, uint64_t ci) { v3 r = { sc->normals[3 * ci], sc->normals[3 * ci + 1], sc->normals[3 * ci + 2] }; return r; }
static inline v3 v3f(const float *p) { v3 r = { p[0], p[1], p[2] }; return r; }
static inline int is_emissive(const orc_material *m) { return m->ke[0] > 0.0f || m->ke[1] > 0.0f || m->ke[2] > 0.0f; }

static inline int quick(trace_ctx *tc, v3 o, v3 d, float range) {
    tc->cnt->any++;
    return orc_quick_check_intersection(tc->sc->bvh, o, d, range, NULL);
}

/* tracer.go:443-668.  Returns colour; *indirect_emissive as the second result. */
/* ---- textures: obj_loader.go:11-17, renderer.go:122-169, :199-291 ---- */

/* obj_loader.go:11-17 */
static float tile(float val) {
    float ip;
    float vf = modff(val, &ip);
    if (vf < 0) vf = 1 + vf;
    return vf;
}

typedef struct { uint8_t R, G, B, A; } rgba8;

/* renderer.go:122-135.  Go's int() of a NaN or out-of-range float is implementation-specific and would index out of
 * the pixel slice; the restatement clamps the texel index into the image (deviation, memory safety only). */
static rgba8 sample_diffuse_map(const orc_image *img, float x, float y) {
    float fx = x * (float)(img->width - 1), fy = y * (float)(img->height - 1);
    int px = (fx == fx) ? (int)fx : 0, py = (fy == fy) ? (int)fy : 0;
    if (px < 0) px = 0;
    if (px > img->width - 1) px = img->width - 1;
    if (py < 0) py = 0;
    if (py > img->height - 1) py = img->height - 1;
    uint32_t packed = img->pixels[(size_t)py * (size_t)img->width + (size_t)px];
    rgba8 c = { (uint8_t)packed, (uint8_t)(packed >> 8), (uint8_t)(packed >> 16), (uint8_t)(packed >> 24) };
    return c;
}

/* renderer.go:137-169 */
static v3 sample_bump_map(const orc_image *img, float x, float y, float strength) {
    float w = (float)img->width, h = (float)img->height;
    float offsetX = 1.0f / w, offsetY = 1.0f / h;
    rgba8 center = sample_diffuse_map(img, x, y);
    rgba8 right = sample_diffuse_map(img, fmodf(x + offsetX, 1.0f), y);
    rgba8 up = sample_diffuse_map(img, x, fmodf(y + offsetY, 1.0f));
    float centerHeight = (((float)0.299 * (float)center.R + (float)0.587 * (float)center.G) + (float)0.114 * (float)center.B) / 255.0f;
    float rightHeight = (((float)0.299 * (float)right.R + (float)0.587 * (float)right.G) + (float)0.114 * (float)right.B) / 255.0f;
    float upHeight = (((float)0.299 * (float)up.R + (float)0.587 * (float)up.G) + (float)0.114 * (float)up.B) / 255.0f;
    float dx = (rightHeight - centerHeight) * strength;
    float dy = (upHeight - centerHeight) * strength;
    v3 n = { -dx, -dy, 1.0f };
    return v3_normalize(n);
}

/* renderer.go:199-291 */
static v3 transform_normal_to_world(v3 tn, v3 wn, v3 A, v3 B, v3 C, const float *uv /* 6 floats of this triangle */) {
    if (tn.x == 0 && tn.y == 0 && tn.z == 1) return wn;
    v3 edge1 = v3_sub(B, A), edge2 = v3_sub(C, A);
    float d1x = uv[2] - uv[0], d1y = uv[3] - uv[1], d2x = uv[4] - uv[0], d2y = uv[5] - uv[1];
    float det = d1x * d2y - d2x * d1y;
    if (fabsf(det) < (float)1e-6) {
        v3 up = { 0, 1, 0 };
        if (!(fabsf(wn.y) < (float)0.9)) { up.x = 1; up.y = 0; }
        v3 tangent = v3_normalize(v3_cross(wn, up));
        v3 bitangent = v3_normalize(v3_cross(wn, tangent));
        v3 r = { (tn.x * tangent.x + tn.y * bitangent.x) + tn.z * wn.x,
                 (tn.x * tangent.y + tn.y * bitangent.y) + tn.z * wn.y,
                 (tn.x * tangent.z + tn.y * bitangent.z) + tn.z * wn.z };
        return v3_normalize(r);
    }
    float invDet = 1.0f / det;
    v3 tangent = { invDet * (d2y * edge1.x - d1y * edge2.x), invDet * (d2y * edge1.y - d1y * edge2.y),
                   invDet * (d2y * edge1.z - d1y * edge2.z) };
    v3 bitangent = { invDet * (-d2x * edge1.x + d1x * edge2.x), invDet * (-d2x * edge1.y + d1x * edge2.y),
                     invDet * (-d2x * edge1.z + d1x * edge2.z) };
    tangent = v3_normalize(v3_sub(tangent, v3_scale(wn, v3_dot(tangent, wn))));
    v3 calc = v3_cross(wn, tangent);
    if (v3_dot(bitangent, calc) < 0) bitangent = v3_normalize(v3_scale(calc, -1.0f));
    else bitangent = v3_normalize(calc);
    v3 r = { (tn.x * tangent.x + tn.y * bitangent.x) + tn.z * wn.x,
             (tn.x * tangent.y + tn.y * bitangent.y) + tn.z * wn.y,
             (tn.x * tangent.z + tn.y * bitangent.z) + tn.z * wn.z };
    return v3_normalize(r);
}

/* tracer.go:468-514: albedo and shading normal of HandleDiffuseMaterial for a textured material */
static void diffuse_surface(const orc_scene *sc, const orc_material *m, int64_t tri, v3 P, v3 *albedo, v3 *normal) {
    *albedo = v3f(m->kd);
    const int has_image = (m->diffuse_map || m->bump_map) && sc->uvs && sc->images;     /* Material.HasImage (obj_loader.go:54) */
    if (!has_image) return;
    const float *uv = sc->uvs + 6 * (size_t)tri;
    v3 A = vert(sc, sc->tri_idx[3 * (size_t)tri]), B = vert(sc, sc->tri_idx[3 * (size_t)tri + 1]), C = vert(sc, sc->tri_idx[3 * (size_t)tri + 2]);
    v3 v0 = v3_sub(B, A), v1 = v3_sub(C, A), v2 = v3_sub(P, A);
    float dot00 = v3_dot(v0, v0), dot01 = v3_dot(v0, v1), dot02 = v3_dot(v0, v2), dot11 = v3_dot(v1, v1), dot12 = v3_dot(v1, v2);
    float invDenom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    float u = (dot11 * dot02 - dot01 * dot12) * invDenom;
    float v = (dot00 * dot12 - dot01 * dot02) * invDenom;
    float w = 1.0f - u - v;
    float x = tile((w * uv[0] + u * uv[2]) + v * uv[4]);
    float y = tile((w * uv[1] + u * uv[3]) + v * uv[5]);
    if (m->diffuse_map && m->diffuse_map <= sc->nimages) {
        rgba8 c = sample_diffuse_map(&sc->images[m->diffuse_map - 1], x, y);
        albedo->x = powf((float)c.R / 255.0f, (float)2.2);
        albedo->y = powf((float)c.G / 255.0f, (float)2.2);
        albedo->z = powf((float)c.B / 255.0f, (float)2.2);
    }
    if (m->bump_map && m->bump_map <= sc->nimages) {
        v3 bn = sample_bump_map(&sc->images[m->bump_map - 1], x, y, 1.0f);
        *normal = v3_normalize(transform_normal_to_world(bn, *normal, A, B, C, uv));
    }
}

/* probes for the known-answer tests (tests/test_oracle.py) */
float orc_probe_tile(float v) { return tile(v); }
uint32_t orc_probe_texel(const uint32_t *pixels, int w, int h, float x, float y) {
    orc_image img = { pixels, w, h };
    rgba8 c = sample_diffuse_map(&img, x, y);
    return (uint32_t)c.R | ((uint32_t)c.G << 8) | ((uint32_t)c.B << 16) | ((uint32_t)c.A << 24);
}
void orc_probe_bump(const uint32_t *pixels, int w, int h, float x, float y, float strength, float *out3) {
    orc_image img = { pixels, w, h };
    v3 n = sample_bump_map(&img, x, y, strength);
    out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
}
void orc_probe_tbn(const float *tn, const float *wn, const float *abc9, const float *uv6, float *out3) {
    v3 r = transform_normal_to_world(v3f(tn), v3f(wn), v3f(abc9), v3f(abc9 + 3), v3f(abc9 + 6), uv6);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

static v3 handle_diffuse(trace_ctx *tc, v3 ro, v3 rd, int bounces, int64_t tri, v3 P, v3 normal, int bounceIndex,
                         float energy, int *indirect_emissive) {
    const orc_scene *sc = tc->sc;
    (void)ro; (void)rd;
    const orc_material *m = &sc->materials[sc->tri_material[tri]];
    v3 zero = { 0, 0, 0 };
    *indirect_emissive = 0;
    if (bounceIndex > 0 && is_emissive(m)) { *indirect_emissive = 1; return v3f(m->ke); }      /* :461-465 */
    v3 albedo;                                                                               /* :468-521 */
    diffuse_surface(sc, m, tri, P, &albedo, &normal);
    v3 ambientContribution = v3_scale(albedo, tc->p.ambient);                                /* :524 */
    v3 direct = zero;
    v3 rayOrigin = v3_add(P, v3_scale(normal, (float)0.001));                                /* :528 */
    if (sc->sky.kind != 0) {                                                                 /* :531-542 */
        v3 rn = cosine_sample(tc->rng, normal);
        if (!quick(tc, rayOrigin, rn, 100000.0f)) direct = v3_add(direct, v3_cmul(sky_sample(&sc->sky, rn), albedo));
    }
    for (uint64_t li = 0; li < sc->nlights; ++li) {                                          /* :545-560 */
        const orc_light *L = &sc->lights[li];
        v3 contribution = zero;
        if (L->kind == 0) {                                                                  /* lights.go:18-28 */
            v3 ld = L->vec;                                                                  /* un-normalised */
            float ndotr = v3_dot(ld, normal);
            if (!(ndotr < 0.0f)) {
                if (!quick(tc, rayOrigin, ld, tc->p.step_size)) contribution = v3_scale(L->color, ndotr * L->intensity);
            }
        } else {                                                                             /* lights.go:38-54 */
            v3 ld = v3_normalize(v3_sub(L->vec, rayOrigin));
            v3 toLight = v3_sub(L->vec, rayOrigin);
            float distance = v3_length(toLight);
            toLight = v3_normalize(toLight);
            float ndotl = v3_dot(normal, toLight);
            if (!(ndotl <= 0.0f)) {
                if (!quick(tc, rayOrigin, ld, distance)) {
                    float att = 1.0f / (distance * distance);
                    contribution = v3_scale(L->color, ndotl * L->intensity * att);
                }
            }
        }
        direct = v3_add(direct, v3_cmul(albedo, contribution));
    }
    if (sc->nemissive > 0) {                                                                 /* :562-618 */
        v3 eo = v3_add(P, v3_scale(normal, (float)0.01));
        v3 ec = zero;
        uint32_t choice = rng_intn(tc->rng, (uint32_t)sc->nemissive);
        uint32_t et = sc->emissive[choice];
        v3 p0 = vert(sc, sc->tri_idx[3 * (size_t)et]), p1 = vert(sc, sc->tri_idx[3 * (size_t)et + 1]), p2 = vert(sc, sc->tri_idx[3 * (size_t)et + 2]);
        v3 lightPoint = sample_triangle_point(tc->rng, p0, p1, p2);
        v3 lsn = interpolate_normal(lightPoint, p0, p1, p2, nrm(sc, 3 * (uint64_t)et), nrm(sc, 3 * (uint64_t)et + 1), nrm(sc, 3 * (uint64_t)et + 2));
        v3 toLight = v3_sub(lightPoint, eo);
        float distance = v3_length(toLight);
        toLight = v3_normalize(toLight);
        float ndotl = v3_dot(toLight, normal);
        if (!(ndotl <= 0.0f)) {
            if (!quick(tc, eo, toLight, distance - (float)0.01)) {
                float sndorl = -v3_dot(toLight, lsn);
                if (!(sndorl <= 0.0f)) {
                    float geometryTerm = ndotl * sndorl / (distance * distance);
                    float pdf = triangle_area(p0, p1, p2);
                    pdf = 1.0f / (pdf * (float)sc->nemissive);
                    float pdf_brdf = v3_dot(normal, toLight) / PI32;
                    float pdf_sa = pdf * (distance * distance) / sndorl;
                    float weight = mis_weight(pdf_sa, pdf_brdf);
                    v3 le = v3f(sc->materials[sc->tri_material[et]].ke);
                    v3 brdf = v3_scale(albedo, (float)(1.0 / 3.14159265358979323846));
                    ec = v3_scale(v3_cmul(le, brdf), geometryTerm * weight / pdf);
                }
            }
        }
        direct = v3_add(direct, ec);
    }
    v3 indirect = zero;                                                                      /* :620-648 */
    if (bounces > 0) {
        v3 t1, t2; tangent_frame(normal, &t1, &t2);
        for (int s = 0; s < tc->p.scatter_rays; ++s) {
            v3 dir = cosine_sample2(tc->rng, normal, t1, t2);
            v3 c = trace_ray(tc, v3_add(P, v3_scale(normal, (float)0.001)), dir, bounces - 1, bounceIndex + 1, normal, 0, energy);
            c = v3_cmul(c, albedo);
            indirect = v3_add(indirect, c);
        }
        indirect = v3_scale(indirect, 1.0f / (float)tc->p.scatter_rays);
    }
    v3 final = direct;                                                                       /* :658-664 */
    final = v3_add(final, ambientContribution);
    final = v3_add(final, v3_scale(indirect, 1.0f));
    if (bounceIndex == 0 && is_emissive(m)) final = v3_add(final, v3f(m->ke));
    tc->cnt->diffuse_events++;                                                               /* :666 */
    return final;
}

/* tracer.go:724-747 */
static v3 sample_glossy(orc_rng *r, v3 reflDir, float roughness) {
    v3 up;
    if (fabsf(reflDir.y) < (float)0.9) { up.x = 0; up.y = 1; up.z = 0; } else { up.x = 1; up.y = 0; up.z = 0; }
    v3 t1 = v3_normalize(v3_cross(reflDir, up));
    v3 t2 = v3_normalize(v3_cross(reflDir, t1));
    float theta = rng_f32(r) * 2.0f * PI32;
    float alpha = roughness * roughness;
    float u = rng_f32(r);
    float phi = atanf(alpha * sqrtf(u) / sqrtf(1.0f - u));
    float x = cosf(theta) * sinf(phi), y = sinf(theta) * sinf(phi), z = cosf(phi);
    return v3_normalize(v3_add(v3_add(v3_scale(t1, x), v3_scale(t2, y)), v3_scale(reflDir, z)));
}

/* tracer.go:670-722 */
static v3 handle_reflective(trace_ctx *tc, v3 rd, int bounces, int64_t tri, v3 P, v3 normal, int bounceIndex, float energy) {
    const orc_scene *sc = tc->sc;
    const orc_material *m = &sc->materials[sc->tri_material[tri]];
    float roughness = 1.0f / (1.0f + m->ns / 100.0f);
    v3 spec = v3f(m->ks);
    float dp = v3_dot(rd, normal);
    v3 reflDir = v3_normalize(v3_sub(rd, v3_scale(normal, 2.0f * dp)));
    v3 sum = { 0, 0, 0 };
    for (int s = 0; s < tc->p.scatter_rays; ++s) {
        v3 sd = sample_glossy(tc->rng, reflDir, roughness);
        v3 c = trace_ray(tc, v3_add(P, v3_scale(normal, (float)0.001)), sd, bounces - 1, bounceIndex + 1, normal, 1, energy);
        sum = v3_add(sum, c);
    }
    sum = v3_scale(sum, 1.0f / (float)tc->p.scatter_rays);
    return v3_cmul(sum, spec);
}

/* tracer.go:143-158 / :188-203 */
static v3 mis_on_emitter(trace_ctx *tc, v3 dc, v3 ro, v3 rd, int64_t tri, v3 P, v3 normal, v3 lastNormal) {
    const orc_scene *sc = tc->sc;
    float pdf_brdf = v3_dot(rd, lastNormal) / PI32;
    v3 A = vert(sc, sc->tri_idx[3 * (size_t)tri]), B = vert(sc, sc->tri_idx[3 * (size_t)tri + 1]), C = vert(sc, sc->tri_idx[3 * (size_t)tri + 2]);
    float area = triangle_area(A, B, C);
    float pdf_area = 1.0f / ((float)sc->nemissive * area);
    float cosLight = gomaxf(0.0f, v3_dot(rd, normal));
    float distance = v3_length(v3_sub(P, ro));
    float pdf_sa = pdf_area * distance * distance / cosLight;
    return v3_scale(dc, mis_weight(pdf_brdf, pdf_sa));
}

/* tracer.go:15-264 */
static v3 trace_ray(trace_ctx *tc, v3 ro, v3 rd, int bounces, int bounceIndex, v3 lastNormal, int isSpecular, float energy) {
    const orc_scene *sc = tc->sc;
    v3 zero = { 0, 0, 0 };
    if (energy < (float)1e-2 || bounces < 0) return zero;                                    /* :16 */
    v3 refr = zero, diffuse = zero, specular = zero, reflective = zero;
    ray_state st; memset(&st, 0, sizeof st);
    int curved = sc->bh.present;
    float vt_initial = 1.0f;
    if (curved) { st = get_initial_state(ro, rd, &sc->bh); vt_initial = st.vt; }             /* :28-31 */
    v3 rayPosition = ro;
    for (int step = 0; step < tc->p.max_steps; ++step) {                                     /* :34 */
        float t; int64_t tri;
        tc->cnt->closest++;
        int hit = orc_check_intersection(sc->bvh, ro, rd, tc->p.step_size, &t, &tri, NULL);
        if (hit) {
            v3 P = v3_add(rayPosition, v3_scale(rd, t));                                     /* :37 */
            v3 A = vert(sc, sc->tri_idx[3 * (size_t)tri]), B = vert(sc, sc->tri_idx[3 * (size_t)tri + 1]), C = vert(sc, sc->tri_idx[3 * (size_t)tri + 2]);
            v3 normal = v3_normalize(interpolate_normal(P, A, B, C, nrm(sc, 3 * (uint64_t)tri), nrm(sc, 3 * (uint64_t)tri + 1), nrm(sc, 3 * (uint64_t)tri + 2)));
            const orc_material *m = &sc->materials[sc->tri_material[tri]];
            if (m->flags & 1u) {                                                             /* :49-91 */
                float ri = m->ni;
                int goingOut = 0;
                if (v3_dot(normal, rd) > 0.0f) { normal = v3_scale(normal, -1.0f); goingOut = 1; ri = trk_previous(tc->trk); }
                int tir;
                v3 rdir = get_refracted_ray(rd, normal, trk_current(tc->trk), ri, &tir);
                v3 rorg = v3_add(P, v3_scale(rdir, (float)0.001));
                if (tir) {
                    refr = v3_scale(trace_ray(tc, rorg, rdir, bounces - 1, bounceIndex + 1, lastNormal, isSpecular, energy * (float)0.95), energy);
                } else {
                    if (goingOut) trk_pop(tc->trk); else trk_update(tc->trk, ri);
                    refr = v3_scale(trace_ray(tc, rorg, rdir, bounces - 1, bounceIndex, lastNormal, isSpecular, energy * (float)0.95), energy);
                }
            }
            float doppler = 1.0f, grav = 1.0f;
            if ((m->flags & 2u) && sc->bh.present) {                                         /* :95-113 */
                v3 rel = v3_sub(P, sc->bh.pos);
                v3 spin = { 0, 1, 0 };
                v3 tang = v3_normalize(v3_cross(spin, rel));
                v3 vel = v3_scale(tang, (float)0.9999);
                v3 lightDir = v3_scale(rd, -1.0f);
                float vpar = v3_dot(lightDir, vel);
                doppler = sqrtf((1.0f + vpar) / (1.0f - vpar));
                grav = vt_initial / st.vt;
                if (sc->bh.has_disk) diffuse = disk_color(&sc->bh, P, sc->bh.pos);
            }
            float reflectivity = (m->ks[0] + m->ks[1] + m->ks[2]) / 3.0f;                    /* :116 */
            int take_reflective;
            if (reflectivity < (float)0.1) take_reflective = 0;
            else if (reflectivity < (float)0.9) take_reflective = rng_f32(tc->rng) < reflectivity;   /* :163 */
            else take_reflective = 1;
            if (take_reflective) {
                reflective = v3_add(handle_reflective(tc, rd, bounces, tri, P, normal, bounceIndex, energy), refr);   /* :165,:211 */
            } else {
                int ie;
                v3 dc = handle_diffuse(tc, ro, rd, bounces, tri, P, normal, bounceIndex, energy, &ie);
                if (ie && !isSpecular) dc = mis_on_emitter(tc, dc, ro, rd, tri, P, normal, lastNormal);
                diffuse = dc;
            }
            float diffuseScale = 1.0f;                                                       /* :218-223 */
            if (v3_length(refr) > 0.0f) diffuseScale = (float)0.1;
            v3 final = v3_add(v3_add(v3_add(v3_scale(diffuse, diffuseScale), specular), reflective), refr);
            return v3_scale(final, doppler * grav);
        }
        if (!curved) {                                                                       /* :226-228 */
            rayPosition = v3_add(rayPosition, v3_scale(rd, tc->p.step_size));
            ro = rayPosition;
        } else {                                                                             /* :229-256 */
            st = rk4_step(&st, tc->p.step_size, &sc->bh);
            v3 rel = { st.px, st.py, st.pz };
            v3 np = v3_add(rel, sc->bh.pos);
            v3 vel = { st.vx, st.vy, st.vz };
            rd = v3_normalize(vel);
            ro = np; rayPosition = np;
            if (v3_dot(rel, rel) <= sc->bh.rs * sc->bh.rs) return zero;
        }
    }
    if (sc->sky.kind == 0) return zero;                                                      /* :260-263 */
    return sky_sample(&sc->sky, rd);
}

/* ------------------------------------------------------------------ scene construction & render */

orc_scene *orc_scene_create(const orc_bvh *bvh, const float *verts, const uint32_t *tri_idx, uint64_t ntris,
                            const float *normals, const uint32_t *tri_material, const float *materials, uint64_t nmat,
                            const float *lights, uint64_t nlights, const float *sky, const uint16_t *sky_img, int sky_w,
                            int sky_h, const float *blackhole) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->bvh = bvh; s->verts = verts; s->tri_idx = tri_idx; s->ntris = ntris; s->normals = normals; s->tri_material = tri_material;
    s->nmaterials = nmat;
    s->materials = (orc_material *)calloc(nmat ? nmat : 1, sizeof(orc_material));
    for (uint64_t i = 0; i < nmat; ++i) {           /* PtMaterial rows (16 words): Kd Ks Ke Ni Ns, u32 flags */
        const float *r = materials + 16 * i;
        orc_material *m = &s->materials[i];
        memcpy(m->kd, r, 12); memcpy(m->ks, r + 3, 12); memcpy(m->ke, r + 6, 12);
        m->ni = r[9]; m->ns = r[10]; memcpy(&m->flags, r + 11, 4);
        memcpy(&m->diffuse_map, r + 12, 4); memcpy(&m->bump_map, r + 13, 4);
    }
    s->emissive = (uint32_t *)malloc(sizeof(uint32_t) * (ntris ? ntris : 1));
    for (uint64_t t = 0; t < ntris; ++t)            /* renderer.go:311-321 */
        if (is_emissive(&s->materials[tri_material[t]])) s->emissive[s->nemissive++] = (uint32_t)t;
    s->nlights = nlights;
    s->lights = (orc_light *)calloc(nlights ? nlights : 1, sizeof(orc_light));
    for (uint64_t i = 0; i < nlights; ++i) {        /* rows of 12 floats: kind, colour, dir/pos, intensity */
        const float *r = lights + 12 * i;
        memcpy(&s->lights[i].kind, r, 4); s->lights[i].color = v3f(r + 1); s->lights[i].vec = v3f(r + 4); s->lights[i].intensity = r[7];
    }
    memcpy(&s->sky.kind, sky, 4); s->sky.horizon = v3f(sky + 1); s->sky.zenith = v3f(sky + 4); s->sky.ground = v3f(sky + 7);
    s->sky.intensity = sky[10]; s->sky.img = sky_img; s->sky.img_w = sky_w; s->sky.img_h = sky_h;
    if (s->sky.kind == 3 && (!sky_img || sky_w < 1 || sky_h < 1)) s->sky.kind = 0;
    if (blackhole) {
        s->bh.present = 1; s->bh.pos = v3f(blackhole); s->bh.rs = blackhole[3]; { int32_t hd; memcpy(&hd, blackhole + 4, 4); s->bh.has_disk = hd != 0; }
        s->bh.inner = blackhole[5]; s->bh.outer = blackhole[6];
    }
    return s;
}
/* vnmu.UVs + the CachedImage table (obj_loader.go:29-57); pixels = all images back to back, borrowed */
void orc_scene_set_textures(orc_scene *s, const float *uvs, uint64_t nimages, const int32_t *widths, const int32_t *heights,
                            const uint32_t *pixels) {
    free(s->images);
    s->uvs = uvs; s->nimages = nimages;
    s->images = (orc_image *)calloc(nimages ? nimages : 1, sizeof(orc_image));
    size_t off = 0;
    for (uint64_t i = 0; i < nimages; ++i) {
        s->images[i].pixels = pixels + off; s->images[i].width = widths[i]; s->images[i].height = heights[i];
        off += (size_t)widths[i] * (size_t)heights[i];
    }
}
void orc_scene_free(orc_scene *s) { if (!s) return; free(s->materials); free(s->emissive); free(s->lights); free(s->images); free(s); }

/* One primary sample: main.go:861-879.  cam = pos, fwd, right, up (xyz each), frustum distance. */
static v3 sample_pixel(const orc_scene *sc, const orc_params *p, const float *cam, int X, int Y, uint32_t sample,
                       orc_trace_counters *cnt) {
    orc_rng rng = rng_init(p->seed, (uint32_t)(Y * p->width + X), sample);
    float rx = ((float)X + rng_f32(&rng)) / (float)p->width;
    float ry = ((float)Y + rng_f32(&rng)) / (float)p->height;
    float px = (rx - 0.5f) * 2.0f, py = (ry - 0.5f) * 2.0f;
    v3 pos = v3f(cam), fwd = v3f(cam + 3), right = v3f(cam + 6), up = v3f(cam + 9);
    v3 rayOrigin = v3_add(v3_add(v3_add(pos, v3_scale(fwd, cam[12])), v3_scale(up, py)), v3_scale(right, px));
    v3 d = { rayOrigin.x - pos.x, rayOrigin.y - pos.y, rayOrigin.z - pos.z };
    d = v3_normalize(d);
    orc_tracker trk; trk.len = 0; trk_update(&trk, 1.0f);
    trace_ctx tc; tc.sc = sc; tc.p = *p; tc.rng = &rng; tc.trk = &trk; tc.cnt = cnt;
    v3 zero = { 0, 0, 0 };
    v3 c = trace_ray(&tc, pos, d, p->bounces, 0, zero, 0, 1.0f);
    cnt->samples++;
    return c;
}

/* Renders rect (x0,y0,w,h), samples [spp_begin, spp_begin+spp_count) of every pixel, ADDING into
 * accum_rgb[w*h*3] / counts[w*h] in sample order (Pixel.AddSample's R,G,B,SampleCount: renderer.go:29,59-61).
 * counters: closest, any, diffuse_events, samples. */
void orc_render(const orc_scene *sc, const float *cam, const orc_params *p, int x0, int y0, int w, int h,
                uint32_t spp_begin, uint32_t spp_count, float *accum_rgb, uint32_t *counts, uint64_t *counters,
                int nthreads) {
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16) reduction(+ : c0, c1, c2, c3)
    for (int64_t i = 0; i < (int64_t)w * h; ++i) {
        int X = x0 + (int)(i % w), Y = y0 + (int)(i / w);
        orc_trace_counters cnt; memset(&cnt, 0, sizeof cnt);
        for (uint32_t s = 0; s < spp_count; ++s) {
            v3 c = sample_pixel(sc, p, cam, X, Y, spp_begin + s, &cnt);
            accum_rgb[3 * i] += c.x; accum_rgb[3 * i + 1] += c.y; accum_rgb[3 * i + 2] += c.z;
            counts[i] += 1;
        }
        c0 += cnt.closest; c1 += cnt.any; c2 += cnt.diffuse_events; c3 += cnt.samples;
    }
    if (counters) { counters[0] += c0; counters[1] += c1; counters[2] += c2; counters[3] += c3; }
}

/* The goroutine tile renderer itself (main.go:435, :454-491, :832-899, :935-937), for the CPU baseline: the image is
 * split into splits_x x splits_y tiles of (W / splits_x) x (H / splits_y) pixels (main.go:454-457, :474-482; pixels beyond
 * the last full tile are not rendered, as in the reference) and ONE worker thread is started per tile -- the reference's
 * threadCount goroutines, which cover the image only when threadCount == splitsX * splitsY.  A worker walks its tile in
 * the effective schedule of GetNoisiestPixel at the committed constants (row-major, every pixel picked once and given
 * `spp` samples: tile.go:34-66 with samplesPerPixel == maxSamplesPerPixel) and adds the samples as
 * Pixel.AddSample does.  accum_rgb / counts are full frames [H][W].  Returns the wall-clock seconds from starting the
 * workers to joining the last one (the reference times nothing else either: main.go:935-937 -> :904). */
typedef struct { const orc_scene *sc; const float *cam; const orc_params *p; int x0, y0, w, h; uint32_t spp;
                 float *accum; uint32_t *counts; orc_trace_counters cnt; } tile_job;
static void *tile_worker(void *arg) {
    tile_job *j = (tile_job *)arg;
    memset(&j->cnt, 0, sizeof j->cnt);
    const int W = j->p->width;
    for (int y = 0; y < j->h; ++y)
        for (int x = 0; x < j->w; ++x) {
            const int X = j->x0 + x, Y = j->y0 + y;
            const size_t i = (size_t)Y * W + X;
            for (uint32_t s = 0; s < j->spp; ++s) {                                          /* main.go:861-879 */
                v3 c = sample_pixel(j->sc, j->p, j->cam, X, Y, s, &j->cnt);
                j->accum[3 * i] += c.x; j->accum[3 * i + 1] += c.y; j->accum[3 * i + 2] += c.z;
                j->counts[i] += 1;
            }
        }
    return NULL;
}
double orc_render_tile_workers(const orc_scene *sc, const float *cam, const orc_params *p, int splits_x, int splits_y,
                               uint32_t spp, float *accum_rgb, uint32_t *counts, uint64_t *counters) {
    if (splits_x < 1 || splits_y < 1) return -1.0;
    const int nt = splits_x * splits_y, tw = p->width / splits_x, th = p->height / splits_y;
    tile_job *jobs = (tile_job *)calloc((size_t)nt, sizeof(tile_job));
    pthread_t *thr = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int started = 0;
    for (int i = 0; i < nt; ++i) {
        tile_job *j = &jobs[i];
        j->sc = sc; j->cam = cam; j->p = p; j->x0 = (i % splits_x) * tw; j->y0 = (i / splits_x) * th; j->w = tw; j->h = th;
        j->spp = spp; j->accum = accum_rgb; j->counts = counts;
        if (pthread_create(&thr[i], NULL, tile_worker, j) != 0) break;
        ++started;
    }
    for (int i = 0; i < started; ++i) pthread_join(thr[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (counters)
        for (int i = 0; i < started; ++i) {
            counters[0] += jobs[i].cnt.closest; counters[1] += jobs[i].cnt.any; counters[2] += jobs[i].cnt.diffuse_events;
            counters[3] += jobs[i].cnt.samples;
        }
    free(jobs); free(thr);
    if (started != nt) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The adaptive worker loop as the reference runs it when maxSamplesPerPixel > samplesPerPixel (main.go:837-897): pick
 * the tile's noisiest pixel with Tile.GetNoisiestPixel (tile.go:34-66), give it samplesPerPixel samples, each through
 * Pixel.AddSample (renderer.go:25-82: the sums, the count, Welford's mean / M2 / variance and the luminance range behind
 * Pixel.Contrast), until the
 * picker returns nil.  Sequential over ONE tile, exactly in the reference's order.  A sample's radiance comes from
 * `get(user, pixel, sample index)`: the pixel's SampleCount at that moment indexes the counter-based RNG,
 * so the k-th sample a pixel ever gets is the same whenever it is taken.  Returns the number of picks. */
typedef v3 (*pick_sample_fn)(void *user, int64_t pixel, uint32_t sample);
enum { ORC_PIXEL_STATS = 9 };          /* per pixel: MinLuminance, MaxLuminance, Mean[3], M2[3], Variance (renderer.go:12-22) */
static int64_t pick_loop(int64_t P, int spp, int max_samples, pick_sample_fn get, void *user, float *sums, uint32_t *counts,
                         float *stats) {
    int64_t picks = 0;
    for (;;) {
        int64_t best = -1;                                                     /* tile.go:35-36 */
        float max_priority = -1.0f;
        for (int64_t i = 0; i < P; ++i) {                                      /* :38 */
            const int n = (int)counts[i];
            if (n >= max_samples) continue;                                    /* :39-41 */
            float priority;
            if (n < 4) priority = 1e9f;                                        /* :45-46 */
            else {
                const float contrast = stats[ORC_PIXEL_STATS * i + 1] - stats[ORC_PIXEL_STATS * i];   /* Pixel.Contrast, renderer.go:81 */
                if (n > 32 && contrast < 0.01f) continue;                      /* :51-53 */
                priority = contrast / sqrtf((float)n);                         /* :56 */
            }
            if (best < 0 || priority > max_priority || (priority == max_priority && n < (int)counts[best])) {   /* :60 */
                best = i; max_priority = priority;
            }
        }
        if (best < 0) return picks;                                            /* main.go:849-852 */
        ++picks;
        for (int k = 0; k < spp; ++k) {                                        /* main.go:861 */
            const v3 c = get(user, best, counts[best]);
            counts[best] += 1;                                                 /* renderer.go:29 */
            sums[3 * best] += c.x; sums[3 * best + 1] += c.y; sums[3 * best + 2] += c.z;   /* :59-61 */
            float *st = stats + ORC_PIXEL_STATS * best;                         /* min, max luminance; Mean; M2; Variance */
            const float n = (float)counts[best];                               /* :30 */
            const float dx = c.x - st[2], dy = c.y - st[3], dz = c.z - st[4];  /* :33-37 */
            st[2] += dx / n; st[3] += dy / n; st[4] += dz / n;                 /* :39-41 */
            st[5] += dx * (c.x - st[2]); st[6] += dy * (c.y - st[3]); st[7] += dz * (c.z - st[4]);   /* :43-51 */
            if (counts[best] > 1) st[8] = ((st[5] + st[6]) + st[7]) / (3.0f * (n - 1.0f));           /* :53-57 */
            const float lum = (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z;  /* :84-86 */
            if (counts[best] == 1) { st[0] = lum; st[1] = lum; }               /* :67-70 */
            else {
                if (lum < st[0]) st[0] = lum;                                  /* :72-77 */
                if (lum > st[1]) st[1] = lum;
            }
            if ((int)counts[best] > max_samples) break;                        /* main.go:880-882 */
        }
    }
}
/* ... on given samples: samples[(pixel * S + s) * 3], S > max_samples (a pixel can end one sample over, main.go:880). */
typedef struct { const float *samples; int64_t S; } given_samples;
static v3 given_sample(void *user, int64_t pixel, uint32_t s) {
    const given_samples *g = (const given_samples *)user;
    const float *c = g->samples + 3 * (pixel * g->S + (int64_t)(s < g->S ? s : g->S - 1));
    v3 r = { c[0], c[1], c[2] };
    return r;
}
int64_t orc_pick_simulate(const float *samples, int64_t P, int64_t S, int spp, int max_samples, float *sums, uint32_t *counts,
                          float *stats) {
    if (S <= max_samples || spp < 1) return -1;
    given_samples g = { samples, S };
    return pick_loop(P, spp, max_samples, given_sample, &g, sums, counts, stats);
}
/* ... and on the scene itself: tile (x0, y0, w, h) of the frame, pixels in the tile's row-major order (main.go:474-482);
 * sums / counts / stats are the TILE's own [h][w] arrays (stats: ORC_PIXEL_STATS floats per pixel). */
typedef struct { const orc_scene *sc; const float *cam; const orc_params *p; int x0, y0, w; orc_trace_counters cnt; } traced_samples;
static v3 traced_sample(void *user, int64_t pixel, uint32_t s) {
    traced_samples *t = (traced_samples *)user;
    return sample_pixel(t->sc, t->p, t->cam, t->x0 + (int)(pixel % t->w), t->y0 + (int)(pixel / t->w), s, &t->cnt);
}
int64_t orc_render_tile_adaptive(const orc_scene *sc, const float *cam, const orc_params *p, int x0, int y0, int w, int h,
                                 int spp, int max_samples, float *sums, uint32_t *counts, float *stats) {
    if (spp < 1 || w < 1 || h < 1) return -1;
    traced_samples t; memset(&t, 0, sizeof t);
    t.sc = sc; t.cam = cam; t.p = p; t.x0 = x0; t.y0 = y0; t.w = w;
    return pick_loop((int64_t)w * h, spp, max_samples, traced_sample, &t, sums, counts, stats);
}

/* Per-sample radiance (no accumulation), for debugging / fine-grained comparison: out[w*h*spp_count*3],
 * sample-major inside a pixel. */
void orc_render_samples(const orc_scene *sc, const float *cam, const orc_params *p, int x0, int y0, int w, int h,
                        uint32_t spp_begin, uint32_t spp_count, float *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)w * h; ++i) {
        int X = x0 + (int)(i % w), Y = y0 + (int)(i / w);
        orc_trace_counters cnt; memset(&cnt, 0, sizeof cnt);
        for (uint32_t s = 0; s < spp_count; ++s) {
            v3 c = sample_pixel(sc, p, cam, X, Y, spp_begin + s, &cnt);
            float *o = out + 3 * ((size_t)i * spp_count + s);
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
        }
    }
}

/* vec3.go:95-106 ToRGBA + main.go:886-896 (mean, horizontal flip: img.Set(width - X, Y); X = 0 is dropped and
 * image column 0 is never written).  rgba: width*height*4 bytes, pre-filled by the caller. */
void orc_to_rgba(const float *accum_rgb, const uint32_t *counts, int width, int height, uint8_t *rgba) {
    for (int Y = 0; Y < height; ++Y)
        for (int X = 0; X < width; ++X) {
            size_t i = (size_t)Y * width + X;
            int ox = width - X;
            if (ox < 0 || ox >= width || counts[i] == 0) continue;
            float n = (float)counts[i];
            float c[3] = { accum_rgb[3 * i] / n, accum_rgb[3 * i + 1] / n, accum_rgb[3 * i + 2] / n };
            uint8_t *o = rgba + 4 * ((size_t)Y * width + ox);
            for (int k = 0; k < 3; ++k) {
                float s = sqrtf(gomaxf(0.0f, c[k]));
                float v = gomaxf(0.0f, gominf(1.0f, s)) * 255.0f;       /* Clamp01 (utils.go:238-240) */
                o[k] = (v == v) ? (uint8_t)v : 0;
            }
            o[3] = 255;
        }
}
