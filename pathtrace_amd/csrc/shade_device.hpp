// shade_device.hpp -- gfx950 device code for the integrator's shading math.
//
// Restates, in the reference's operand order and with FMA contraction off, the float32 helpers
// TraceRay uses: vec3.go, utils.go:127-307, renderer.go:326-336, skybox.go, BlackHole.go.
// Transcendentals (sin/cos/atan/atan2/acos/pow) come from the device math library, so images
// match the CPU oracle to a tolerance, not bitwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptmi.h"
#include "trav_device.hpp"

namespace ptmi {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 mk(const float *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }       // vec3.go:22
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }       // vec3.go:32
__device__ __forceinline__ V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }        // vec3.go:42
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }    // vec3.go:52
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }                          // vec3.go:56
__device__ __forceinline__ V3 normalize(V3 a) {                                                     // vec3.go:60
    float len = length(a);
    if (len == 0.0f) return mk(0, 0, 0);
    return scale(a, 1.0f / len);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                   // vec3.go:79
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }      // vec3.go:108
__device__ __forceinline__ V3 lerp(V3 a, V3 b, float t) { return add(a, scale(sub(b, a), t)); }     // vec3.go:134

#define PT_PI32 ((float)3.14159265358979323846)

// ---- counter-based RNG shared with the CPU oracle (replaces Go's global math/rand) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
struct Rng { uint32_t key, ctr; };
__device__ __forceinline__ Rng rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    r.key = mix32(mix32(mix32(seed ^ 0xA511E9B3u) + pixel * 0x9E3779B9u) ^ (sample * 0x85EBCA6Bu + 0x27D4EB2Fu));
    r.ctr = 0;
    return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng &r) { return mix32(r.key + 0x9E3779B9u * (r.ctr++)); }
__device__ __forceinline__ float rng_f32(Rng &r) { return (float)(rng_u32(r) >> 8) * (float)(1.0 / 16777216.0); }
__device__ __forceinline__ uint32_t rng_intn(Rng &r, uint32_t n) {
    return (uint32_t)(((unsigned long long)rng_u32(r) * (unsigned long long)n) >> 32);
}

// utils.go:127-151
__device__ __forceinline__ V3 interpolate_normal(V3 p, V3 a, V3 b, V3 c, V3 nA, V3 nB, V3 nC) {
    V3 v0 = sub(b, a), v1 = sub(c, a), v2 = sub(p, a);
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    nA = scale(nA, u); nB = scale(nB, v); nC = scale(nC, w);
    nA = add(nA, nB); nA = add(nA, nC);
    return nA;
}
// utils.go:217-229
__device__ __forceinline__ V3 sample_triangle_point(Rng &r, V3 A, V3 B, V3 C) {
    float u = rng_f32(r);
    float v = rng_f32(r) * (1.0f - u);
    float w = 1.0f - u - v;
    A = scale(A, w); B = scale(B, v); C = scale(C, u);
    A = add(A, B); A = add(A, C);
    return A;
}
// utils.go:231-236
__device__ __forceinline__ float triangle_area(V3 A, V3 B, V3 C) { return 0.5f * length(cross(sub(B, A), sub(C, A))); }

// utils.go:242-253, tracer.go:625-634
__device__ __forceinline__ void tangent_frame(V3 n, V3 &t1, V3 &t2) {
    V3 up = fabsf(n.y) < (float)0.999 ? mk(0, 1, 0) : mk(1, 0, 0);
    t1 = normalize(cross(n, up));
    t2 = cross(n, t1);
}
// utils.go:255-270 / 273-290
__device__ __forceinline__ V3 cosine_sample2(Rng &r, V3 n, V3 t1, V3 t2) {
    float u1 = rng_f32(r), u2 = rng_f32(r);
    float rr = sqrtf(u1);
    float theta = (float)(2.0 * 3.14159265358979323846) * u2;
    float x = rr * cosf(theta), y = rr * sinf(theta);
    float z = sqrtf(gomaxf(0.0f, 1.0f - u1));
    V3 d = scale(t1, x);
    d = add(d, scale(t2, y));
    d = add(d, scale(n, z));
    return normalize(d);
}
// utils.go:292-294
__device__ __forceinline__ V3 reflect_v(V3 V, V3 N) { return sub(V, scale(N, 2.0f * dot(V, N))); }
// utils.go:296-307
__device__ __forceinline__ V3 get_refracted_ray(V3 inDir, V3 inNormal, float n1, float n2, bool &tir) {
    V3 A = normalize(inDir), N = normalize(inNormal);
    float R = n1 / n2;
    float c = -dot(A, N);
    float D = 1.0f - R * R * (1.0f - c * c);
    if (D < 0.0f) { tir = true; return reflect_v(inDir, inNormal); }
    tir = false;
    return normalize(add(scale(A, R), scale(N, c * R - sqrtf(D))));
}
// renderer.go:326-336
__device__ __forceinline__ float mis_weight(float pdf1, float pdf2) {
    if (pdf1 <= 0.0f) return 0.0f;
    if (pdf2 <= 0.0f) return 1.0f;
    float p1 = pdf1 * pdf1, p2 = pdf2 * pdf2;
    return p1 / (p1 + p2);
}
// tracer.go:724-747
__device__ __forceinline__ V3 sample_glossy(Rng &r, V3 reflDir, float roughness) {
    V3 up = fabsf(reflDir.y) < (float)0.9 ? mk(0, 1, 0) : mk(1, 0, 0);
    V3 t1 = normalize(cross(reflDir, up));
    V3 t2 = normalize(cross(reflDir, t1));
    float theta = rng_f32(r) * 2.0f * PT_PI32;
    float alpha = roughness * roughness;
    float u = rng_f32(r);
    float phi = atanf(alpha * sqrtf(u) / sqrtf(1.0f - u));
    float x = cosf(theta) * sinf(phi), y = sinf(theta) * sinf(phi), z = cosf(phi);
    return normalize(add(add(scale(t1, x), scale(t2, y)), scale(reflDir, z)));
}

// skybox.go:23-25, 36-42, 70-107
__device__ __forceinline__ V3 sky_sample(const PtSky &s, const uint16_t *img, V3 dir) {
    if (s.kind == PTMI_SKY_SOLID) return mk(s.horizon);
    if (s.kind == PTMI_SKY_GRADIENT) {
        float angle = dot(dir, mk(0, 1, 0));
        if (angle < 0.0f) return mk(s.ground);
        return scale(add(scale(mk(s.horizon), 1.0f - angle), scale(mk(s.zenith), angle)), s.intensity);
    }
    if (s.kind == PTMI_SKY_IMAGE) {
        V3 d = normalize(dir);
        float phi = atan2f(d.z, d.x), theta = acosf(d.y);
        float u = (phi + PT_PI32) / (2.0f * PT_PI32), v = theta / PT_PI32;
        float fx = u * (float)(s.image_width - 1), fy = v * (float)(s.image_height - 1);
        int x = (fx == fx) ? (int)fx : 0, y = (fy == fy) ? (int)fy : 0;
        x = x < 0 ? 0 : (x > s.image_width - 1 ? s.image_width - 1 : x);
        y = y < 0 ? 0 : (y > s.image_height - 1 ? s.image_height - 1 : y);
        const uint16_t *p = img + 3 * ((size_t)y * s.image_width + x);
        return scale(mk((float)p[0] / 65535.0f, (float)p[1] / 65535.0f, (float)p[2] / 65535.0f), s.intensity);
    }
    return mk(0, 0, 0);
}

// ---- BlackHole.go ------------------------------------------------------------------------------
struct RayState { float px, py, pz, vx, vy, vz, pt, vt; };

// BlackHole.go:24-69
__device__ __forceinline__ RayState get_initial_state(V3 position, V3 direction, const PtBlackHole &bh) {
    RayState z = {0, 0, 0, 0, 0, 0, 0, 0};
    V3 rel = sub(position, mk(bh.position));
    float r = length(rel);
    if (r == 0.0f) return z;
    float rs = bh.rs;
    float p_dot_v = dot(rel, direction), v_dot_v = dot(direction, direction);
    float A = rs / r - 1.0f;
    float B = 2.0f * rs * p_dot_v / (r * r);
    float C = v_dot_v + rs * p_dot_v * p_dot_v / (r * r * r);
    float disc = B * B - 4.0f * A * C;
    if (disc < 0.0f) return z;
    RayState s;
    s.vt = (-B + sqrtf(disc)) / (2.0f * A);
    s.pt = 0.0f; s.px = rel.x; s.py = rel.y; s.pz = rel.z; s.vx = direction.x; s.vy = direction.y; s.vz = direction.z;
    return s;
}
// The correctly rounded square root and reciprocal of an argument that is nowhere near the ends of the exponent range
// (MID_LO < x < MID_HI for the root, MID_LO^1.5 .. MID_HI^1.5 for the reciprocal): the instruction sequences the
// compiler emits for sqrtf(x) and 1.0f / x without the operand scaling for denormal results and without the fix-up of
// zero / infinite / NaN operands, neither of which can apply in that range -- 9 and 7 VALU instructions instead of 16
// and 11.  Equal to sqrtf / the division bit for bit on every float of the range (ptmi_selftest_math runs them all).
constexpr float MID_LO = 0x1p-40f, MID_HI = 0x1p40f;
#ifndef PTMI_MID_MATH
#define PTMI_MID_MATH 1          // 0: the compiler's general sequences everywhere (A/B builds, tools/build_variant.sh)
#endif
__device__ __forceinline__ float sqrt_mid(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);                         // within 1 ulp
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);          // its neighbours
    const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
    float r = vp <= 0.0f ? sd : s;
    r = vs > 0.0f ? su : r;
    return r;
}
__device__ __forceinline__ float rcp_mid(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = r;                                                       // 1.0f * r
    q = __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
}

// BlackHole.go:74-110
// r_sq = (px * px + py * py) + pz * pz of `s`, which the caller may already hold (the march's horizon test is that sum)
__device__ __forceinline__ void get_acceleration(const RayState &s, float rs, float r_sq, float &at, float &ax, float &ay, float &az) {
    float px = s.px, py = s.py, pz = s.pz, vx = s.vx, vy = s.vy, vz = s.vz;
    // one unsigned compare for max(rs^2, MID_LO) < r_sq < MID_HI (r_sq is a sum of squares: its sign bit is clear, and
    // the bit patterns of such floats order like their values; NaN and infinity lie above MID_HI)
    const float lo = fmaxf(rs * rs, MID_LO);
    const uint32_t span = lo < MID_HI ? __float_as_uint(MID_HI) - __float_as_uint(lo) - 1u : 0u;   // (uniform: the compiler hoists it out of the march)
    // every lane runs the short forms (for an argument outside their range the value is simply not used); the lanes
    // outside -- inside the horizon, or at the ends of the exponent range -- are the rare ones and take the branch
    float r = sqrt_mid(r_sq);
    float r_cubed = r_sq * r;
    float inv_r_cubed = rcp_mid(r_cubed);
    bool inside = false;
    if (!(PTMI_MID_MATH && __float_as_uint(r_sq) - __float_as_uint(lo) - 1u < span)) {
        inside = r_sq <= rs * rs;
        r = sqrtf(r_sq);
        r_cubed = r_sq * r;
        inv_r_cubed = 1.0f / r_cubed;
    }
    float x_dot_v = (px * vx + py * vy) + pz * vz;
    float x_dot_v_sq = x_dot_v * x_dot_v;
    float rs_x_dot_v = rs * x_dot_v;
    float two = 2.0f * x_dot_v * inv_r_cubed;
    at = -2.0f * rs_x_dot_v * inv_r_cubed;
    float factor = (1.0f - 3.0f * rs * x_dot_v_sq * inv_r_cubed) * inv_r_cubed;
    float rs_inv_factor = -rs * factor;
    ax = px * rs_inv_factor + rs * two * vx;
    ay = py * rs_inv_factor + rs * two * vy;
    az = pz * rs_inv_factor + rs * two * vz;
    if (inside) { at = ax = ay = az = 0.0f; }                          // BlackHole.go:82-84
}
struct Rk4D { float at, ax, ay, az, vt, vx, vy, vz; };
// BlackHole.go:118-127
__device__ __forceinline__ RayState evaluate_rk4(const RayState &s, float dt, const Rk4D &k) {
    RayState o;
    o.pt = s.pt + k.vt * dt; o.px = s.px + k.vx * dt; o.py = s.py + k.vy * dt; o.pz = s.pz + k.vz * dt;
    o.vt = s.vt + k.at * dt; o.vx = s.vx + k.ax * dt; o.vy = s.vy + k.ay * dt; o.vz = s.vz + k.az * dt;
    return o;
}
// BlackHole.go:131-163
__device__ __forceinline__ float state_r_sq(const RayState &s) { return (s.px * s.px + s.py * s.py) + s.pz * s.pz; }   // BlackHole.go:80
__device__ __forceinline__ void get_acceleration(const RayState &s, float rs, float &at, float &ax, float &ay, float &az) {
    get_acceleration(s, rs, state_r_sq(s), at, ax, ay, az);
}
// s_r_sq = state_r_sq(s)
__device__ __forceinline__ RayState rk4_step(const RayState &s, float s_r_sq, float h, float rs) {
    Rk4D k1, k2, k3, k4;
    get_acceleration(s, rs, s_r_sq, k1.at, k1.ax, k1.ay, k1.az);
    k1.vt = s.vt; k1.vx = s.vx; k1.vy = s.vy; k1.vz = s.vz;
    RayState tmp = evaluate_rk4(s, h * 0.5f, k1);
    get_acceleration(tmp, rs, k2.at, k2.ax, k2.ay, k2.az);
    k2.vt = tmp.vt; k2.vx = tmp.vx; k2.vy = tmp.vy; k2.vz = tmp.vz;
    tmp = evaluate_rk4(s, h * 0.5f, k2);
    get_acceleration(tmp, rs, k3.at, k3.ax, k3.ay, k3.az);
    k3.vt = tmp.vt; k3.vx = tmp.vx; k3.vy = tmp.vy; k3.vz = tmp.vz;
    tmp = evaluate_rk4(s, h, k3);
    get_acceleration(tmp, rs, k4.at, k4.ax, k4.ay, k4.az);
    k4.vt = tmp.vt; k4.vx = tmp.vx; k4.vy = tmp.vy; k4.vz = tmp.vz;
    float h6 = h / 6.0f;
    RayState o;
    o.pt = s.pt + h6 * (((k1.vt + 2.0f * k2.vt) + 2.0f * k3.vt) + k4.vt);
    o.px = s.px + h6 * (((k1.vx + 2.0f * k2.vx) + 2.0f * k3.vx) + k4.vx);
    o.py = s.py + h6 * (((k1.vy + 2.0f * k2.vy) + 2.0f * k3.vy) + k4.vy);
    o.pz = s.pz + h6 * (((k1.vz + 2.0f * k2.vz) + 2.0f * k3.vz) + k4.vz);
    o.vt = s.vt + h6 * (((k1.at + 2.0f * k2.at) + 2.0f * k3.at) + k4.at);
    o.vx = s.vx + h6 * (((k1.ax + 2.0f * k2.ax) + 2.0f * k3.ax) + k4.ax);
    o.vy = s.vy + h6 * (((k1.ay + 2.0f * k2.ay) + 2.0f * k3.ay) + k4.ay);
    o.vz = s.vz + h6 * (((k1.az + 2.0f * k2.az) + 2.0f * k3.az) + k4.az);
    return o;
}

// Gradient noise standing in for go-perlin's Noise2D(alpha=2, beta=2, n=4); same lattice hash as the oracle.
__device__ __forceinline__ double grad2(int ix, int iy, double fx, double fy) {
    uint32_t h = mix32((uint32_t)ix * 0x9E3779B9u ^ mix32((uint32_t)iy * 0x85EBCA6Bu + 0x165667B1u)) & 7u;
    const double q = 0.70710678118654752;
    double gx = h == 0 ? 1.0 : h == 1 ? -1.0 : h < 4 ? 0.0 : ((h & 1) ? -q : q);
    double gy = h < 2 ? 0.0 : h == 2 ? 1.0 : h == 3 ? -1.0 : (h < 6 ? q : -q);
    return gx * fx + gy * fy;
}
__device__ __forceinline__ double noise2(double x, double y) {
    double flx = floor(x), fly = floor(y);
    int ix = (int)flx, iy = (int)fly;
    double fx = x - flx, fy = y - fly;
    double sx = fx * fx * (3.0 - 2.0 * fx), sy = fy * fy * (3.0 - 2.0 * fy);
    double a = grad2(ix, iy, fx, fy), b = grad2(ix + 1, iy, fx - 1.0, fy);
    double c = grad2(ix, iy + 1, fx, fy - 1.0), d = grad2(ix + 1, iy + 1, fx - 1.0, fy - 1.0);
    double u = a + sx * (b - a), v = c + sx * (d - c);
    return u + sy * (v - u);
}
__device__ __forceinline__ double perlin_noise2d(double x, double y) {
    double sc = 1.0, sum = 0.0;
    for (int i = 0; i < 4; ++i) { sum += noise2(x, y) / sc; sc *= 2.0; x *= 2.0; y *= 2.0; }
    return sum;
}
// BlackHole.go:174-242
__device__ __forceinline__ V3 disk_color(const PtBlackHole &bh, V3 world) {
    V3 rel = sub(world, mk(bh.position));
    float radius = sqrtf(rel.x * rel.x + rel.z * rel.z);
    float angle = atan2f(rel.z, rel.x);
    float t = (radius - bh.disk_inner) / (bh.disk_outer - bh.disk_inner);
    if (t < 0.0f) t = 0.0f;
    if (t > 1.0f) t = 1.0f;
    V3 hot = mk(1.0f, 1.0f, 0.85f), mid = mk(1.0f, 0.6f, 0.0f), cool = mk(0.8f, 0.1f, 0.0f);
    V3 temp = t < 0.5f ? lerp(hot, mid, t * 2.0f) : lerp(mid, cool, (t - 0.5f) * 2.0f);
    float intensity = 1.0f / (t * t + 0.1f);
    temp = scale(temp, intensity);
    float nx = (radius / bh.disk_outer) * 3.0f * 2.0f;
    float ny = (angle / (float)(2.0 * 3.14159265358979323846)) * 3.0f;
    float nv = (float)(perlin_noise2d((double)nx, (double)ny) + 1.0) / 2.0f;
    nv = powf(nv, 5.0f);
    return scale(temp, nv);
}

// ---- textures: obj_loader.go:11-17, renderer.go:122-169, :199-291 -----------------------------------

struct TexDesc { uint32_t offset; int32_t width, height; uint32_t pad; };   // one per texture, offset in texels

// obj_loader.go:11-17
__device__ __forceinline__ float tile_uv(float val) {
    float ip;
    float vf = modff(val, &ip);
    if (vf < 0) vf = 1 + vf;
    return vf;
}

// renderer.go:122-135; returns the packed texel.  NaN / out-of-range coordinates are clamped into the image.
__device__ __forceinline__ uint32_t sample_diffuse_map(const uint32_t *pixels, const TexDesc &t, float x, float y) {
    const float fx = x * (float)(t.width - 1), fy = y * (float)(t.height - 1);
    int px = (fx == fx) ? (int)fx : 0, py = (fy == fy) ? (int)fy : 0;
    px = px < 0 ? 0 : (px > t.width - 1 ? t.width - 1 : px);
    py = py < 0 ? 0 : (py > t.height - 1 ? t.height - 1 : py);
    return pixels[(size_t)t.offset + (size_t)py * (size_t)t.width + (size_t)px];
}

__device__ __forceinline__ float texel_height(uint32_t p) {     // renderer.go:152-154
    const float R = (float)(p & 255u), G = (float)((p >> 8) & 255u), B = (float)((p >> 16) & 255u);
    return (((float)0.299 * R + (float)0.587 * G) + (float)0.114 * B) / 255.0f;
}

// renderer.go:137-169
__device__ __forceinline__ V3 sample_bump_map(const uint32_t *pixels, const TexDesc &t, float x, float y, float strength) {
    const float w = (float)t.width, h = (float)t.height;
    const float offsetX = 1.0f / w, offsetY = 1.0f / h;
    const float centerHeight = texel_height(sample_diffuse_map(pixels, t, x, y));
    const float rightHeight = texel_height(sample_diffuse_map(pixels, t, fmodf(x + offsetX, 1.0f), y));
    const float upHeight = texel_height(sample_diffuse_map(pixels, t, x, fmodf(y + offsetY, 1.0f)));
    const float dx = (rightHeight - centerHeight) * strength;
    const float dy = (upHeight - centerHeight) * strength;
    return normalize(mk(-dx, -dy, 1.0f));
}

// renderer.go:199-291; uv = the six UV floats of the triangle
__device__ __forceinline__ V3 transform_normal_to_world(V3 tn, V3 wn, V3 A, V3 B, V3 C, const float *uv) {
    if (tn.x == 0 && tn.y == 0 && tn.z == 1) return wn;
    const V3 edge1 = sub(B, A), edge2 = sub(C, A);
    const float d1x = uv[2] - uv[0], d1y = uv[3] - uv[1], d2x = uv[4] - uv[0], d2y = uv[5] - uv[1];
    const float det = d1x * d2y - d2x * d1y;
    V3 tangent, bitangent;
    if (fabsf(det) < (float)1e-6) {
        V3 up = mk(0, 1, 0);
        if (!(fabsf(wn.y) < (float)0.9)) up = mk(1, 0, 0);
        tangent = normalize(cross(wn, up));
        bitangent = normalize(cross(wn, tangent));
    } else {
        const float invDet = 1.0f / det;
        tangent = mk(invDet * (d2y * edge1.x - d1y * edge2.x), invDet * (d2y * edge1.y - d1y * edge2.y),
                     invDet * (d2y * edge1.z - d1y * edge2.z));
        bitangent = mk(invDet * (-d2x * edge1.x + d1x * edge2.x), invDet * (-d2x * edge1.y + d1x * edge2.y),
                       invDet * (-d2x * edge1.z + d1x * edge2.z));
        tangent = normalize(sub(tangent, scale(wn, dot(tangent, wn))));
        const V3 calc = cross(wn, tangent);
        if (dot(bitangent, calc) < 0) bitangent = normalize(scale(calc, -1.0f));
        else bitangent = normalize(calc);
    }
    return normalize(mk((tn.x * tangent.x + tn.y * bitangent.x) + tn.z * wn.x,
                        (tn.x * tangent.y + tn.y * bitangent.y) + tn.z * wn.y,
                        (tn.x * tangent.z + tn.y * bitangent.z) + tn.z * wn.z));
}

// tracer.go:468-514: albedo and shading normal of HandleDiffuseMaterial for a textured material
__device__ __forceinline__ void diffuse_surface(const PtMaterial &m, const float *uvs, const TexDesc *tex, uint32_t ntex,
                                                const uint32_t *pixels, uint32_t prim, V3 P, V3 A, V3 B, V3 C,
                                                V3 &albedo, V3 &normal) {
    albedo = mk(m.kd);
    if (!(m.diffuse_map || m.bump_map) || !uvs || !ntex) return;        // Material.HasImage (obj_loader.go:54)
    float uv[6];
    {
        const float2 *p = (const float2 *)(uvs + 6 * (size_t)prim);
        const float2 a = p[0], b = p[1], c = p[2];
        uv[0] = a.x; uv[1] = a.y; uv[2] = b.x; uv[3] = b.y; uv[4] = c.x; uv[5] = c.y;
    }
    const V3 v0 = sub(B, A), v1 = sub(C, A), v2 = sub(P, A);
    const float dot00 = dot(v0, v0), dot01 = dot(v0, v1), dot02 = dot(v0, v2), dot11 = dot(v1, v1), dot12 = dot(v1, v2);
    const float invDenom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    const float u = (dot11 * dot02 - dot01 * dot12) * invDenom;
    const float v = (dot00 * dot12 - dot01 * dot02) * invDenom;
    const float w = 1.0f - u - v;
    const float x = tile_uv((w * uv[0] + u * uv[2]) + v * uv[4]);
    const float y = tile_uv((w * uv[1] + u * uv[3]) + v * uv[5]);
    if (m.diffuse_map && m.diffuse_map <= ntex) {
        const uint32_t c = sample_diffuse_map(pixels, tex[m.diffuse_map - 1], x, y);
        albedo = mk(powf((float)(c & 255u) / 255.0f, (float)2.2), powf((float)((c >> 8) & 255u) / 255.0f, (float)2.2),
                    powf((float)((c >> 16) & 255u) / 255.0f, (float)2.2));
    }
    if (m.bump_map && m.bump_map <= ntex) {
        const V3 bn = sample_bump_map(pixels, tex[m.bump_map - 1], x, y, 1.0f);
        normal = normalize(transform_normal_to_world(bn, normal, A, B, C, uv));
    }
}

}  // namespace ptmi
