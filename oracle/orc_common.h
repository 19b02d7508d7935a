/*
 * oracle/orc_common.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar f32, no FMA contraction) of the Go reference's
 * float32 vector type and helpers.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load anything under oracle/.  The product
 * (pathtrace_amd/, libptmi.so) never includes, links or calls this code.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and
 * no Go toolchain exists in the build image, so this restatement is pinned only
 * by source-order review against the cited reference lines.
 *
 * Build flags that the restatement depends on (oracle/Makefile):
 *   -O2 -ffp-contract=off -fno-fast-math -mfpmath=sse -msse2
 * i.e. every + - * / and sqrt is a single correctly rounded IEEE-754 binary32
 * operation, exactly what Go/amd64 (GOAMD64=v1, no FMA fusion) emits.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* linear_bvh.go:9   const INF = math.MaxFloat32 (NOT +Inf) */
#define ORC_INF 3.40282346638528859811704183484516925440e+38f

typedef struct { float x, y, z; } v3;

/* Go builtin min/max on floats (spec "Min and max"): NaN if any argument is NaN,
 * and -0 < +0.  Used by linear_bvh.go:44-45,58-60,69 and bvh.go:55-60,265-270. */
static inline float gominf(float a, float b) {
    if (a != a || b != b) return NAN;
    if (a == 0.0f && b == 0.0f) return (signbit(a) || signbit(b)) ? -0.0f : 0.0f;
    return a < b ? a : b;
}
static inline float gomaxf(float a, float b) {
    if (a != a || b != b) return NAN;
    if (a == 0.0f && b == 0.0f) return (signbit(a) && signbit(b)) ? -0.0f : 0.0f;
    return a > b ? a : b;
}

/* vec3.go:22-50 */
static inline v3 v3_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 v3_scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
/* vec3.go:52-54   x*x + y*y + z*z, left-associated */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* vec3.go:56-58   both math32 packages give a correctly rounded f32 sqrt */
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
/* vec3.go:60-66   zero-length -> zero vector; else three muls by (1/len) */
static inline v3 v3_normalize(v3 a) {
    float len = v3_length(a);
    if (len == 0.0f) { v3 z = { 0, 0, 0 }; return z; }
    return v3_scale(a, 1.0f / len);
}
/* vec3.go:79-85 */
static inline v3 v3_cross(v3 a, v3 b) {
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
/* vec3.go:108-116 */
static inline v3 v3_cmul(v3 a, v3 b) { v3 r = { a.x * b.x, a.y * b.y, a.z * b.z }; return r; }
/* vec3.go:118-124 */
static inline v3 v3_inverse(v3 a) { v3 r = { 1.0f / a.x, 1.0f / a.y, 1.0f / a.z }; return r; }
/* vec3.go:134-136 */
static inline v3 v3_lerp(v3 a, v3 b, float t) { return v3_add(a, v3_scale(v3_sub(b, a), t)); }

#endif
