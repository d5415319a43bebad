/*
 * oracle/oracle_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).  PARITY UNPINNED (see topo_oracle.cpp).
 *
 * Scalar f32 arithmetic the CPU oracle uses to restate the reference's WGSL.  WGSL leaves the
 * precision of sin/cos/normalize and FMA contraction to the implementation, so the build freezes
 * one definition (DESIGN.md "Arithmetic spec"): IEEE-754 binary32 add/sub/mul/div/sqrt, fma only
 * where fmaf() is written (compile with -ffp-contract=off), evaluation order as written here,
 * and the sin/cos below.  This header is written independently of the product's device header
 * (topo-renderer_amd/csrc/topo_math.h); tests compare the two bit-for-bit.
 */
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace omath {

static const float R0 = 6371000.0f;            /* render_shader.wgsl:1, geometry.rs:5 */
static const float NEAR_Z = 50.0f;             /* postprocessing_shader.wgsl:19, camera.rs:6 */
static const float FAR_Z = 500000.0f;          /* postprocessing_shader.wgsl:20, camera.rs:7 */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* WGSL radians(e) = e * pi / 180, evaluated as one multiply by the f32 constant. */
static inline float radians(float deg) { return deg * 0.017453292519943295f; }

/*
 * sin/cos spec: k = rint(x * 2/pi); r = fma(-k,P3, fma(-k,P2, fma(-k,P1, x))) (Cody-Waite); z = r*r;
 *   sin(r) = fma(r*z, fma(fma(S3,z,S2), z, S1), r),  cos(r) = fma(z*z, fma(fma(C3,z,C2), z, C1), fma(-0.5,z,1))
 * (Cephes single-precision minimax coefficients, |r| <= pi/4, error < 1 ulp), then quadrant fix-up.
 * fmaf() is the correctly rounded fused multiply-add on every platform (hardware or libm).
 */
static inline void sincos_spec(float x, float* s_out, float* c_out) {
    const float TWO_OVER_PI = 0.63661977236758134f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188e-8f;
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, P1, x);
    r = fmaf(-k, P2, r);
    r = fmaf(-k, P3, r);
    float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s = fmaf(r * z, ps, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    float so, co;
    switch (q) {
        case 0: so = s; co = c; break;
        case 1: so = c; co = -s; break;
        case 2: so = -s; co = -c; break;
        default: so = -c; co = s; break;
    }
    *s_out = so;
    *c_out = co;
}
static inline float sin_spec(float x) { float s, c; sincos_spec(x, &s, &c); return s; }
static inline float cos_spec(float x) { float s, c; sincos_spec(x, &s, &c); return c; }

struct v3 { float x, y, z; };

static inline v3 add(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3 sub(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3 scale(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
/* WGSL dot: (a.x*b.x + a.y*b.y) + a.z*b.z */
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
/* WGSL normalize(e) = e / length(e) */
static inline v3 normalize(v3 a) {
    float len = sqrtf(dot(a, a));
    return {a.x / len, a.y / len, a.z / len};
}
static inline float fract(float x) { return x - floorf(x); }
static inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
/* WGSL smoothstep(lo, hi, x): t = clamp((x-lo)/(hi-lo), 0, 1); t*t*(3 - 2t) */
static inline float smoothstep(float lo, float hi, float x) {
    float t = clamp01((x - lo) / (hi - lo));
    return t * t * (3.0f - 2.0f * t);
}
/* WGSL mix(x, y, a) = x*(1-a) + y*a */
static inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* float -> unorm8 (textureStore to rgba8unorm): round-half-up of clamp(v,0,1)*255 */
static inline uint8_t unorm8(float v) { return (uint8_t)floorf(clamp01(v) * 255.0f + 0.5f); }
/* unorm8 -> float (textureLoad from rgba8unorm) */
static inline float from_unorm8(uint8_t c) { return (float)c / 255.0f; }

/* GPU shader mat4 (column-major) * vec4(p, 1): per component t = c0*x; t = fma(c1,y,t); t = fma(c2,z,t); t + c3 */
static inline void mat4_mul_point(const float* m, float x, float y, float z, float* out) {
    for (int r = 0; r < 4; ++r) {
        float acc = m[0 + r] * x;
        acc = fmaf(m[4 + r], y, acc);
        acc = fmaf(m[8 + r], z, acc);
        out[r] = acc + m[12 + r];
    }
}
/* host (glam) mat4 * vec4: ((c0*x + c1*y) + c2*z) + c3*w, per component, no fma (default x86-64 build) */
static inline void mat4_mul_vec4(const float* m, float x, float y, float z, float w, float* out) {
    for (int r = 0; r < 4; ++r) {
        float acc = m[0 + r] * x;
        acc = acc + m[4 + r] * y;
        acc = acc + m[8 + r] * z;
        acc = acc + m[12 + r] * w;
        out[r] = acc;
    }
}
/* upper-left 3x3 of a column-major mat4 times (x,y,z,0); the w=0 column contributes nothing */
static inline v3 mat4_mul_dir(const float* m, v3 n) {
    v3 o;   /* same fma chain as mat4_mul_point, w = 0 */
    o.x = fmaf(m[8], n.z, fmaf(m[4], n.y, m[0] * n.x));
    o.y = fmaf(m[9], n.z, fmaf(m[5], n.y, m[1] * n.x));
    o.z = fmaf(m[10], n.z, fmaf(m[6], n.y, m[2] * n.x));
    return o;
}

/* dist_from_depth: postprocessing_shader.wgsl:52-54, camera.rs:12-14 */
static inline float dist_from_depth(float d) { return FAR_Z * NEAR_Z / (FAR_Z - d * (FAR_Z - NEAR_Z)); }

}  // namespace omath
