// topo_math.h -- arithmetic of the terrain path, frozen so that results are reproducible bit for bit.
//
// WGSL leaves sin/cos/normalize precision, FMA contraction and sRGB rounding to the implementation; the
// build fixes one definition (DESIGN.md "Arithmetic spec"): IEEE binary32 +,-,*,/,sqrt and fma ONLY where
// written as fmaf() (translation units including this header are compiled with -ffp-contract=off, so the
// compiler never fuses on its own), evaluation order as written, Cody-Waite + Cephes minimax sin/cos,
// exact-curve sRGB tables.
//
// Everything here is `TOPO_HD` (host + device) so the same functions run inside the HIP kernels and,
// for unit tests only, under g++ (tests/host_emul.cpp).  Nothing in this file touches memory it is not
// handed, and nothing here knows about the CPU oracle.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TOPO_HD __host__ __device__ __forceinline__
#else
#define TOPO_HD inline
#endif

#include "srgb_tables.h"

// Tile heights / normals are reached through pointers stored in a descriptor in memory, so the compiler only
// knows them as generic ("flat") pointers: flat loads count on both wait counters and force vmcnt(0)+lgkmcnt(0)
// waits that drain every prefetch.  These casts tell it the truth (hipMalloc memory = global address space).
#if defined(__HIP_DEVICE_COMPILE__)
#define TOPO_GLOBAL_F32(p) ((const __attribute__((address_space(1))) float*)(p))
#define TOPO_GLOBAL_U32(p) ((const __attribute__((address_space(1))) uint32_t*)(p))
#define TOPO_GLOBAL_U32_RW(p) ((__attribute__((address_space(1))) uint32_t*)(p))
#else
#define TOPO_GLOBAL_F32(p) (p)
#define TOPO_GLOBAL_U32(p) (p)
#define TOPO_GLOBAL_U32_RW(p) (p)
#endif

namespace topo {

constexpr float kR0 = 6371000.0f;    // render_shader.wgsl:1, compute_normals_shader.wgsl:1
constexpr float kNear = 50.0f;       // postprocessing_shader.wgsl:19
constexpr float kFar = 500000.0f;    // postprocessing_shader.wgsl:20

struct f3 {
    float x, y, z;
};

TOPO_HD uint32_t f_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
TOPO_HD float bits_f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// WGSL radians()
TOPO_HD float deg2rad(float d) { return d * 0.017453292519943295f; }

// ---- division -------------------------------------------------------------------------------------
// The spec's divisions are IEEE-754 quotients.  On the GPU the compiler expands `/` into a Newton-Raphson core
// wrapped in range handling (2x v_div_scale, v_div_fmas, v_div_fixup: 11 instructions); every division of this
// path has operands and quotient far inside the exponent range (|x|, |y|, |x/y| within 2^-96 .. 2^96, stated at
// the call sites), where the scaling is the identity and the core alone yields the same bits.  The core is
// written out below for the device; the g++ build (tests/host_emul.cpp) uses the plain operator, and the GPU
// parity tests plus tests/test_gpu_parity.py::test_division_probe compare the two.  Not preserved: the sign of a
// zero quotient (-0/y gives +0); no consumer of these quotients can tell the two zeros apart.
struct Recip {   // a divisor and its refined reciprocal, shared by all quotients with that divisor
    float y, r;
};
TOPO_HD Recip recip_of(float y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float a = __builtin_amdgcn_rcpf(y);             // 1 ulp
    return {y, fmaf(fmaf(-y, a, 1.0f), a, a)};            // one Newton-Raphson step
#else
    return {y, 0.0f};
#endif
}
TOPO_HD float div_by(float x, Recip d) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = x * d.r;
    q = fmaf(fmaf(-d.y, q, x), d.r, q);
    return fmaf(fmaf(-d.y, q, x), d.r, q);
#else
    return x / d.y;
#endif
}
TOPO_HD float div_f(float x, float y) { return div_by(x, recip_of(y)); }
// sqrt(x), correctly rounded: the hardware estimate (1 ulp) moved to whichever neighbour the two residuals pick,
// i.e. the compiler's own expansion of sqrtf without its input scaling (x is 0.01 .. 1e15 on this path).
TOPO_HD float sqrt_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TOPO_EXP_OLD_SQRT)
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = fmaf(-dn, s, x), rup = fmaf(-up, s, x);
    s = rdn <= 0.0f ? dn : s;
    return rup > 0.0f ? up : s;
#else
    return sqrtf(x);
#endif
}

// x / C for a constant C whose correctly rounded reciprocal RC is known: one Markstein correction of x * RC.
// Exhaustively checked against the IEEE quotient for C = 255 (every finite x) and C = 0.15f - 0.05f
// (2^-97 <= |x| < 2^123 and 0), see tests/test_emul_cpu.py::test_constant_division.
TOPO_HD float markstein_div(float x, float C, float RC) {
    const float q = x * RC;
    return fmaf(fmaf(-q, C, x), RC, q);
}
TOPO_HD float div_const(float x, float C, float RC) {
#if defined(__HIP_DEVICE_COMPILE__)
    return markstein_div(x, C, RC);
#else
    (void)RC;
    return x / C;
#endif
}

// Exact unsigned division by a run-time constant (n < 2^31, 2 <= d < 2^31): q = (n * m) >> (31 + l) with
// l = ceil(log2 d), m = ceil(2^(31+l) / d) < 2^32 (Granlund & Montgomery 1994, Theorem 4.2 with N = 31).
struct FastDiv {
    uint32_t m, s;   // multiplier, shift - 32
};
inline FastDiv fastdiv_make(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    const uint64_t num = 1ull << (31 + l);
    return {(uint32_t)((num + d - 1) / d), l - 1};
}
TOPO_HD uint32_t fastdiv(uint32_t n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(n, f.m) >> f.s;
#else
    return (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.s;
#endif
}

// sin & cos of x (|x| < ~1e4): k = rint(x*2/pi); r = x - k*pi/2 in three fma steps (Cody-Waite);
// Cephes sinf/cosf kernels on |r| <= pi/4 in fma Horner form; quadrant swap.
TOPO_HD void sincos_f(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    const float z = r * r;
    float a = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    a = fmaf(a, z, -1.6666654611e-1f);
    const float s = fmaf(r * z, a, r);
    float b = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    b = fmaf(b, z, 4.166664568298827e-2f);
    const float c = fmaf(z * z, b, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const bool swap = (q & 1) != 0;
    float so = swap ? c : s;
    float co = swap ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    sn = so;
    cs = co;
}
TOPO_HD float cos_f(float x) {
    float s, c;
    sincos_f(x, s, c);
    return c;
}

TOPO_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
TOPO_HD f3 cross3(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
TOPO_HD f3 normalize3(f3 a) {   // |a| is 0.3 .. 2 on this path (interpolated unit-ish normals, stencil cross products > 1 m^2)
    const Recip len = recip_of(sqrt_f(dot3(a, a)));
    return {div_by(a.x, len), div_by(a.y, len), div_by(a.z, len)};
}
TOPO_HD float fract_f(float x) { return x - floorf(x); }
// clamp(x, 0, 1).  On the device the median-of-three instruction (which the compiler folds into the producing
// instruction's clamp modifier): the same value for every x except NaN, which no stage of this path produces from
// finite heights (and for which the reference's own result is unspecified).
TOPO_HD float sat(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f);
#else
    return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
#endif
}

// rgba8unorm store / load
TOPO_HD uint32_t to_unorm8(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(sat(v) * 255.0f + 0.5f);      // the conversion truncates, which is floor for the non-negative operand
#else
    return (uint32_t)floorf(sat(v) * 255.0f + 0.5f);
#endif
}
// textureLoad of an rgba8unorm channel = c / 255.0f.  Evaluated as one Newton-Markstein step on c * RN(1/255):
// bit-identical to the IEEE quotient for every c in 0..255 (checked exhaustively by tests/test_emul_cpu.py) at
// 3 instructions instead of the ~11 of a correctly rounded f32 division.
TOPO_HD float from_unorm8(uint32_t c) {
    const float x = (float)c, r = 1.0f / 255.0f;
    const float q = x * r;
    return fmaf(fmaf(-q, 255.0f, x), r, q);
}

// column-major 4x4 times (x, y, z, 1): one fused-multiply-add chain per row (what every shader compiler emits
// for OpMatrixTimesVector): t = m0*x; t = fma(m1, y, t); t = fma(m2, z, t); t = t + m3.
TOPO_HD void mat4_point(const float* m, float x, float y, float z, float* o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = m[r] * x;
        t = fmaf(m[4 + r], y, t);
        t = fmaf(m[8 + r], z, t);
        o[r] = t + m[12 + r];
    }
}

TOPO_HD float linear_depth(float d) { return div_f(kFar * kNear, kFar - d * (kFar - kNear)); }   // divisor in [50, 5e5] for d in [0, 1]

// ---- sRGB ----------------------------------------------------------------------------------------
// `thresh` points at 256 floats (TOPO_SRGB_THRESH_BITS reinterpreted; entry 255 = +inf), `decode` at 256.
// encode = number of thresholds <= l.
TOPO_HD uint32_t srgb_encode(const float* thresh, float l) {
    // 8 fixed, independent-of-data probes.  (A "fast estimate + walk" variant measured 1.5x SLOWER inside
    // k_resolve: divergent loops of dependent LDS loads; see profiles/README.md.)
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = 128; step >= 1; step >>= 1) {
        // invariant: thresh[0..lo) <= l.  Probe index lo+step-1 (<= 254).
        if (thresh[lo + step - 1] <= l) lo += step;
    }
    return lo;
}

// The same count through a 12-bit first-level table: `lut` holds, for each bin [i/4096, (i+1)/4096), the number of
// thresholds <= its left edge (TOPO_SRGB_LUT12_WORDS as bytes); at most two thresholds lie inside a bin, so two
// probes finish the count.  `thresh` must be readable up to index 256 with entries 255.. = NaN (never <= anything).  Equal to
// srgb_encode for every f32 input, NaN included (tests/test_emul_cpu.py::test_srgb_lut_encode_equals_probes).
TOPO_HD uint32_t srgb_encode_lut(const float* thresh, const uint8_t* lut, float l) {
    const uint32_t bin = (uint32_t)fminf(fmaxf(l * 4096.0f, 0.0f), 4095.0f);   // exact product; NaN and negatives -> bin 0
    const uint32_t base = lut[bin];
    return base + (thresh[base] <= l ? 1u : 0u) + (thresh[base + 1] <= l ? 1u : 0u);
}

// Three channels at once, written so that the three table lookups of each stage are independent and in flight together
// (one after the other, each encode is two dependent LDS round trips: six per pixel; this way two).
TOPO_HD uint32_t srgb_encode_lut3(const float* thresh, const uint8_t* lut, float r, float g, float b) {
    const uint32_t br = (uint32_t)fminf(fmaxf(r * 4096.0f, 0.0f), 4095.0f), bg = (uint32_t)fminf(fmaxf(g * 4096.0f, 0.0f), 4095.0f),
                   bb = (uint32_t)fminf(fmaxf(b * 4096.0f, 0.0f), 4095.0f);
    const uint32_t ar = lut[br], ag = lut[bg], ab = lut[bb];
    const float r0 = thresh[ar], r1 = thresh[ar + 1], g0 = thresh[ag], g1 = thresh[ag + 1], b0 = thresh[ab], b1 = thresh[ab + 1];
    const uint32_t cr = ar + (r0 <= r ? 1u : 0u) + (r1 <= r ? 1u : 0u), cg = ag + (g0 <= g ? 1u : 0u) + (g1 <= g ? 1u : 0u),
                   cb = ab + (b0 <= b ? 1u : 0u) + (b1 <= b ? 1u : 0u);
    return cr | (cg << 8) | (cb << 16);
}

// ---- normal stencil (compute_normals*.wgsl) ------------------------------------------------------
// Returns the packed rgba8unorm texel (alpha 0) for the four neighbour heights; `x`,`y` are the metric
// half-steps the shaders build from the pixel scale and the row latitude.
TOPO_HD uint32_t normal_texel(float x, float y, float hT, float hL, float hR, float hB) {
    const f3 top = {0.0f, y, hT}, left = {-x, 0.0f, hL}, right = {x, 0.0f, hR}, bottom = {0.0f, -y, hB};
    const f3 dxv = {right.x - left.x, right.y - left.y, right.z - left.z};
    const f3 dyv = {top.x - bottom.x, top.y - bottom.y, top.z - bottom.z};
#if defined(__HIP_DEVICE_COMPILE__)
    // dxv.y and dyv.x are exact zeros, so with finite heights cross3's products by them are zeros too and each component
    // is a single product (up to the sign of a zero component, which the + 1 below erases): same texel, 5 operations for 9
    f3 n = normalize3(f3{-(dxv.z * dyv.y), -(dxv.x * dyv.z), dxv.x * dyv.y});
#else
    f3 n = normalize3(cross3(dxv, dyv));
#endif
    n = {0.5f * (n.x + 1.0f), 0.5f * (n.y + 1.0f), 0.5f * (n.z + 1.0f)};
    return to_unorm8(n.x) | (to_unorm8(n.y) << 8) | (to_unorm8(n.z) << 16) | (to_unorm8(0.0f) << 24);
}

// The same texel by a shorter route, when that route can be trusted.  The texel is three 8-bit codes floor(t_i) with
// t_i = 255 * 0.5 (n_i + 1) + 0.5 = 127.5 n_i + 128, n = v / |v|.  normal_texel() reaches t_i through a correctly rounded
// sqrt, three IEEE divisions and four more roundings: |t_spec - T_i| <= 4.6e-5 for the real-number value T_i (n: 3 ulp of
// relative error = 1.8e-7, n + 1: 6e-8, x 127.5 = 3.1e-5; the two roundings near 255: 7.6e-6 each).  The short route --
// one reciprocal square root estimate (v_rsq_f32, 1 ulp), three products, one fma each -- has |t' - T_i| <= 3.8e-5
// (n': 2.4e-7 relative, x 127.5; fma: 7.6e-6).  So whenever the fractional part of t' is at least kNormalGuard = 2e-4
// away from 0 and from 1, floor(t') = floor(t_spec) -- the two differ by at most 8.4e-5 -- and the texel is known
// without the divisions; otherwise (1.2 texels in a thousand, and any non-finite input) the caller evaluates
// normal_texel().  Exhaustive-by-sampling check of exactly this claim: tests/test_emul_cpu.py::test_normal_texel_fast_path.
constexpr float kNormalGuard = 2.0e-4f;
TOPO_HD bool normal_texel_fast(float x, float y, float hT, float hL, float hR, float hB, uint32_t& texel) {
    const float dxx = x - (-x), dyy = y - (-y), dxz = hR - hL, dyz = hT - hB;      // the same differences normal_texel forms
    const float vx = -(dxz * dyy), vy = -(dxx * dyz), vz = dxx * dyy;
    const float ss = (vx * vx + vy * vy) + vz * vz;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rsqf(ss);
#else
    const float r = 1.0f / sqrtf(ss);          // (test builds: also within 1 ulp of the reciprocal square root)
#endif
    const float tx = fmaf(vx * r, 127.5f, 128.0f), ty = fmaf(vy * r, 127.5f, 128.0f), tz = fmaf(vz * r, 127.5f, 128.0f);
    // |g - 1/2| <= 1/2 - guard  <=>  guard <= g <= 1 - guard up to the rounding of g - 1/2 (6e-8: the guard band has 1.2e-4 to
    // spare), g = the fractional part of t.  ss >= 1e-30 keeps the squares' underflow out of the error bound (a term that
    // underflows is < 1e-8 of such an ss); with it |v_i r| <= 1 + 1e-6, so t lies in [0.49, 255.51] and needs no range test:
    // there the fractional part is exact, the conversion to integer truncates = floors, and an overflowed ss gives r = 0,
    // t = 128 (or NaN where v_i is infinite), g_y = g_z = 0: refused; a NaN height makes ss NaN, which fails the first test.
    // (The kernel is as much bound by instruction issue as by HBM: on the device the fractional part is one instruction
    // (v_fract_f32: exact for t >= 0), the three band tests are one three-way maximum and one comparison, the codes come
    // straight from t.)
    constexpr float kLim = 0.5f - kNormalGuard;
#if defined(__HIP_DEVICE_COMPILE__)
    const float gx = __builtin_amdgcn_fractf(tx), gy = __builtin_amdgcn_fractf(ty), gz = __builtin_amdgcn_fractf(tz);
    const float worst = fmaxf(fmaxf(fabsf(gx - 0.5f), fabsf(gy - 0.5f)), fabsf(gz - 0.5f));      // (a NaN operand is ignored: see above why none can decide)
    const bool ok = ss >= 1.0e-30f && worst <= kLim;
#else
    const float gx = tx - floorf(tx), gy = ty - floorf(ty), gz = tz - floorf(tz);      // exact: t in [0, 256)
    const bool ok = ss >= 1.0e-30f && fabsf(gx - 0.5f) <= kLim && fabsf(gy - 0.5f) <= kLim && fabsf(gz - 0.5f) <= kLim;
#endif
    texel = (uint32_t)tx | ((uint32_t)ty << 8) | ((uint32_t)tz << 16);      // (truncation = floor for t >= 0; alpha: to_unorm8(0) = 0; not used unless ok)
    return ok;
}

// ---- fragment shading (render_shader.wgsl:75-115) ------------------------------------------------
// fract(x) = x - floor(x).  The device's v_fract_f32 returns that very value for every f32 x -- NaN and the infinities
// included -- EXCEPT the negatives above -2^-24, where the difference rounds to 1.0 and the instruction returns the largest
// float below 1 (tools/exp_fract_probe.py: all 2^32 inputs; tests/test_gpu_parity.py::test_fract_probe).  kInstr selects the
// instruction (one issue slot for two); the caller has to know that x is not such a negative.
template <bool kInstr>
TOPO_HD float fract_t(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (kInstr) return __builtin_amdgcn_fractf(x);
#endif
    return x - floorf(x);
}
// kFirstInstr: the two fractions of the (scaled) inputs may use the instruction.  The third one always may: its operand is a
// product of numbers that are >= 0 (px, py in [0, 1], d >= 0) or NaN.
template <bool kFirstInstr>
TOPO_HD float hash12n_t(float sx, float sy) {
    float px = fract_t<kFirstInstr>(sx * 5.3987f);
    float py = fract_t<kFirstInstr>(sy * 5.4421f);
    const float d = py * (px + 21.5351f) + px * (py + 14.3137f);
    px += d;
    py += d;
    return fract_t<true>(px * py * 95.4307f);
}
TOPO_HD float hash12n(float sx, float sy) { return hash12n_t<false>(sx, sy); }

// ditherRGB's six hashes at p = (px, py): out[k] = lin + (hash42n(p)_k + hash42n(p + 0.13)_k - 1) / 255
template <bool kFirstInstr>
TOPO_HD void dither_rgb(float px, float py, float lin, float out[4]) {
    const float qx = px + 0.13f, qy = py + 0.13f;
    const float off[3] = {0.0f, 0.07f, 0.11f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float h1 = k == 0 ? hash12n_t<kFirstInstr>(px, py) : hash12n_t<kFirstInstr>(px + off[k], py + off[k]);
        const float h2 = k == 0 ? hash12n_t<kFirstInstr>(qx, qy) : hash12n_t<kFirstInstr>(qx + off[k], qy + off[k]);
        out[k] = lin + div_const(1.0f * (h1 + h2 - 1.0f), 255.0f, 1.0f / 255.0f);
    }
}

// Linear colour of fs_main for one fragment.  frag = pixel centre, cam = camera_pos.xy.
TOPO_HD void shade_fragment(int view_mode, f3 sun, float cam_x, float cam_y, float frag_x, float frag_y, f3 wpos,
                            f3 wnrm, float out[4]) {
    if (view_mode == 2) {
        out[0] = 0.5f * (wnrm.x + 1.0f);
        out[1] = 0.5f * (wnrm.y + 1.0f);
        out[2] = 0.5f * (wnrm.z + 1.0f);
        out[3] = 1.0f;
        return;
    }
    const float d = dot3(normalize3(wnrm), sun);
    const float lin = 0.01f + 0.7f * (d > 0.0f ? d : 0.0f);
    out[3] = 1.0f;
    if (view_mode == 1) {
        out[0] = out[1] = out[2] = lin;
        return;
    }
    const float px = frag_x + cam_x - wpos.x, py = frag_y + cam_y - wpos.y;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TOPO_EXP_SPEC_FRACT)      // (experiment build TOPO_EXP_SPEC_FRACT: the two-instruction form everywhere)
    // The twelve first-level fractions take (p + c) * m with c in {0, 0.07, 0.11, 0.13, 0.13 + 0.07, 0.13 + 0.11} and m > 5: such
    // an operand is a negative above -2^-24 only if p + c lies in (-1.2e-8, 0), i.e. p within 2e-8 of -c, and every -c lies in
    // [-0.25, 0].  So a pixel with neither coordinate in [-0.3, 0] may use the instruction throughout (18 issue slots for 36);
    // the wave takes the two-instruction form when any of its pixels is in the window (p is a distance of hundreds of metres
    // to hundreds of kilometres: about one row of 64 pixels in 200).  A NaN coordinate fails the test -- and is NaN in both forms.
    const bool window = fminf(fabsf(px + 0.15f), fabsf(py + 0.15f)) <= 0.15f;
    if (__builtin_amdgcn_ballot_w64(window) == 0ull) {
        dither_rgb<true>(px, py, lin, out);
        return;
    }
#endif
    dither_rgb<false>(px, py, lin, out);
}

// ---- post pass (postprocessing_shader.wgsl:68-96) -------------------------------------------------
// `c8` = the render-target texel (sRGB8 rgb + unorm8 alpha); taps in the shader's loop order (i outer = x
// offset -1..1, j inner = y offset -1..1, centre skipped).
// kLut selects srgb_encode_lut (with `lut`) over srgb_encode: same result, fewer probes.
template <bool kLut>
TOPO_HD uint32_t post_pixel_t(const float* thresh, const float* decode, uint32_t c8, float center, const float ln[8], const uint8_t* lut,
                              bool srgb_target = true) {
    // center / ln[] are ALREADY linear_depth() of the depth taps (each is a pure function of its texel, so a
    // kernel may compute it once per texel and share it between the up to nine pixels that tap it).
    // srgb_target: the targets are *Srgb formats (decode on sample, encode on store); otherwise plain unorm8 both ways.
    float contour = 8.0f * center;
#pragma unroll
    for (int k = 0; k < 8; ++k) contour -= ln[k];
    float t = sat(div_const(div_f(contour, center) - 0.05f, 0.15f - 0.05f, 1.0f / (0.15f - 0.05f)));   // center in [50, 5e5]
    const float a = t * t * (3.0f - 2.0f * t);
    // mix(x, 0, 0) = x*1 + 0*0 = x and the decode->encode round trip of a code is the identity (checked when the
    // tables are generated; to_unorm8(from_unorm8(c)) = c for the plain formats), alpha stays 255; mix(x, 0, 1) = x*0 + 0*1 = 0:
    // both ends skip the table work.
    if (a == 0.0f && (c8 >> 24) == 255u) return c8;
    if (a == 1.0f && (c8 >> 24) == 255u) return 0xFF000000u;
    const float dr = srgb_target ? decode[c8 & 255u] : from_unorm8(c8 & 255u), dg = srgb_target ? decode[(c8 >> 8) & 255u] : from_unorm8((c8 >> 8) & 255u),
                db = srgb_target ? decode[(c8 >> 16) & 255u] : from_unorm8((c8 >> 16) & 255u);
    const float r = dr * (1.0f - a) + 0.0f * a;
    const float g = dg * (1.0f - a) + 0.0f * a;
    const float b = db * (1.0f - a) + 0.0f * a;
    const float al = from_unorm8(c8 >> 24) * (1.0f - a) + 1.0f * a;
    if (!srgb_target) return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | (to_unorm8(al) << 24);
    if (kLut) return srgb_encode_lut3(thresh, lut, r, g, b) | (to_unorm8(al) << 24);
    return srgb_encode(thresh, r) | (srgb_encode(thresh, g) << 8) | (srgb_encode(thresh, b) << 16) |
           (to_unorm8(al) << 24);
}
// The same mix for a render-target sample that is not a texel (the pixelise branch: a bilinear sample in linear light):
// rc = the sampled colour (decoded), alpha included.  Returns the surface texel, R G B A from the low byte.
TOPO_HD uint32_t post_mix(const float* thresh, const float rc[4], float center, const float ln[8], bool srgb_target) {
    float contour = 8.0f * center;
#pragma unroll
    for (int k = 0; k < 8; ++k) contour -= ln[k];
    const float t = sat(div_const(div_f(contour, center) - 0.05f, 0.15f - 0.05f, 1.0f / (0.15f - 0.05f)));
    const float a = t * t * (3.0f - 2.0f * t);
    const float r = rc[0] * (1.0f - a) + 0.0f * a, g = rc[1] * (1.0f - a) + 0.0f * a, b = rc[2] * (1.0f - a) + 0.0f * a;
    const float al = rc[3] * (1.0f - a) + 1.0f * a;
    if (!srgb_target) return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | (to_unorm8(al) << 24);
    return srgb_encode(thresh, r) | (srgb_encode(thresh, g) << 8) | (srgb_encode(thresh, b) << 16) | (to_unorm8(al) << 24);
}

// The pixelise branch of the post shader (postprocessing_shader.wgsl:70-74): uv = floor(uv n) / n, then textureSample of the
// render target through s_render (mag Linear, min Nearest, clamp-to-edge, one level: texture.rs:69-78).  What WebGPU leaves to
// the implementation is fixed in DESIGN.md ("Raster spec", item 10): derivatives across the pixel's 2 x 2 quad (quads start at
// even coordinates), rho = max(|du/dx| tw, |dv/dy| th); rho > 1: the nearest texel (floor(u tw), floor(v th)); otherwise the
// four texels around (u tw - 1/2, v th - 1/2), clamped to the edge, weighted with the f32 fractions by WGSL's mix.
// texel(x, y, out): the decoded render-target texel at CLAMPED integer coordinates.
TOPO_HD float pixelized_uv(float frag, float viewport, float n) { return div_f(floorf(div_f(frag, viewport) * n), n); }      // (n >= 1, viewport >= 1)
template <typename Texel>
TOPO_HD void sample_pixelized(int32_t px, int32_t py, float vw, float vh, float n, int32_t tw, int32_t th, Texel&& texel, float out[4]) {
    const float u = pixelized_uv((float)px + 0.5f, vw, n), v = pixelized_uv((float)py + 0.5f, vh, n);
    const float ux = pixelized_uv((float)(px ^ 1) + 0.5f, vw, n), vy = pixelized_uv((float)(py ^ 1) + 0.5f, vh, n);
    const float rx = fabsf(ux - u) * (float)tw, ry = fabsf(vy - v) * (float)th;
    auto clampi = [](int32_t a, int32_t hi) { return a < 0 ? 0 : (a > hi ? hi : a); };
    if ((rx > ry ? rx : ry) > 1.0f) {
        texel(clampi((int32_t)floorf(u * (float)tw), tw - 1), clampi((int32_t)floorf(v * (float)th), th - 1), out);
        return;
    }
    const float tx = u * (float)tw - 0.5f, ty = v * (float)th - 0.5f;
    const float x0 = floorf(tx), y0 = floorf(ty), fx = tx - x0, fy = ty - y0;
    const int32_t ix = (int32_t)x0, iy = (int32_t)y0;
    float c00[4], c10[4], c01[4], c11[4];
    texel(clampi(ix, tw - 1), clampi(iy, th - 1), c00);
    texel(clampi(ix + 1, tw - 1), clampi(iy, th - 1), c10);
    texel(clampi(ix, tw - 1), clampi(iy + 1, th - 1), c01);
    texel(clampi(ix + 1, tw - 1), clampi(iy + 1, th - 1), c11);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float top = c00[k] * (1.0f - fx) + c10[k] * fx, bot = c01[k] * (1.0f - fx) + c11[k] * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}

TOPO_HD uint32_t post_pixel(const float* thresh, const float* decode, uint32_t c8, float center, const float ln[8], bool srgb_target = true) {
    return post_pixel_t<false>(thresh, decode, c8, center, ln, nullptr, srgb_target);
}

}  // namespace topo
