// topo_pipeline.h -- per-vertex / per-triangle / per-pixel stages of the terrain path.
//
// The reference draws every tile's mesh through the fixed-function rasteriser (pipeline.rs:221-246);
// here the same semantics are spelled out as plain functions (DESIGN.md "Raster spec") that the HIP
// kernels in topo_kernels.hip call.  All functions are pure (TOPO_HD) so tests can also execute them
// under g++ (tests/host_emul.cpp); the product only ever runs them on the GPU.
#pragma once

#include "topo_math.h"

namespace topo {

// ---- device-side descriptors ---------------------------------------------------------------------
struct TileDev {                 // one loaded 1x1 degree tile (RenderBuffer, render_buffer.rs:23-31)
    const float* heights;        // w*h f32, row-major, row 0 = north      (R32Float texture)
    uint32_t* normals;           // w*h packed rgba8unorm, zero-initialised (Rgba8Unorm texture)
    const float* block_minmax;   // 2 floats per raster block: min, max height of its vertices
    const float* trig_lon;       // w (sin, cos) pairs: sincos_f(vertex_lon(t, x)) of every vertex column ...
    const float* trig_lat;       // h pairs: sincos_f(vertex_lat(t, y)) of every vertex row (load phase, k_block_minmax)
    const double* block_bounds;  // for the cull, view-independent (k_block_minmax): 4 doubles per raster block (bounding-sphere
                                 // centre and radius), then 12 per block (unit directions of its four corners), then 1 per
                                 // block (sagitta of the patch over the flat hull of its corners, metres)
    float raster_x, raster_y;    // TerrainUniforms (render/data.rs:113-121)
    float model_x, model_y;
    float scale_x, scale_y;
    float rot[9];                // upper-left 3x3 of normal_to_world_rot, column-major
    float pad_;
};

struct ViewDev {                 // one frame's Uniforms (render/data.rs:33-41), the fields the shaders read
    float proj[16];              // camera_proj, column-major
    float cam_x, cam_y;          // camera_pos.xy (dither seed only)
    float sun[3];
    int32_t view_mode;
};

constexpr uint32_t kNoTri = 0xFFFFFFFFu;
constexpr uint64_t kVisClear = 0x3F800000FFFFFFFFull;   // depth 1.0 | no triangle

// vertex flags
constexpr int kVtxOk = 0, kVtxNear = 1, kVtxGuard = 2;

struct alignas(16) SVert {       // post-transform vertex as the rasteriser sees it
    int32_t X, Y;                // 1/256 pixel, round-half-even
    float z;                     // z_ndc
    int32_t flag;                // kVtxOk / kVtxNear (z_clip < 0) / kVtxGuard (outside +-2^20 px or non-finite)
};

struct VFull {                   // full vs_main output
    float clip[4];
    f3 wpos;
    f3 wnrm;
};

// ---- vertex stage: render_shader.wgsl:35-73 --------------------------------------------------------
// to_model + radians: longitude depends only on the vertex column, latitude only on its row, so a raster
// block evaluates each sin/cos once per row/column (same function of the same input -> same bits).
TOPO_HD float vertex_lon(const TileDev& t, uint32_t vx) { return deg2rad(((float)vx - t.raster_x) * t.scale_x + t.model_x); }
TOPO_HD float vertex_lat(const TileDev& t, uint32_t vy) { return deg2rad(((float)vy - t.raster_y) * -t.scale_y + t.model_y); }
TOPO_HD f3 world_from_sincos(float height, float sla, float cla, float slo, float clo) {
    const float R = kR0 + height;
    return {R * cla * clo, R * cla * slo, R * sla};
}
TOPO_HD f3 vertex_world(const TileDev& t, uint32_t vx, uint32_t vy, float height) {
    float sla, cla, slo, clo;
    sincos_f(vertex_lat(t, vy), sla, cla);
    sincos_f(vertex_lon(t, vx), slo, clo);
    return world_from_sincos(height, sla, cla, slo, clo);
}

// 2 * (c / 255) - 1: the shader's decode of one normal channel.  `ndec`, when given, is that function tabulated for
// c = 0..255 (k_resolve keeps it in LDS: one lookup instead of six instructions per channel).
TOPO_HD float normal_channel(uint32_t c) { return 2.0f * from_unorm8(c) - 1.0f; }
// kTableOnly: the caller vouches for `ndec` (k_resolve: its LDS table) -- otherwise the compiler, which cannot prove a pointer into
// LDS non-null, puts a test and a branch in front of every one of the nine lookups of a record pass, each with its own wait.
template <bool kTableOnly = false>
TOPO_HD f3 vertex_normal(const TileDev& t, uint32_t packed, const float* ndec = nullptr) {
    const bool table = kTableOnly || ndec != nullptr;
    const float nx = table ? ndec[packed & 255u] : normal_channel(packed & 255u);
    const float ny = table ? ndec[(packed >> 8) & 255u] : normal_channel((packed >> 8) & 255u);
    const float nz = table ? ndec[(packed >> 16) & 255u] : normal_channel((packed >> 16) & 255u);
    const float* m = t.rot;   // mat * vec4(n, 0): fma chain, the zero w column drops out
    return {fmaf(m[6], nz, fmaf(m[3], ny, m[0] * nx)), fmaf(m[7], nz, fmaf(m[4], ny, m[1] * nx)),
            fmaf(m[8], nz, fmaf(m[5], ny, m[2] * nx))};
}

// clip -> framebuffer (WebGPU framebufferCoords), snapped.  Returns the flag.
TOPO_HD int clip_to_screen(const float clip[4], float W, float H, SVert& s) {
    s.X = 0;
    s.Y = 0;
    s.z = 0.0f;
    if (!(clip[2] >= 0.0f)) {
        s.flag = kVtxNear;
        return kVtxNear;
    }
    const float iw = div_f(1.0f, clip[3]);           // perspective divide = one reciprocal + multiplies (w >= near here)
    const float nx = clip[0] * iw, ny = clip[1] * iw, nz = clip[2] * iw;
    const float hw = 0.5f * W, hh = 0.5f * H;       // exact: W, H are small integers
    const float xf = fmaf(nx, hw, hw);              // 0.5*(ndc.x + 1)*W
    const float yf = fmaf(-ny, hh, hh);             // 0.5*(1 - ndc.y)*H
    if (!(fabsf(xf) <= 1048576.0f) || !(fabsf(yf) <= 1048576.0f)) {
        s.flag = kVtxGuard;
        return kVtxGuard;
    }
    s.X = (int32_t)rintf(xf * 256.0f);
    s.Y = (int32_t)rintf(yf * 256.0f);
    s.z = nz;
    s.flag = kVtxOk;
    return kVtxOk;
}

// ---- mesh topology: render_buffer.rs:185-219 -------------------------------------------------------
// Triangle `tri` of a tile (index-buffer order): cell = tri/2 with i = cell / (h-1) (x, outer loop) and
// j = cell % (h-1) (y, inner loop); k = tri & 1.  Vertex (i, j) samples texel (x=i, y=j).
TOPO_HD void triangle_vertices(uint32_t tri, uint32_t hm1, uint32_t vx[3], uint32_t vy[3]) {
    const uint32_t cell = tri >> 1, k = tri & 1u;
    const uint32_t i = cell / hm1, j = cell - i * hm1;
    const bool even = ((i + j) & 1u) == 0;
    // a=(i,j) b=(i,j+1) c=(i+1,j) d=(i+1,j+1)
    if (k == 0) {                       // [a, b, d] | [a, b, c]
        vx[0] = i; vy[0] = j;
        vx[1] = i; vy[1] = j + 1;
        vx[2] = i + 1; vy[2] = even ? j + 1 : j;
    } else if (even) {                  // [d, c, a]
        vx[0] = i + 1; vy[0] = j + 1;
        vx[1] = i + 1; vy[1] = j;
        vx[2] = i; vy[2] = j;
    } else {                            // [d, c, b]
        vx[0] = i + 1; vy[0] = j + 1;
        vx[1] = i + 1; vy[1] = j;
        vx[2] = i; vy[2] = j + 1;
    }
}

// ---- triangle setup + coverage ---------------------------------------------------------------------
// Front face = counter-clockwise on screen = negative doubled area in y-down framebuffer space.
// Edge a->b: F(p) = dy*(px-ax) - dx*(py-ay) >= 0 inside; it owns its boundary iff it is a left edge
// (dy > 0) or a top edge (dy == 0 && dx < 0).  Pixel centres sit at (256*px+128, 256*py+128).
struct TriSetup {
    int64_t ax[3], ay[3], dx[3], dy[3];   // edges e0 = v1->v2 (weight of v0), e1 = v2->v0, e2 = v0->v1
    int64_t bias[3];
    float iA;                             // 1 / doubled area (f32 reciprocal of the exact integer)
    float z0, dz1, dz2;
    int32_t px0, px1, py0, py1;           // inclusive pixel bbox clipped to the target
};

TOPO_HD int64_t floor_div256(int64_t a) { return a >> 8; }   // arithmetic shift == floor division by 256

TOPO_HD bool triangle_setup(const SVert& s0, const SVert& s1, const SVert& s2, int32_t W, int32_t H, TriSetup& ts) {
    const int64_t X0 = s0.X, Y0 = s0.Y, X1 = s1.X, Y1 = s1.Y, X2 = s2.X, Y2 = s2.Y;
    const int64_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    if (area2 >= 0) return false;
    int64_t mnx = X0 < X1 ? X0 : X1; mnx = mnx < X2 ? mnx : X2;
    int64_t mxx = X0 > X1 ? X0 : X1; mxx = mxx > X2 ? mxx : X2;
    int64_t mny = Y0 < Y1 ? Y0 : Y1; mny = mny < Y2 ? mny : Y2;
    int64_t mxy = Y0 > Y1 ? Y0 : Y1; mxy = mxy > Y2 ? mxy : Y2;
    int64_t px0 = floor_div256(mnx - 128 + 255), px1 = floor_div256(mxx - 128);
    int64_t py0 = floor_div256(mny - 128 + 255), py1 = floor_div256(mxy - 128);
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > W - 1) px1 = W - 1;
    if (py1 > H - 1) py1 = H - 1;
    if (px0 > px1 || py0 > py1) return false;
    ts.px0 = (int32_t)px0; ts.px1 = (int32_t)px1; ts.py0 = (int32_t)py0; ts.py1 = (int32_t)py1;
    const int64_t ex[3] = {X1, X2, X0}, ey[3] = {Y1, Y2, Y0};
    const int64_t fx[3] = {X2, X0, X1}, fy[3] = {Y2, Y0, Y1};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        ts.ax[e] = ex[e]; ts.ay[e] = ey[e];
        ts.dx[e] = fx[e] - ex[e]; ts.dy[e] = fy[e] - ey[e];
        const bool own = (ts.dy[e] > 0) || (ts.dy[e] == 0 && ts.dx[e] < 0);
        ts.bias[e] = own ? 0 : -1;
    }
    ts.iA = div_f(1.0f, (float)(-area2));
    ts.z0 = s0.z; ts.dz1 = s1.z - s0.z; ts.dz2 = s2.z - s0.z;
    return true;
}

// Coverage + depth of pixel (px,py).  Returns false when not covered or clipped by the far plane.
TOPO_HD bool triangle_pixel(const TriSetup& ts, int32_t px, int32_t py, float& z, float b[3]) {
    const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
    int64_t F[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        F[e] = ts.dy[e] * (cx - ts.ax[e]) - ts.dx[e] * (cy - ts.ay[e]);
        if (F[e] + ts.bias[e] < 0) return false;
    }
    b[0] = (float)F[0] * ts.iA;
    b[1] = (float)F[1] * ts.iA;
    b[2] = (float)F[2] * ts.iA;
    float zz = fmaf(b[1], ts.dz1, fmaf(b[2], ts.dz2, ts.z0));
    if (!(zz < 1.0f)) return false;
    if (zz < 0.0f) zz = 0.0f;
    z = zz;
    return true;
}

// Barycentrics of pixel (px,py) for a triangle known to own it (the resolve pass): the same integers and the same
// reciprocal triangle_setup/triangle_pixel produce, without the bounding box, the ownership biases or the
// coverage test.  Small triangles (all deltas < 2^14) use 32-bit products (24-bit multiplies on the device).
#if defined(__HIP_DEVICE_COMPILE__)
#define TOPO_MUL24(a, b) __mul24((a), (b))
#else
#define TOPO_MUL24(a, b) ((a) * (b))
#endif
TOPO_HD bool triangle_bary(const SVert& s0, const SVert& s1, const SVert& s2, int32_t px, int32_t py, float b[3]) {
    const int32_t X0 = s0.X, Y0 = s0.Y, X1 = s1.X, Y1 = s1.Y, X2 = s2.X, Y2 = s2.Y;
    const int32_t cx = px * 256 + 128, cy = py * 256 + 128;
    int32_t mnx = X0 < X1 ? X0 : X1; mnx = mnx < X2 ? mnx : X2;
    int32_t mxx = X0 > X1 ? X0 : X1; mxx = mxx > X2 ? mxx : X2;
    int32_t mny = Y0 < Y1 ? Y0 : Y1; mny = mny < Y2 ? mny : Y2;
    int32_t mxy = Y0 > Y1 ? Y0 : Y1; mxy = mxy > Y2 ? mxy : Y2;
    if (cx < mnx || cx > mxx || cy < mny || cy > mxy) return false;
    if ((mxx - mnx) < (1 << 14) && (mxy - mny) < (1 << 14)) {
        const int32_t area2 = TOPO_MUL24(X1 - X0, Y2 - Y0) - TOPO_MUL24(Y1 - Y0, X2 - X0);
        if (area2 >= 0) return false;
        const float iA = div_f(1.0f, (float)(-area2));
        b[0] = (float)(TOPO_MUL24(Y2 - Y1, cx - X1) - TOPO_MUL24(X2 - X1, cy - Y1)) * iA;
        b[1] = (float)(TOPO_MUL24(Y0 - Y2, cx - X2) - TOPO_MUL24(X0 - X2, cy - Y2)) * iA;
        b[2] = (float)(TOPO_MUL24(Y1 - Y0, cx - X0) - TOPO_MUL24(X1 - X0, cy - Y0)) * iA;
    } else {
        const int64_t area2 = (int64_t)(X1 - X0) * (Y2 - Y0) - (int64_t)(Y1 - Y0) * (X2 - X0);
        if (area2 >= 0) return false;
        const float iA = div_f(1.0f, (float)(-area2));
        b[0] = (float)((int64_t)(Y2 - Y1) * (cx - X1) - (int64_t)(X2 - X1) * (cy - Y1)) * iA;
        b[1] = (float)((int64_t)(Y0 - Y2) * (cx - X2) - (int64_t)(X0 - X2) * (cy - Y2)) * iA;
        b[2] = (float)((int64_t)(Y1 - Y0) * (cx - X0) - (int64_t)(X1 - X0) * (cy - Y0)) * iA;
    }
    return true;
}

TOPO_HD uint64_t vis_key(float z, uint32_t id) { return ((uint64_t)f_bits(z) << 32) | id; }

// ---- lane bodies of the raster kernels ---------------------------------------------------------------------
// The pixel walks of k_raster (stage 2) and k_raster_big are written as plain functions of the lane index with the
// fragment sink passed in, so that tests/host_emul.cpp can run every lane of a work item on the CPU with a
// bounds-checking sink (tests/test_emul_cpu.py::test_big_item_lanes_*, test_raster_rows_*): each fragment must lie
// inside the target and inside the item's region, be emitted exactly once, and carry the key triangle_pixel() gives.
// (Round 1 lost a GPU box to an out-of-range visibility index in an uncommitted variant of the lane tiling below:
// DESIGN.md "The exp_direct fault".)
#if defined(__HIP_DEVICE_COMPILE__)
#define TOPO_RCP_EST(x) __builtin_amdgcn_rcpf(x)     // span ESTIMATES only: the exact integers decide every pixel
#else
#define TOPO_RCP_EST(x) (1.0f / (x))
#endif

// A triangle whose snapped vertices span < 2^14 sub-pixels (64 px) in x and y: every edge-function product fits int32.
TOPO_HD bool spans_fit_int32(int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2, int32_t Y2) {
    const int32_t mnx = X0 < X1 ? (X0 < X2 ? X0 : X2) : (X1 < X2 ? X1 : X2), mxx = X0 > X1 ? (X0 > X2 ? X0 : X2) : (X1 > X2 ? X1 : X2);
    const int32_t mny = Y0 < Y1 ? (Y0 < Y2 ? Y0 : Y2) : (Y1 < Y2 ? Y1 : Y2), mxy = Y0 > Y1 ? (Y0 > Y2 ? Y0 : Y2) : (Y1 > Y2 ? Y1 : Y2);
    return (mxx - mnx) < (1 << 14) && (mxy - mny) < (1 << 14);
}

// Stage 2 of k_raster: walk the pixel rows of one classified triangle (spans_fit_int32).  A float estimate of each
// edge's crossing narrows a row to its covered span (padded by a pixel either side); the exact integer test then
// decides every pixel, so the estimate can only cost time, never change coverage.  Coverage, barycentrics and depth
// are the values triangle_pixel() gives.  emit(pixel index, key).
template <typename Emit>
TOPO_HD void raster_rows(int32_t W, int32_t H, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2, int32_t Y2, float z0,
                         float z1, float z2, uint32_t id, Emit&& emit) {
    const int32_t mnx = X0 < X1 ? (X0 < X2 ? X0 : X2) : (X1 < X2 ? X1 : X2), mxx = X0 > X1 ? (X0 > X2 ? X0 : X2) : (X1 > X2 ? X1 : X2);
    const int32_t mny = Y0 < Y1 ? (Y0 < Y2 ? Y0 : Y2) : (Y1 < Y2 ? Y1 : Y2), mxy = Y0 > Y1 ? (Y0 > Y2 ? Y0 : Y2) : (Y1 > Y2 ? Y1 : Y2);
    const int32_t area2 = TOPO_MUL24(X1 - X0, Y2 - Y0) - TOPO_MUL24(Y1 - Y0, X2 - X0);
    const int32_t bx0 = (mnx + 127) >> 8, bx1 = (mxx - 128) >> 8, by0 = (mny + 127) >> 8, by1 = (mxy - 128) >> 8;
    const int32_t px0 = bx0 > 0 ? bx0 : 0, px1 = bx1 < W - 1 ? bx1 : W - 1;
    const int32_t py0 = by0 > 0 ? by0 : 0, py1 = by1 < H - 1 ? by1 : H - 1;
    // edges e0 = v1->v2, e1 = v2->v0, e2 = v0->v1
    const int32_t dx0 = X2 - X1, dy0 = Y2 - Y1, dx1 = X0 - X2, dy1 = Y0 - Y2, dx2 = X1 - X0, dy2 = Y1 - Y0;
    const int32_t b0 = ((dy0 > 0) || (dy0 == 0 && dx0 < 0)) ? 0 : -1;
    const int32_t b1 = ((dy1 > 0) || (dy1 == 0 && dx1 < 0)) ? 0 : -1;
    const int32_t b2 = ((dy2 > 0) || (dy2 == 0 && dx2 < 0)) ? 0 : -1;
    const int32_t cx = px0 * 256 + 128, cy = py0 * 256 + 128;
    int32_t r0 = TOPO_MUL24(dy0, cx - X1) - TOPO_MUL24(dx0, cy - Y1) + b0;      // biased: covered <=> all three >= 0
    int32_t r1 = TOPO_MUL24(dy1, cx - X2) - TOPO_MUL24(dx1, cy - Y2) + b1;
    int32_t r2 = TOPO_MUL24(dy2, cx - X0) - TOPO_MUL24(dx2, cy - Y0) + b2;
    const int32_t m0 = dy0 * 256, m1 = dy1 * 256, m2 = dy2 * 256;
    // hardware reciprocal (v_rcp_f32) is plenty for the span ESTIMATE
    const float i0 = m0 ? TOPO_RCP_EST((float)m0) : 0.0f, i1 = m1 ? TOPO_RCP_EST((float)m1) : 0.0f, i2 = m2 ? TOPO_RCP_EST((float)m2) : 0.0f;
    const float iA = div_f(1.0f, (float)(-area2));
    const float dz1 = z1 - z0, dz2 = z2 - z0;
    const int32_t nx = px1 - px0;
    for (int32_t py = py0; py <= py1; ++py) {
        // conservative span [lo, hi] (relative to px0, always inside [0, nx]) from each edge's crossing -r/m
        int32_t lo = 0, hi = nx;
        bool dead = false;
        {
            const float q0 = -(float)r0 * i0, q1 = -(float)r1 * i1, q2 = -(float)r2 * i2;
            if (m0 > 0) { const int32_t c = (int32_t)q0 - 1; lo = lo > c ? lo : c; } else if (m0 < 0) { const int32_t c = (int32_t)q0 + 1; hi = hi < c ? hi : c; } else dead |= r0 < 0;
            if (m1 > 0) { const int32_t c = (int32_t)q1 - 1; lo = lo > c ? lo : c; } else if (m1 < 0) { const int32_t c = (int32_t)q1 + 1; hi = hi < c ? hi : c; } else dead |= r1 < 0;
            if (m2 > 0) { const int32_t c = (int32_t)q2 - 1; lo = lo > c ? lo : c; } else if (m2 < 0) { const int32_t c = (int32_t)q2 + 1; hi = hi < c ? hi : c; } else dead |= r2 < 0;
        }
        if (!dead) {
            int32_t F0 = r0 + m0 * lo, F1 = r1 + m1 * lo, F2 = r2 + m2 * lo;
            for (int32_t k = lo; k <= hi; ++k) {
                if ((F0 | F1 | F2) >= 0) {
                    const float w1 = (float)(F1 - b1) * iA, w2 = (float)(F2 - b2) * iA;
                    float z = fmaf(w1, dz1, fmaf(w2, dz2, z0));
                    if (z < 1.0f) {
                        if (z < 0.0f) z = 0.0f;
                        emit((uint32_t)(py * W + px0 + k), vis_key(z, id));
                    }
                }
                F0 += m0;
                F1 += m1;
                F2 += m2;
            }
        }
        r0 -= dx0 * 256;
        r1 -= dx1 * 256;
        r2 -= dx2 * 256;
    }
}

// k_raster_big, medium triangles (spans_fit_int32): the part of the triangle's pixel box inside region (rx, ry)
// (64 px units) is tiled row-major by the wave's 64 lanes, the row length rounded up to 8/16/32/64; lane `lane`
// visits the pixels (px0 + u, py0 + v0 + k*rows) and hands them over four at a time: emit4(pix[4], key[4], py[4]).
// CONTRACT: an entry with key == kVisClear carries no fragment and its pix[] / py[] are NOT valid (lanes beyond
// the box keep stepping their pixel counter) -- a sink must not touch pix[k] without testing key[k].
template <typename Emit4>
TOPO_HD void big_medium_lane(const int32_t X[3], const int32_t Y[3], const float zv[3], uint32_t id, int32_t W, int32_t H,
                             int32_t rx, int32_t ry, uint32_t lane, Emit4&& emit4) {
    const int32_t X0 = X[0], Y0 = Y[0], X1 = X[1], Y1 = Y[1], X2 = X[2], Y2 = Y[2];
    const int32_t mnx = X0 < X1 ? (X0 < X2 ? X0 : X2) : (X1 < X2 ? X1 : X2), mxx = X0 > X1 ? (X0 > X2 ? X0 : X2) : (X1 > X2 ? X1 : X2);
    const int32_t mny = Y0 < Y1 ? (Y0 < Y2 ? Y0 : Y2) : (Y1 < Y2 ? Y1 : Y2), mxy = Y0 > Y1 ? (Y0 > Y2 ? Y0 : Y2) : (Y1 > Y2 ? Y1 : Y2);
    const int32_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);      // |factors| < 2^14: exact in int32
    if (area2 >= 0) return;
    int32_t px0 = (mnx + 127) >> 8, px1 = (mxx - 128) >> 8, py0 = (mny + 127) >> 8, py1 = (mxy - 128) >> 8;
    px0 = px0 > 0 ? px0 : 0; px1 = px1 < W - 1 ? px1 : W - 1;
    py0 = py0 > 0 ? py0 : 0; py1 = py1 < H - 1 ? py1 : H - 1;
    px0 = px0 > rx * 64 ? px0 : rx * 64; px1 = px1 < rx * 64 + 63 ? px1 : rx * 64 + 63;
    py0 = py0 > ry * 64 ? py0 : ry * 64; py1 = py1 < ry * 64 + 63 ? py1 : ry * 64 + 63;
    const int32_t bw = px1 - px0 + 1, bh = py1 - py0 + 1;
    if (bw <= 0 || bh <= 0) return;
    const int32_t dx0 = X2 - X1, dy0 = Y2 - Y1, dx1 = X0 - X2, dy1 = Y0 - Y2, dx2 = X1 - X0, dy2 = Y1 - Y0;
    const int32_t b0 = ((dy0 > 0) || (dy0 == 0 && dx0 < 0)) ? 0 : -1;
    const int32_t b1 = ((dy1 > 0) || (dy1 == 0 && dx1 < 0)) ? 0 : -1;
    const int32_t b2 = ((dy2 > 0) || (dy2 == 0 && dx2 < 0)) ? 0 : -1;
    // biased edge functions at the box's first pixel centre, and their steps per pixel in x (A) and y (-B);
    // the box is at most 64 px wide and high, so every value below stays under 2^29
    const int32_t cx0 = px0 * 256 + 128, cy0 = py0 * 256 + 128;
    const int32_t R0 = dy0 * (cx0 - X1) - dx0 * (cy0 - Y1) + b0;
    const int32_t R1 = dy1 * (cx0 - X2) - dx1 * (cy0 - Y2) + b1;
    const int32_t R2 = dy2 * (cx0 - X0) - dx2 * (cy0 - Y0) + b2;
    const int32_t A0 = dy0 * 256, A1 = dy1 * 256, A2 = dy2 * 256, B0 = dx0 * 256, B1 = dx1 * 256, B2 = dx2 * 256;
    const float iA = div_f(1.0f, (float)(-area2));
    const float z0 = zv[0], dz1 = zv[1] - zv[0], dz2 = zv[2] - zv[0];
    // lanes tile the box row-major, the row length rounded up to a power of two: 8 x 8, 16 x 4, 32 x 2 or 64 x 1
    const uint32_t sh = bw <= 8 ? 3u : bw <= 16 ? 4u : bw <= 32 ? 5u : 6u;
    const int32_t u = (int32_t)(lane & ((1u << sh) - 1u)), v0 = (int32_t)(lane >> sh), rows = 64 >> sh;
    int32_t F0 = R0 + TOPO_MUL24(A0, u) - TOPO_MUL24(B0, v0);
    int32_t F1 = R1 + TOPO_MUL24(A1, u) - TOPO_MUL24(B1, v0);
    int32_t F2 = R2 + TOPO_MUL24(A2, u) - TOPO_MUL24(B2, v0);
    const int32_t S0 = B0 * rows, S1 = B1 * rows, S2 = B2 * rows;
    uint32_t pixel = (uint32_t)((py0 + v0) * W + px0 + u);
    const uint32_t pstep = (uint32_t)(rows * W);
    const bool ucol = u < bw;
    int32_t yrow = py0 + v0;
    for (int32_t vb = 0; vb < bh; vb += 4 * rows) {
        uint32_t pix[4];
        uint64_t key[4];
        int32_t pyk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            key[k] = kVisClear;
            pix[k] = pixel;
            pyk[k] = yrow;
            if (ucol && vb + k * rows + v0 < bh && (F0 | F1 | F2) >= 0) {
                const float w1 = (float)(F1 - b1) * iA, w2 = (float)(F2 - b2) * iA;
                float z = fmaf(w1, dz1, fmaf(w2, dz2, z0));
                if (z < 1.0f) {
                    if (z < 0.0f) z = 0.0f;
                    key[k] = vis_key(z, id);
                }
            }
            F0 -= S0; F1 -= S1; F2 -= S2;
            pixel += pstep;
            yrow += rows;
        }
        emit4(pix, key, pyk);
    }
}

// k_raster_big, giants (a vertex pair >= 64 px apart): the 64-bit edge functions of triangle_setup, evaluated once per
// lane at its pixel of the first 8x8 sub-chunk of region (rx, ry) and then stepped (8 px in x: + 2048 dy, 8 px in y:
// - 2048 dx), so a sub-chunk costs three 64-bit additions instead of six 64-bit multiplications.  emit(pixel index, key, py).
template <typename Emit>
TOPO_HD void big_giant_lane(const int32_t X[3], const int32_t Y[3], const float zv[3], uint32_t id, int32_t W, int32_t H,
                            int32_t rx, int32_t ry, uint32_t lane, Emit&& emit) {
    SVert s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k].X = X[k]; s[k].Y = Y[k]; s[k].z = zv[k]; s[k].flag = kVtxOk; }
    TriSetup ts;
    if (!triangle_setup(s[0], s[1], s[2], W, H, ts)) return;
    const int32_t bx0 = ts.px0 > rx * 64 ? ts.px0 : rx * 64, bx1 = ts.px1 < rx * 64 + 63 ? ts.px1 : rx * 64 + 63;
    const int32_t by0 = ts.py0 > ry * 64 ? ts.py0 : ry * 64, by1 = ts.py1 < ry * 64 + 63 ? ts.py1 : ry * 64 + 63;
    if (bx0 > bx1 || by0 > by1) return;
    const int32_t lx = (int32_t)(lane & 7), ly = (int32_t)(lane >> 3);
    const int32_t sx0 = bx0 & ~7, sy0 = by0 & ~7;
    const int64_t cx = (int64_t)(sx0 + lx) * 256 + 128, cy = (int64_t)(sy0 + ly) * 256 + 128;
    int64_t Fr0 = ts.dy[0] * (cx - ts.ax[0]) - ts.dx[0] * (cy - ts.ay[0]) + ts.bias[0];   // biased: covered <=> all >= 0
    int64_t Fr1 = ts.dy[1] * (cx - ts.ax[1]) - ts.dx[1] * (cy - ts.ay[1]) + ts.bias[1];
    int64_t Fr2 = ts.dy[2] * (cx - ts.ax[2]) - ts.dx[2] * (cy - ts.ay[2]) + ts.bias[2];
    const int64_t ax0 = ts.dy[0] * 2048, ax1 = ts.dy[1] * 2048, ax2 = ts.dy[2] * 2048;
    const int64_t ay0 = ts.dx[0] * 2048, ay1 = ts.dx[1] * 2048, ay2 = ts.dx[2] * 2048;
    // barycentric numerators of covered pixels are in [0, |area2|]: below 2^48 (any triangle under ~46000 px
    // across) the int64 -> f32 conversion is one fma of two exact 24-bit halves, the same single rounding
    const bool narrow = -((X[1] - (int64_t)X[0]) * (Y[2] - (int64_t)Y[0]) - (Y[1] - (int64_t)Y[0]) * (X[2] - (int64_t)X[0])) < (1ll << 48);
    const int32_t b1 = (int32_t)ts.bias[1], b2 = (int32_t)ts.bias[2];
    for (int32_t sy = sy0; sy <= by1; sy += 8) {
        int64_t F0 = Fr0, F1 = Fr1, F2 = Fr2;
        const int32_t py = sy + ly;
        const bool rowin = py >= by0 && py <= by1;
        for (int32_t sx = sx0; sx <= bx1; sx += 8) {
            const int32_t px = sx + lx;
            if (rowin && px >= bx0 && px <= bx1 && (F0 | F1 | F2) >= 0) {
                const int64_t U1 = F1 - b1, U2 = F2 - b2;
                float f1, f2;
                if (narrow) {
                    f1 = fmaf((float)(int32_t)(U1 >> 24), 16777216.0f, (float)(int32_t)((uint32_t)U1 & 0xFFFFFFu));
                    f2 = fmaf((float)(int32_t)(U2 >> 24), 16777216.0f, (float)(int32_t)((uint32_t)U2 & 0xFFFFFFu));
                } else {
                    f1 = (float)U1;
                    f2 = (float)U2;
                }
                float z = fmaf(f1 * ts.iA, ts.dz1, fmaf(f2 * ts.iA, ts.dz2, ts.z0));
                if (z < 1.0f) {
                    if (z < 0.0f) z = 0.0f;
                    emit((size_t)py * W + px, vis_key(z, id), py);
                }
            }
            F0 += ax0; F1 += ax1; F2 += ax2;
        }
        Fr0 -= ay0; Fr1 -= ay1; Fr2 -= ay2;
    }
}

// ---- near-plane clipping of one triangle -------------------------------------------------------------
// Sutherland-Hodgman against z_clip >= 0, intersections always computed from the inside vertex towards the
// outside one (t = z_in / (z_in - z_out)), output fan-triangulated from the first emitted vertex.
// Returns the number of output vertices (0, 3 or 4).
TOPO_HD VFull lerp_vertex(const VFull& I, const VFull& O, float t) {
    VFull r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.clip[k] = I.clip[k] + t * (O.clip[k] - I.clip[k]);
    r.clip[2] = 0.0f;
    r.wpos = {I.wpos.x + (O.wpos.x - I.wpos.x) * t, I.wpos.y + (O.wpos.y - I.wpos.y) * t,
              I.wpos.z + (O.wpos.z - I.wpos.z) * t};
    r.wnrm = {I.wnrm.x + (O.wnrm.x - I.wnrm.x) * t, I.wnrm.y + (O.wnrm.y - I.wnrm.y) * t,
              I.wnrm.z + (O.wnrm.z - I.wnrm.z) * t};
    return r;
}

TOPO_HD VFull clip_edge(const VFull& I, const VFull& O) {   // I inside (z_clip >= 0), O outside
    return lerp_vertex(I, O, div_f(I.clip[2], I.clip[2] - O.clip[2]));
}

// Triangle number `fan` (0 or 1) of the clipped polygon, written out case by case so that nothing is indexed at
// run time (runtime-indexed arrays live in scratch memory on the GPU).  The polygon Sutherland-Hodgman emits --
// walk i = 0,1,2: emit v_i if inside, then the crossing of edge (i, i+1) if it crosses -- is, by inside mask:
//   v0        : [v0, X01, X20]          v0 v1   : [v0, v1, X12, X20]
//   v1        : [X01, v1, X12]          v1 v2   : [X01, v1, v2, X20]
//   v2        : [X12, v2, X20]          v0 v2   : [v0, X01, X12, v2]
// and fan k is (p0, p[k+1], p[k+2]).  Returns false if that triangle does not exist.
TOPO_HD bool clip_near_fan(const VFull& v0, const VFull& v1, const VFull& v2, uint32_t fan, VFull& a, VFull& b, VFull& c) {
    const uint32_t mask = (v0.clip[2] >= 0.0f ? 1u : 0u) | (v1.clip[2] >= 0.0f ? 2u : 0u) | (v2.clip[2] >= 0.0f ? 4u : 0u);
    switch (mask) {
        case 7u:
            if (fan != 0) return false;
            a = v0; b = v1; c = v2;
            return true;
        case 1u:
            if (fan != 0) return false;
            a = v0; b = clip_edge(v0, v1); c = clip_edge(v0, v2);
            return true;
        case 2u:
            if (fan != 0) return false;
            a = clip_edge(v1, v0); b = v1; c = clip_edge(v1, v2);
            return true;
        case 4u:
            if (fan != 0) return false;
            a = clip_edge(v2, v1); b = v2; c = clip_edge(v2, v0);
            return true;
        case 3u:   // [v0, v1, X12, X20]
            a = v0;
            if (fan == 0) { b = v1; c = clip_edge(v1, v2); } else { b = clip_edge(v1, v2); c = clip_edge(v0, v2); }
            return fan < 2;
        case 6u:   // [X01, v1, v2, X20]
            a = clip_edge(v1, v0);
            if (fan == 0) { b = v1; c = v2; } else { b = v2; c = clip_edge(v2, v0); }
            return fan < 2;
        case 5u:   // [v0, X01, X12, v2]
            a = v0;
            if (fan == 0) { b = clip_edge(v0, v1); c = clip_edge(v2, v1); } else { b = clip_edge(v2, v1); c = v2; }
            return fan < 2;
        default:
            return false;
    }
}

// ---- perspective-correct varyings --------------------------------------------------------------------
// a = (a0*q0 + a1*q1 + a2*q2) * (1 / (q0+q1+q2)), sums as fma chains
TOPO_HD void interpolate_q(const VFull& v0, const VFull& v1, const VFull& v2, float q0, float q1, float q2, f3& wpos, f3& wnrm) {
    const float iq = div_f(1.0f, (q0 + q1) + q2);
    wpos.x = fmaf(v2.wpos.x, q2, fmaf(v1.wpos.x, q1, v0.wpos.x * q0)) * iq;
    wpos.y = fmaf(v2.wpos.y, q2, fmaf(v1.wpos.y, q1, v0.wpos.y * q0)) * iq;
    wpos.z = fmaf(v2.wpos.z, q2, fmaf(v1.wpos.z, q1, v0.wpos.z * q0)) * iq;
    wnrm.x = fmaf(v2.wnrm.x, q2, fmaf(v1.wnrm.x, q1, v0.wnrm.x * q0)) * iq;
    wnrm.y = fmaf(v2.wnrm.y, q2, fmaf(v1.wnrm.y, q1, v0.wnrm.y * q0)) * iq;
    wnrm.z = fmaf(v2.wnrm.z, q2, fmaf(v1.wnrm.z, q1, v0.wnrm.z * q0)) * iq;
}
// Weights of an uncut triangle from its screen-space barycentrics: q_i = b_i * (1/w_i).
TOPO_HD void interpolate(const VFull& v0, const VFull& v1, const VFull& v2, const float b[3], f3& wpos, f3& wnrm) {
    interpolate_q(v0, v1, v2, b[0] * div_f(1.0f, v0.clip[3]), b[1] * div_f(1.0f, v1.clip[3]), b[2] * div_f(1.0f, v2.clip[3]), wpos, wnrm);
}
// Weights for a fragment of a primitive cut by the near plane (Raster spec 8): homogeneous barycentrics of the pixel
// centre with respect to the primitive's own three vertices, q = adj([x y w]) . (gx, gy, 1) with (gx, gy) the centre in
// normalised device coordinates.  Every product and difference is a separate rounding (no fma inside the 2x2
// determinants); the three terms of a weight are one fma chain.
TOPO_HD float det2(float a, float b, float c, float d) {
    const float p = a * b, q = c * d;
    return p - q;
}
TOPO_HD void homogeneous_weights(const VFull& v0, const VFull& v1, const VFull& v2, float W, float H, int32_t px, int32_t py, float q[3]) {
    const float gx = fmaf((float)px + 0.5f, div_f(2.0f, W), -1.0f);
    const float gy = fmaf(-((float)py + 0.5f), div_f(2.0f, H), 1.0f);
    const float *c0 = v0.clip, *c1 = v1.clip, *c2 = v2.clip;
    q[0] = fmaf(det2(c1[1], c2[3], c1[3], c2[1]), gx, fmaf(det2(c1[3], c2[0], c1[0], c2[3]), gy, det2(c1[0], c2[1], c1[1], c2[0])));
    q[1] = fmaf(det2(c2[1], c0[3], c2[3], c0[1]), gx, fmaf(det2(c2[3], c0[0], c2[0], c0[3]), gy, det2(c2[0], c0[1], c2[1], c0[0])));
    q[2] = fmaf(det2(c0[1], c1[3], c0[3], c1[1]), gx, fmaf(det2(c0[3], c1[0], c0[0], c1[3]), gy, det2(c0[0], c1[1], c0[1], c1[0])));
}

// Visibility-only version: clip position from the height alone.
TOPO_HD void vertex_clip(const TileDev& t, const ViewDev& v, uint32_t vx, uint32_t vy, float height, float clip[4]) {
    const f3 p = vertex_world(t, vx, vy, height);
    mat4_point(v.proj, p.x, p.y, p.z, clip);
}

// The triangle (after near clipping) that draw-order id `id` = (draw << 1 | fan index) names, set up for
// pixel evaluation.  Returns false if it does not exist / is culled (cannot happen for an id that won a pixel).
struct ResolvedTri {
    VFull v[3];
    SVert s[3];
    TriSetup ts;
};
// kShading (the resolve pass): a primitive cut by the near plane is not clipped -- its varyings come from its own three
// vertices (homogeneous_weights) -- so the function returns false with `cut` set and r.v[] holding those vertices.
// Without kShading (k_raster_rare: coverage and depth) the piece `fan` of the clipped polygon is produced.
template <bool kShading = false, bool kTableOnly = false>
TOPO_HD bool resolve_vertices(const TileDev& t, uint32_t tile_w, FastDiv div_hm1, uint32_t hm1, const ViewDev& view, int32_t W, int32_t H,
                              uint32_t tri, uint32_t fan, const float* ndec, ResolvedTri& r, bool* cut = nullptr) {
    // the triangle's three vertices are corners of one grid cell: two distinct longitudes, two distinct latitudes,
    // whose sin/cos pairs come from the tile's tables (same function of the same input as vertex_world)
    const uint32_t cell = tri >> 1, k = tri & 1u;
    const uint32_t ci = fastdiv(cell, div_hm1), cj = cell - ci * hm1;
    const bool even = ((ci + cj) & 1u) == 0;
    // corner offsets (render_buffer.rs:191-219): k=0: a, b, (even ? d : c);  k=1: d, c, (even ? a : b)
    const uint32_t ox[3] = {k, k, 1u - k};
    const uint32_t oy[3] = {k, 1u - k, k == 0 ? (even ? 1u : 0u) : (even ? 0u : 1u)};
    // Every load first, then the arithmetic: the three heights, three normals and eight table entries depend on nothing
    // but the tile descriptor, so they travel together (one trip to memory instead of one per vertex).
    const auto tlon = TOPO_GLOBAL_F32(t.trig_lon) + 2 * ci;
    const auto tlat = TOPO_GLOBAL_F32(t.trig_lat) + 2 * cj;
    const auto hts = TOPO_GLOBAL_F32(t.heights);
    const auto nrm = TOPO_GLOBAL_U32(t.normals);
    const size_t idx0 = (size_t)(cj + oy[0]) * tile_w + (ci + ox[0]), idx1 = (size_t)(cj + oy[1]) * tile_w + (ci + ox[1]),
                 idx2 = (size_t)(cj + oy[2]) * tile_w + (ci + ox[2]);
    const float h0 = hts[idx0], h1 = hts[idx1], h2 = hts[idx2];
    const uint32_t n0 = nrm[idx0], n1 = nrm[idx1], n2 = nrm[idx2];
    const float slo0 = tlon[0], clo0 = tlon[1], slo1 = tlon[2], clo1 = tlon[3];
    const float sla0 = tlat[0], cla0 = tlat[1], sla1 = tlat[2], cla1 = tlat[3];
    const float hq[3] = {h0, h1, h2};
    const uint32_t nq[3] = {n0, n1, n2};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        r.v[q].wpos = world_from_sincos(hq[q], oy[q] ? sla1 : sla0, oy[q] ? cla1 : cla0, ox[q] ? slo1 : slo0, ox[q] ? clo1 : clo0);
        r.v[q].wnrm = vertex_normal<kTableOnly>(t, nq[q], ndec);
        mat4_point(view.proj, r.v[q].wpos.x, r.v[q].wpos.y, r.v[q].wpos.z, r.v[q].clip);
    }
    const bool all_in = r.v[0].clip[2] >= 0.0f && r.v[1].clip[2] >= 0.0f && r.v[2].clip[2] >= 0.0f;
    if (cut) *cut = !all_in;
    if (kShading) {
        if (!all_in) return false;
    } else if (!all_in) {   // near-plane clipping: replace the vertices by those of fan triangle `fan`
        VFull a, b, c;
        if (!clip_near_fan(r.v[0], r.v[1], r.v[2], fan, a, b, c)) return false;
        r.v[0] = a; r.v[1] = b; r.v[2] = c;
    } else if (fan != 0) {
        return false;
    }
    if (clip_to_screen(r.v[0].clip, (float)W, (float)H, r.s[0]) != kVtxOk) return false;
    if (clip_to_screen(r.v[1].clip, (float)W, (float)H, r.s[1]) != kVtxOk) return false;
    if (clip_to_screen(r.v[2].clip, (float)W, (float)H, r.s[2]) != kVtxOk) return false;
    return true;
}

TOPO_HD bool resolve_triangle(const TileDev& t, uint32_t tile_w, FastDiv div_hm1, uint32_t hm1, const ViewDev& view, int32_t W,
                              int32_t H, uint32_t tri, uint32_t fan, ResolvedTri& r) {
    return resolve_vertices<false>(t, tile_w, div_hm1, hm1, view, W, H, tri, fan, nullptr, r) && triangle_setup(r.s[0], r.s[1], r.s[2], W, H, r.ts);
}

// Varyings of the fragment at pixel (px, py) whose winner is triangle `tri` (piece `fan`) of tile t: what fs_main receives.
template <bool kTableOnly = false>
TOPO_HD bool resolve_varyings(const TileDev& t, uint32_t tile_w, FastDiv div_hm1, uint32_t hm1, const ViewDev& view, int32_t W, int32_t H,
                              uint32_t tri, uint32_t fan, const float* ndec, int32_t px, int32_t py, f3& wpos, f3& wnrm) {
    ResolvedTri r;
    bool cut = false;
    float q[3];
    if (resolve_vertices<true, kTableOnly>(t, tile_w, div_hm1, hm1, view, W, H, tri, fan, ndec, r, &cut)) {
        float b[3];
        if (fan != 0 || !triangle_bary(r.s[0], r.s[1], r.s[2], px, py, b)) return false;
        q[0] = b[0] * div_f(1.0f, r.v[0].clip[3]); q[1] = b[1] * div_f(1.0f, r.v[1].clip[3]); q[2] = b[2] * div_f(1.0f, r.v[2].clip[3]);
    } else if (cut) {
        homogeneous_weights(r.v[0], r.v[1], r.v[2], (float)W, (float)H, px, py, q);
    } else {
        return false;
    }
    interpolate_q(r.v[0], r.v[1], r.v[2], q[0], q[1], q[2], wpos, wnrm);
    return true;
}

// ---- the same varyings in two steps: once per winning triangle, then per pixel ---------------------------------------
// Many pixels share a winner (the near field is made of triangles hundreds of pixels large), and two thirds of
// resolve_varyings -- three vs_main, the perspective divides, the doubled area -- depend on the triangle alone.
// resolve_setup() computes that part into a 34-word record; resolve_pixel() finishes a pixel from the record and yields
// the SAME values as resolve_varyings, bit for bit (tests/test_emul_cpu.py::test_split_resolve_equals_resolve_varyings;
// k_resolve uses either route per row).  The record is plain 32-bit words with names (floats and the halves of doubles as
// their bit patterns) and no arrays or unions in it: the compiler keeps such a struct in registers, whereas word-indexed
// access to a union ends up in scratch memory.
//
//   kind 1 (uncut): the barycentric numerators are the exact integers F_i(p) = dy_i (cx - ax_i) - dx_i (cy - ay_i), an
//     affine function of the pixel.  With |doubled area| < 2^52 every F_i of a covered pixel lies in [0, |area|], and
//     relative to a reference pixel (ox, oy) at most 63 px / 15 rows away all terms stay below 2^53: binary64 holds them
//     EXACTLY, so F_i = fma(A_i, px - ox, fma(B_i, py - oy, C_i)) with A_i = 256 dy_i, B_i = -256 dx_i, C_i = F_i(ox, oy) is
//     the same integer the int32 / int64 forms of triangle_bary() give, and the f64 -> f32 conversion rounds it once,
//     to nearest even, as the integer -> f32 conversion does.  Two fused multiply-adds and a conversion per numerator
//     replace two multiplications, three subtractions and (beyond 64 px) a 12-instruction int64 -> f32 conversion; F_0
//     comes from F_0 + F_1 + F_2 = |area|.   u0..u11 = A1 B1 C1 A2 B2 C2 (doubles, low word first), u12 u13 = |area|,
//     iw0..2 = 1 / w_k, iA = 1 / |area|.
//   kind 2 (uncut, |area| >= 2^52 or a reference value beyond 2^52 -- triangles tens of thousands of pixels across):
//     u0..u5 = X0 X1 X2 Y0 Y1 Y2, the int64 form per pixel.
//   kind 3 (primitive cut by the near plane): u0..u8 = the nine 2x2 determinants of homogeneous_weights() -- they depend
//     on the primitive alone -- so that a weight is q_i = fma(u[3i], gx, fma(u[3i+1], gy, u[3i+2])).
//   kind 0: no varyings (cannot happen for an id that won a pixel; the pixel then keeps the cleared colour)
//   wx, wy: world position x, y of the three vertices (fs_main reads world_pos.xy only); n: world normals
#define TOPO_TRIREC_WORDS(X)                                                                                                   \
    X(u0) X(u1) X(u2) X(u3) X(u4) X(u5) X(u6) X(u7) X(u8) X(u9) X(u10) X(u11) X(u12) X(u13) X(iw0) X(iw1) X(iw2) X(iA) X(wx0)  \
    X(wx1) X(wx2) X(wy0) X(wy1) X(wy2) X(n0x) X(n0y) X(n0z) X(n1x) X(n1y) X(n1z) X(n2x) X(n2y) X(n2z) X(kind)
struct TriRecord {
#define TOPO_X(f) uint32_t f;
    TOPO_TRIREC_WORDS(TOPO_X)
#undef TOPO_X
};
constexpr int kTriRecordWords = 34;
static_assert(sizeof(TriRecord) == kTriRecordWords * 4, "TriRecord is stored word by word");

TOPO_HD double f64_from_words(uint32_t lo, uint32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double((int)hi, (int)lo);
#else
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    double d;
    memcpy(&d, &u, 8);
    return d;
#endif
}
TOPO_HD uint32_t f64_lo(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__double2loint(d);
#else
    uint64_t u;
    memcpy(&u, &d, 8);
    return (uint32_t)u;
#endif
}
TOPO_HD uint32_t f64_hi(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__double2hiint(d);
#else
    uint64_t u;
    memcpy(&u, &d, 8);
    return (uint32_t)(u >> 32);
#endif
}

// (ox, oy): the reference pixel of kind 1 -- any pixel at most 63 columns / 15 rows up and left of the pixels the record
// will be asked about (k_resolve: the origin of the wave's strip).
// (Every word is a local that is assigned on every path and stored once at the end: a struct that is zeroed first and
// overwritten in the branches ends up in scratch memory.)
template <bool kTableOnly = false>
TOPO_HD void resolve_setup(const TileDev& t, uint32_t tile_w, FastDiv div_hm1, uint32_t hm1, const ViewDev& view, int32_t W, int32_t H,
                           uint32_t tri, uint32_t fan, const float* ndec, int32_t ox, int32_t oy, TriRecord& rec) {
    ResolvedTri r;
    bool cut = false;
    uint32_t kind = 0, iw0 = 0, iw1 = 0, iw2 = 0, iA = 0;
    uint32_t u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0, u8 = 0, u9 = 0, u10 = 0, u11 = 0, u12 = 0, u13 = 0;
    if (resolve_vertices<true, kTableOnly>(t, tile_w, div_hm1, hm1, view, W, H, tri, fan, ndec, r, &cut)) {
        const int32_t X0 = r.s[0].X, Y0 = r.s[0].Y, X1 = r.s[1].X, Y1 = r.s[1].Y, X2 = r.s[2].X, Y2 = r.s[2].Y;
        const int64_t area2 = (int64_t)(X1 - X0) * (Y2 - Y0) - (int64_t)(Y1 - Y0) * (X2 - X0);
        if (fan == 0 && area2 < 0) {
            // edges e1 = v2 -> v0 (weight of v1), e2 = v0 -> v1 (weight of v2), as triangle_bary() numbers them
            const int64_t cx = (int64_t)ox * 256 + 128, cy = (int64_t)oy * 256 + 128;
            const int64_t dy1 = Y0 - Y2, dx1 = X0 - X2, dy2 = Y1 - Y0, dx2 = X1 - X0;
            const int64_t C1 = dy1 * (cx - X2) - dx1 * (cy - Y2), C2 = dy2 * (cx - X0) - dx2 * (cy - Y0);
            constexpr int64_t kLim = 1ll << 52;
            const bool affine = -area2 < kLim && C1 > -kLim && C1 < kLim && C2 > -kLim && C2 < kLim;
            const double A1 = (double)(dy1 * 256), B1 = (double)(dx1 * -256), A2 = (double)(dy2 * 256), B2 = (double)(dx2 * -256);
            const double C1d = (double)C1, C2d = (double)C2, AR = (double)(-area2);
            kind = affine ? 1u : 2u;
            u0 = affine ? f64_lo(A1) : (uint32_t)X0; u1 = affine ? f64_hi(A1) : (uint32_t)X1;
            u2 = affine ? f64_lo(B1) : (uint32_t)X2; u3 = affine ? f64_hi(B1) : (uint32_t)Y0;
            u4 = affine ? f64_lo(C1d) : (uint32_t)Y1; u5 = affine ? f64_hi(C1d) : (uint32_t)Y2;
            u6 = f64_lo(A2); u7 = f64_hi(A2); u8 = f64_lo(B2); u9 = f64_hi(B2); u10 = f64_lo(C2d); u11 = f64_hi(C2d);
            u12 = f64_lo(AR); u13 = f64_hi(AR);
            iA = f_bits(div_f(1.0f, (float)(-area2)));
            iw0 = f_bits(div_f(1.0f, r.v[0].clip[3]));
            iw1 = f_bits(div_f(1.0f, r.v[1].clip[3]));
            iw2 = f_bits(div_f(1.0f, r.v[2].clip[3]));
        }
    } else if (cut) {
        kind = 3;
        const float *c0 = r.v[0].clip, *c1 = r.v[1].clip, *c2 = r.v[2].clip;      // the determinants of homogeneous_weights()
        u0 = f_bits(det2(c1[1], c2[3], c1[3], c2[1])); u1 = f_bits(det2(c1[3], c2[0], c1[0], c2[3])); u2 = f_bits(det2(c1[0], c2[1], c1[1], c2[0]));
        u3 = f_bits(det2(c2[1], c0[3], c2[3], c0[1])); u4 = f_bits(det2(c2[3], c0[0], c2[0], c0[3])); u5 = f_bits(det2(c2[0], c0[1], c2[1], c0[0]));
        u6 = f_bits(det2(c0[1], c1[3], c0[3], c1[1])); u7 = f_bits(det2(c0[3], c1[0], c0[0], c1[3])); u8 = f_bits(det2(c0[0], c1[1], c0[1], c1[0]));
    }
    rec.kind = kind;
    rec.u0 = u0; rec.u1 = u1; rec.u2 = u2; rec.u3 = u3; rec.u4 = u4; rec.u5 = u5; rec.u6 = u6; rec.u7 = u7; rec.u8 = u8; rec.u9 = u9;
    rec.u10 = u10; rec.u11 = u11; rec.u12 = u12; rec.u13 = u13;
    rec.iw0 = iw0; rec.iw1 = iw1; rec.iw2 = iw2; rec.iA = iA;
    rec.wx0 = f_bits(r.v[0].wpos.x); rec.wx1 = f_bits(r.v[1].wpos.x); rec.wx2 = f_bits(r.v[2].wpos.x);
    rec.wy0 = f_bits(r.v[0].wpos.y); rec.wy1 = f_bits(r.v[1].wpos.y); rec.wy2 = f_bits(r.v[2].wpos.y);
    rec.n0x = f_bits(r.v[0].wnrm.x); rec.n0y = f_bits(r.v[0].wnrm.y); rec.n0z = f_bits(r.v[0].wnrm.z);
    rec.n1x = f_bits(r.v[1].wnrm.x); rec.n1y = f_bits(r.v[1].wnrm.y); rec.n1z = f_bits(r.v[1].wnrm.z);
    rec.n2x = f_bits(r.v[2].wnrm.x); rec.n2y = f_bits(r.v[2].wnrm.y); rec.n2z = f_bits(r.v[2].wnrm.z);
}

// Where a pixel is, in the three forms the record kinds use: k_resolve keeps the column forms per lane and strip, the row
// forms per row (wave-uniform).
struct PixelAt {
    int32_t px, py;        // kind 2
    double dx, dy;         // kind 1: px - ox, py - oy of the record's reference pixel
    float gx, gy;          // kind 3: the pixel centre in normalised device coordinates (homogeneous_weights)
};
TOPO_HD float pixel_gx(int32_t px, float two_over_w) { return fmaf((float)px + 0.5f, two_over_w, -1.0f); }      // two_over_w = div_f(2, W)
TOPO_HD float pixel_gy(int32_t py, float two_over_h) { return fmaf(-((float)py + 0.5f), two_over_h, 1.0f); }
TOPO_HD PixelAt pixel_at(int32_t W, int32_t H, int32_t px, int32_t py, int32_t ox, int32_t oy) {
    return PixelAt{px, py, (double)(px - ox), (double)(py - oy), pixel_gx(px, div_f(2.0f, (float)W)), pixel_gy(py, div_f(2.0f, (float)H))};
}

// world_pos.xy and the world normal of the fragment at `at`; false = no varyings (the pixel keeps the cleared colour).
// (Evaluating kinds 1 and 3 for every pixel and selecting the weights -- no divergent branch -- measured 3 % slower in
// k_resolve than the branches: the binary64 forms run at half rate and a third of the c4 near field is kind 3.)
TOPO_HD bool resolve_pixel(const TriRecord& rec, const PixelAt& at, float& wposx, float& wposy, f3& wnrm) {
    const double F1 = fma(f64_from_words(rec.u0, rec.u1), at.dx, fma(f64_from_words(rec.u2, rec.u3), at.dy, f64_from_words(rec.u4, rec.u5)));
    const double F2 = fma(f64_from_words(rec.u6, rec.u7), at.dx, fma(f64_from_words(rec.u8, rec.u9), at.dy, f64_from_words(rec.u10, rec.u11)));
    const double F0 = (f64_from_words(rec.u12, rec.u13) - F1) - F2;
    float f0 = (float)F0, f1 = (float)F1, f2 = (float)F2;
#ifndef TOPO_PIXEL_UNCONDITIONAL      // (experiment build TOPO_PIXEL_UNCONDITIONAL: kinds 1 and 3 both evaluated and selected -- measured 3 % slower: the f64 forms are half rate)
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
    if (rec.kind == 3u) {
        q0 = fmaf(bits_f(rec.u0), at.gx, fmaf(bits_f(rec.u1), at.gy, bits_f(rec.u2)));
        q1 = fmaf(bits_f(rec.u3), at.gx, fmaf(bits_f(rec.u4), at.gy, bits_f(rec.u5)));
        q2 = fmaf(bits_f(rec.u6), at.gx, fmaf(bits_f(rec.u7), at.gy, bits_f(rec.u8)));
    } else {
        if (rec.kind == 2u) {
            const int32_t X0 = (int32_t)rec.u0, X1 = (int32_t)rec.u1, X2 = (int32_t)rec.u2, Y0 = (int32_t)rec.u3, Y1 = (int32_t)rec.u4, Y2 = (int32_t)rec.u5;
            const int32_t cx = at.px * 256 + 128, cy = at.py * 256 + 128;
            f0 = (float)((int64_t)(Y2 - Y1) * (cx - X1) - (int64_t)(X2 - X1) * (cy - Y1));
            f1 = (float)((int64_t)(Y0 - Y2) * (cx - X2) - (int64_t)(X0 - X2) * (cy - Y2));
            f2 = (float)((int64_t)(Y1 - Y0) * (cx - X0) - (int64_t)(X1 - X0) * (cy - Y0));
        }
        const float iA = bits_f(rec.iA);
        q0 = (f0 * iA) * bits_f(rec.iw0); q1 = (f1 * iA) * bits_f(rec.iw1); q2 = (f2 * iA) * bits_f(rec.iw2);
    }
#else
    if (rec.kind == 2u) {
        const int32_t X0 = (int32_t)rec.u0, X1 = (int32_t)rec.u1, X2 = (int32_t)rec.u2, Y0 = (int32_t)rec.u3, Y1 = (int32_t)rec.u4, Y2 = (int32_t)rec.u5;
        const int32_t cx = at.px * 256 + 128, cy = at.py * 256 + 128;
        f0 = (float)((int64_t)(Y2 - Y1) * (cx - X1) - (int64_t)(X2 - X1) * (cy - Y1));
        f1 = (float)((int64_t)(Y0 - Y2) * (cx - X2) - (int64_t)(X0 - X2) * (cy - Y2));
        f2 = (float)((int64_t)(Y1 - Y0) * (cx - X0) - (int64_t)(X1 - X0) * (cy - Y0));
    }
    const float iA = bits_f(rec.iA);
    const float a0 = (f0 * iA) * bits_f(rec.iw0), a1 = (f1 * iA) * bits_f(rec.iw1), a2 = (f2 * iA) * bits_f(rec.iw2);
    const float h0 = fmaf(bits_f(rec.u0), at.gx, fmaf(bits_f(rec.u1), at.gy, bits_f(rec.u2)));
    const float h1 = fmaf(bits_f(rec.u3), at.gx, fmaf(bits_f(rec.u4), at.gy, bits_f(rec.u5)));
    const float h2 = fmaf(bits_f(rec.u6), at.gx, fmaf(bits_f(rec.u7), at.gy, bits_f(rec.u8)));
    const bool cut = rec.kind == 3u;
    const float q0 = cut ? h0 : a0, q1 = cut ? h1 : a1, q2 = cut ? h2 : a2;
#endif
    const float iq = div_f(1.0f, (q0 + q1) + q2);
    wposx = fmaf(bits_f(rec.wx2), q2, fmaf(bits_f(rec.wx1), q1, bits_f(rec.wx0) * q0)) * iq;
    wposy = fmaf(bits_f(rec.wy2), q2, fmaf(bits_f(rec.wy1), q1, bits_f(rec.wy0) * q0)) * iq;
    wnrm.x = fmaf(bits_f(rec.n2x), q2, fmaf(bits_f(rec.n1x), q1, bits_f(rec.n0x) * q0)) * iq;
    wnrm.y = fmaf(bits_f(rec.n2y), q2, fmaf(bits_f(rec.n1y), q1, bits_f(rec.n0y) * q0)) * iq;
    wnrm.z = fmaf(bits_f(rec.n2z), q2, fmaf(bits_f(rec.n1z), q1, bits_f(rec.n0z) * q0)) * iq;
    return rec.kind != 0u;
}

// ---- overlay pass: line_shader.wgsl + LineRenderer's pipeline state ---------------------------------
// The label leader lines and label backgrounds are CPU-tessellated (lyon) into `GpuVertex` triangles
// (line_renderer.rs:18-25: position vec2, normal vec2, color vec3, z_index i32 = 32 bytes) and drawn INTO the post pass:
// opaque (blend None), CCW front / back-face culled (:271-275), depth test Greater with write against the post pass's
// depth attachment (pipeline.rs:24-32), which the full-screen post quad has just filled with z = 1/4096
// (postprocessing_shader.wgsl:62).  vs_main (line_shader.wgsl:27-41): z = f32(z_index) / 4096;
// p = (position + normal * width) * (1, -1); clip = (2 p.x / res_w - 1, 2 p.y / res_h + 1, z, 1).
struct OverlayVertex {         // GpuVertex, #[repr(C)]
    float position[2], normal[2], color[3];
    int32_t z_index;
};
constexpr uint32_t kOverlayBaseDepthBits = 0x39800000u;      // 1.0f / 4096.0f: what the post quad leaves in the depth attachment
constexpr uint64_t kOverlayClear = ((uint64_t)kOverlayBaseDepthBits << 32) | 0xFFFFFFFFull;

TOPO_HD int overlay_vertex(const OverlayVertex& v, float width, float W, float H, SVert& s) {
    const float z = div_f((float)v.z_index, 4096.0f);
    const float px = v.position[0] + v.normal[0] * width, py = -(v.position[1] + v.normal[1] * width);
    float clip[4] = {div_f(2.0f * px, W) - 1.0f, div_f(2.0f * py, H) + 1.0f, z, 1.0f};
    if (!(z >= 0.0f)) clip[2] = 0.0f;      // (never: z_index >= 0; keeps clip_to_screen's near-plane flag out of it)
    const int flag = clip_to_screen(clip, W, H, s);      // w = 1: the perspective divide is exact
    s.z = z;
    return flag;
}
// key of an overlay fragment: larger z wins (Greater), on equal z the EARLIER triangle (the later one fails the test)
TOPO_HD uint64_t overlay_key(float z, uint32_t tri) { return ((uint64_t)f_bits(z) << 32) | (0xFFFFFFFFu - tri); }
// colour of the fragment of triangle (v0, v1, v2) at pixel (px, py): varyings with w = 1 (Raster spec 8 with q_i = b_i)
TOPO_HD bool overlay_color(const OverlayVertex& v0, const OverlayVertex& v1, const OverlayVertex& v2, float width, int32_t W, int32_t H,
                           int32_t px, int32_t py, float rgb[3]) {
    SVert s0, s1, s2;
    if (overlay_vertex(v0, width, (float)W, (float)H, s0) != kVtxOk || overlay_vertex(v1, width, (float)W, (float)H, s1) != kVtxOk ||
        overlay_vertex(v2, width, (float)W, (float)H, s2) != kVtxOk)
        return false;
    float b[3];
    if (!triangle_bary(s0, s1, s2, px, py, b)) return false;
    const float iq = div_f(1.0f, (b[0] + b[1]) + b[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = fmaf(v2.color[k], b[2], fmaf(v1.color[k], b[1], v0.color[k] * b[0])) * iq;
    return true;
}

// ---- overlay pass, text (text_renderer.rs:198-204, :259-291 over glyphon 0.10.0's pipeline) ---------------------------------
// glyphon's instance record `GlyphToRender` (#[repr(C)], 28 bytes): the quad pos .. pos + dim (pixels) shows the atlas texels
// uv .. uv + dim, 1 : 1; color = a << 24 | r << 16 | g << 8 | b; content_type_with_srgb = (1 = mask atlas, 1 = decode the colour's
// r, g, b from sRGB: ColorMode::Accurate on an *Srgb surface).  Its fragment is (color.rgb, color.a * mask) under
// BlendState::ALPHA_BLENDING, in linear light on an *Srgb surface; depth test Greater with write at one depth per call
// (100 / 4096 in the reference), so the FIRST quad over a pixel keeps it.
struct GlyphInstance {
    int32_t pos[2];
    uint16_t dim[2], uv[2];
    uint32_t color;
    uint16_t content_type_with_srgb[2];
    float depth;      // (not read: the call's depth applies to every glyph, as the reference's constant does)
};
static_assert(sizeof(GlyphInstance) == 28, "GlyphToRender is 28 bytes");
// The texel a glyph leaves at a pixel it owns: `dst` = the surface texel in memory order, `mask` = the atlas texel.
// `decode` = the 256-entry sRGB decode table; encode through the 255 thresholds (srgb_encode).
TOPO_HD uint32_t glyph_blend(const GlyphInstance& g, uint32_t mask, uint32_t dst, bool srgb_target, bool bgra, const float* thresh, const float* decode) {
    const uint32_t c = g.color;
    const uint32_t c8[3] = {(c >> 16) & 255u, (c >> 8) & 255u, c & 255u};
    const float sa = from_unorm8(c >> 24) * from_unorm8(mask);
    uint32_t d8[4] = {dst & 255u, (dst >> 8) & 255u, (dst >> 16) & 255u, dst >> 24};
    if (bgra) { const uint32_t t = d8[0]; d8[0] = d8[2]; d8[2] = t; }
    uint32_t r8[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float src = g.content_type_with_srgb[1] == 1 ? decode[c8[k]] : from_unorm8(c8[k]);
        const float d = srgb_target ? decode[d8[k]] : from_unorm8(d8[k]);
        const float v = src * sa + d * (1.0f - sa);
        r8[k] = srgb_target ? srgb_encode(thresh, v) : to_unorm8(v);
    }
    r8[3] = to_unorm8(sa + from_unorm8(d8[3]) * (1.0f - sa));
    if (bgra) { const uint32_t t = r8[0]; r8[0] = r8[2]; r8[2] = t; }
    return r8[0] | (r8[1] << 8) | (r8[2] << 16) | (r8[3] << 24);
}

// ---- peak visibility (render_engine.rs:338-396; glam Mat4::project_point3 + camera.rs:12-14) ---------------
// This is CPU code in the reference (glam, SSE2: separate multiplies and adds, true divisions), restated as is.
// Returns true when the peak projects inside the open NDC cube; then (x_pos, y_pos) is its pixel and peak_dist the
// distance the reference compares (minus 10 m) with the terrain's.
TOPO_HD bool project_peak(const float* m, float x, float y, float z, float w, float h, uint32_t& x_pos, uint32_t& y_pos,
                          float& peak_dist) {
    float res[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = m[r] * x;
        t = m[4 + r] * y + t;
        t = m[8 + r] * z + t;
        res[r] = m[12 + r] + t;
    }
    const float px = res[0] / res[3], py = res[1] / res[3], pz = res[2] / res[3];
    if (!(px > -1.0f && px < 1.0f && py > -1.0f && py < 1.0f && pz < 1.0f)) return false;
    // Rust `as u32`: truncation toward zero, saturating (the operands are in [0, size) here)
    x_pos = (uint32_t)(0.5f * (px + 1.0f) * w);
    y_pos = (uint32_t)(-0.5f * (py - 1.0f) * h);
    peak_dist = linear_depth(pz);
    return true;
}

}  // namespace topo
