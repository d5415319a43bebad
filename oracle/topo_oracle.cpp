/*
 * oracle/topo_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from
 * the product path (topo-renderer_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference (krzyz/topo-renderer) has no test, golden image or fixture on the
 * terrain render path and cannot be built here (no cargo/rustc/Vulkan), so this
 * file is a CPU restatement from the reference's source text alone.  Every function cites the
 * reference lines it follows.  Where WebGPU/WGSL leave behaviour to the implementation (sin/cos
 * precision, rasteriser sub-pixel grid, sRGB conversion rounding) the build freezes one definition,
 * written down in DESIGN.md ("Arithmetic spec", "Raster spec") and restated literally here.
 *
 * Deliberately boring: one frame = clear, draw every tile in BTreeMap order, every triangle in
 * index order through a classic z-buffer with the `Less` test, shade each passing fragment, then
 * the full-screen post pass.  No binning, no culling, no deferred shading.
 */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "oracle_math.h"

using namespace omath;

namespace {

/* ------------------------------------------------------------------------------------------
 * ABI structs: topo-renderer/src/render/data.rs:33-41 (Uniforms, 160 B), :74-80
 * (PostprocessingUniforms, 16 B), :113-121 (TerrainUniforms, 96 B).  Matrices column-major.
 * ------------------------------------------------------------------------------------------ */
struct Uniforms {
    float camera_proj[16];
    float normal_proj[16];
    float camera_pos[4];
    float sun_direction[3];
    int32_t view_mode;
};
static_assert(sizeof(Uniforms) == 160, "Uniforms layout");
struct PostUniforms {
    float viewport[2];
    float pixelize_n;
    float pad;
};
static_assert(sizeof(PostUniforms) == 16, "PostprocessingUniforms layout");
struct TerrainUniforms {
    float raster_point[2];
    float model_point[2];
    float pixel_scale[2];
    float size[2];
    float normal_to_world_rot[16];
};
static_assert(sizeof(TerrainUniforms) == 96, "TerrainUniforms layout");

/* ------------------------------------------------------------------------------------------
 * sRGB conversions.  The reference gets both from the texture unit (render target and surface are
 * *Srgb formats: render_engine.rs:77-84, terrain_renderer.rs:88-93; decode when the post pass samples
 * t_render: postprocessing_shader.wgsl:74).  Spec: ideal IEC 61966-2-1 curve, exactly rounded.
 * The oracle derives its own tables in long double at start-up; tests check them against the
 * product's committed tables.
 * ------------------------------------------------------------------------------------------ */
struct SrgbTables {
    float decode[256];
    float thresh[255]; /* thresh[i-1] = smallest f32 that encodes to >= i */
    SrgbTables() {
        for (int c = 0; c < 256; ++c) {
            long double s = (long double)c / 255.0L;
            long double l = s <= 0.04045L ? s / 12.92L : powl((s + 0.055L) / 1.055L, 2.4L);
            decode[c] = (float)l;
        }
        for (int i = 1; i < 256; ++i) {
            long double s = ((long double)i - 0.5L) / 255.0L;
            long double l = s <= 0.04045L ? s / 12.92L : powl((s + 0.055L) / 1.055L, 2.4L);
            float f = (float)l;
            if ((long double)f < l) f = nextafterf(f, INFINITY);
            thresh[i - 1] = f;
        }
    }
    uint8_t encode(float l) const {
        /* number of thresholds <= l (NaN -> 0) */
        int lo = 0, hi = 255; /* invariant: thresh[0..lo) <= l, thresh[hi..) > l */
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (thresh[mid] <= l) lo = mid + 1; else hi = mid;
        }
        return (uint8_t)lo;
    }
};
const SrgbTables& srgb() { static SrgbTables t; return t; }

/* ------------------------------------------------------------------------------------------
 * Tile key: BTreeMap<GeoLocation, RenderBuffer> order (terrain_renderer.rs:30,407).  GeoLocation
 * derives Ord over {latitude{degree,direction}, longitude{degree,direction}} with S<N, W<E
 * (topo-common/src/lib.rs:7-38); from_coord maps sign>0 to N/E, everything else to S/W (:102-121).
 * ------------------------------------------------------------------------------------------ */
using GeoKey = std::tuple<int, int, int, int>;
GeoKey geo_key(int lat, int lon) {
    return GeoKey(lat < 0 ? -lat : lat, lat > 0 ? 1 : 0, lon < 0 ? -lon : lon, lon > 0 ? 1 : 0);
}

struct Tile {
    int lat, lon;
    uint32_t w, h;
    std::vector<float> hgt;   /* R32Float texture, row-major, row 0 = north (render_buffer.rs:76-90) */
    std::vector<uint8_t> nrm; /* Rgba8Unorm, zero-initialised (texture.rs:162-199) */
    TerrainUniforms tu;
    float height(int x, int y) const { return hgt[(size_t)y * w + x]; }
    void store_normal(int x, int y, v3 n) { /* textureStore(vec4(normal, 0.0)) */
        uint8_t* p = &nrm[((size_t)y * w + x) * 4];
        p[0] = unorm8(n.x); p[1] = unorm8(n.y); p[2] = unorm8(n.z); p[3] = unorm8(0.0f);
    }
};

/* to_latitude: compute_normals_shader.wgsl:15-20 */
float to_latitude(int position_y, const TerrainUniforms& t) {
    return ((float)position_y - t.raster_point[1]) * -t.pixel_scale[1] + t.model_point[1];
}
/* calc_normal: compute_normals_shader.wgsl:53-58 */
v3 calc_normal(v3 left, v3 right, v3 top, v3 bottom) {
    v3 x = sub(right, left);
    v3 y = sub(top, bottom);
    return cross(x, y);
}
v3 encode_normal(v3 n) { /* normal = 0.5 * (normal + vec3f(1)) */
    return {0.5f * (n.x + 1.0f), 0.5f * (n.y + 1.0f), 0.5f * (n.z + 1.0f)};
}

/* compute_normals: compute_normals_shader.wgsl:22-51 (launch: compute_pipeline.rs:64,74) */
void normals_interior(Tile& t) {
    const int W = (int)t.w, H = (int)t.h;
    for (int cy = 0; cy < H; ++cy)
        for (int cx = 0; cx < W; ++cx) {
            if (cx >= W - 1 || cy >= H - 1 || cx < 1 || cy < 1) continue;
            float latitude = to_latitude(cy, t.tu);
            float x = radians(t.tu.pixel_scale[0]) * R0;
            float y = radians(t.tu.pixel_scale[1]) * R0 * cos_spec(radians(latitude));
            v3 top = {0.0f, y, t.height(cx, cy - 1)};
            v3 left = {-x, 0.0f, t.height(cx - 1, cy)};
            v3 right = {x, 0.0f, t.height(cx + 1, cy)};
            v3 bottom = {0.0f, -y, t.height(cx, cy + 1)};
            v3 n = normalize(calc_normal(left, right, top, bottom));
            t.store_normal(cx, cy, encode_normal(n));
        }
}

/* compute_normals_left_right: compute_normals_edge_shader.wgsl:25-64.  `u` is the NEWLY ADDED
 * tile's uniforms whichever side it is on (terrain_renderer.rs:275).  The guard uses dimensions.x
 * although the invocation index runs along y (:33); invocations that would fall outside the
 * texture (non-square tiles only) are skipped (WebGPU drops out-of-bounds texel stores). */
void normals_edge_lr(Tile& lt, Tile& rb, const TerrainUniforms& u) {
    const int W = (int)lt.w, H = (int)lt.h;
    for (int id = 0; id < W; ++id) {
        if (id < 1 || id >= W - 1) continue;
        if (id >= H - 1) continue;
        float latitude = to_latitude(id, u);
        float x = radians(fabsf(u.pixel_scale[0])) * R0;
        float y = radians(fabsf(u.pixel_scale[1])) * R0 * cos_spec(radians(latitude));
        const int lx = W - 1, ly = id; /* coords_left */
        const int rx = 0, ry = id;     /* coords_right */
        v3 top = {0.0f, y, lt.height(lx, ly - 1)};
        v3 left = {-x, 0.0f, lt.height(lx - 1, ly)};
        v3 right = {x, 0.0f, rb.height(rx + 1, ry)};
        v3 bottom = {0.0f, -y, lt.height(lx, ly + 1)};
        v3 n = encode_normal(normalize(calc_normal(left, right, top, bottom)));
        lt.store_normal(lx, ly, n);
        rb.store_normal(rx, ry, n);
    }
}

/* compute_normals_top_bottom: compute_normals_edge_shader.wgsl:66-105; latitude fixed at row H-1 of
 * the uniforms passed in (:85). */
void normals_edge_tb(Tile& lt, Tile& rb, const TerrainUniforms& u) {
    const int W = (int)lt.w, H = (int)lt.h;
    for (int id = 0; id < W; ++id) {
        if (id < 1 || id >= W - 1) continue;
        float latitude = to_latitude(H - 1, u);
        float x = radians(fabsf(u.pixel_scale[0])) * R0;
        float y = radians(fabsf(u.pixel_scale[1])) * R0 * cos_spec(radians(latitude));
        const int tx = id, ty = H - 1; /* coords_top */
        const int bx = id, by = 0;     /* coords_bottom */
        v3 top = {0.0f, y, lt.height(tx, ty - 1)};
        v3 left = {-x, 0.0f, lt.height(tx - 1, ty)};
        v3 right = {x, 0.0f, lt.height(tx + 1, ty)};
        v3 bottom = {0.0f, -y, rb.height(bx, by + 1)};
        v3 n = encode_normal(normalize(calc_normal(left, right, top, bottom)));
        lt.store_normal(tx, ty, n);
        rb.store_normal(bx, by, n);
    }
}

/* compute_normals_corner: compute_normals_corner_shader.wgsl:29-63.  `top` is read from the
 * bottom-right tile at (0, H-2) exactly as written (:49). */
void normals_corner(Tile& lt, Tile& rt, Tile& lb, Tile& rb, const TerrainUniforms& u) {
    const int W = (int)lt.w, H = (int)lt.h;
    float latitude = to_latitude(H - 1, u);
    float x = radians(fabsf(u.pixel_scale[0])) * R0;
    float y = radians(fabsf(u.pixel_scale[1])) * R0 * cos_spec(radians(latitude));
    const int tlx = W - 1, tly = H - 1; /* coords_top_left */
    const int trx = 0, try_ = H - 1;    /* coords_top_right */
    const int blx = W - 1, bly = 0;     /* coords_bottom_left */
    const int brx = 0, bry = 0;         /* coords_bottom_right */
    v3 top = {0.0f, y, rb.height(trx, try_ - 1)};
    v3 left = {-x, 0.0f, lt.height(tlx - 1, tly)};
    v3 right = {x, 0.0f, rt.height(trx + 1, try_)};
    v3 bottom = {0.0f, -y, lb.height(blx, bly + 1)};
    v3 n = encode_normal(normalize(calc_normal(left, right, top, bottom)));
    lt.store_normal(tlx, tly, n);
    rt.store_normal(trx, try_, n);
    lb.store_normal(blx, bly, n);
    rb.store_normal(brx, bry, n);
}

/* ------------------------------------------------------------------------------------------
 * glam 0.31.0 restatement (host side; Cargo.lock:1272-1273).  Source of the crate is not available
 * offline; formulas are the published ones.  Host libm sinf/cosf as Rust's f32::sin_cos.
 * ------------------------------------------------------------------------------------------ */
float to_radians_rs(float deg) { return deg * 0.017453292519943295f; } /* f32::to_radians */

/* Mat3::from_euler(EulerRot::XYZEx, a, b, c) = Rz(c)*Ry(b)*Rx(a) (extrinsic x, then y, then z),
 * Shoemake form with i=a, j=b, h=c: used at render/data.rs:128-133 and camera.rs:46-51. */
void mat3_from_euler_xyz_ex(float a, float b, float c, float m[9] /* column-major 3x3 */) {
    float si = sinf(a), ci = cosf(a), sj = sinf(b), cj = cosf(b), sh = sinf(c), ch = cosf(c);
    float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
    /* m(row, col) */
    float m00 = cj * ch, m01 = sj * sc - cs, m02 = sj * cc + ss;
    float m10 = cj * sh, m11 = sj * ss + cc, m12 = sj * cs - sc;
    float m20 = -sj, m21 = cj * si, m22 = cj * ci;
    m[0] = m00; m[1] = m10; m[2] = m20;
    m[3] = m01; m[4] = m11; m[5] = m21;
    m[6] = m02; m[7] = m12; m[8] = m22;
}

/* TerrainUniforms::new: render/data.rs:124-151 */
TerrainUniforms terrain_uniforms_new(const float raster_point[2], const float model_point[2],
                                     const float pixel_scale[2], uint32_t w, uint32_t h) {
    TerrainUniforms t;
    memset(&t, 0, sizeof t);
    float latitude = model_point[1], longitude = model_point[0];
    float m3[9];
    mat3_from_euler_xyz_ex(0.0f, to_radians_rs(90.0f - latitude), to_radians_rs(longitude), m3);
    t.raster_point[0] = raster_point[0]; t.raster_point[1] = raster_point[1];
    t.model_point[0] = model_point[0]; t.model_point[1] = model_point[1];
    t.pixel_scale[0] = pixel_scale[0]; t.pixel_scale[1] = pixel_scale[1];
    t.size[0] = (float)w; t.size[1] = (float)h;
    /* Mat4::from_mat3 */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) t.normal_to_world_rot[c * 4 + r] = m3[c * 3 + r];
    t.normal_to_world_rot[15] = 1.0f;
    return t;
}

/* geometry::transform: render/geometry.rs:12-20 */
v3 geometry_transform(float h, float lon_deg, float lat_deg) {
    float r = R0 + h;
    float lon = to_radians_rs(lon_deg), lat = to_radians_rs(lat_deg);
    return {r * cosf(lat) * cosf(lon), r * cosf(lat) * sinf(lon), r * sinf(lat)};
}

v3 glam_normalize(v3 v) { /* Vec3::normalize = self * (1 / length) */
    float rl = 1.0f / sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    return {v.x * rl, v.y * rl, v.z * rl};
}

struct quat { float x, y, z, w; };

/* Quat::from_rotation_arc(from, to) */
quat quat_from_rotation_arc(v3 from, v3 to) {
    const float ONE_MINUS_EPS = 1.0f - 2.0f * 1.1920929e-7f;
    float d = dot(from, to);
    if (d > ONE_MINUS_EPS) return {0, 0, 0, 1};
    if (d < -ONE_MINUS_EPS) {
        /* from.any_orthonormal_vector(), rotation by pi: q = (axis*sin(pi/2), cos(pi/2)) */
        float sign = copysignf(1.0f, from.z);
        float a = -1.0f / (sign + from.z);
        float b = from.x * from.y * a;
        v3 axis = {b, sign + from.y * from.y * a, -from.y};
        float s = sinf(3.14159265358979323846f * 0.5f), c = cosf(3.14159265358979323846f * 0.5f);
        return {axis.x * s, axis.y * s, axis.z * s, c};
    }
    v3 c = cross(from, to);
    quat q = {c.x, c.y, c.z, 1.0f + d};
    /* Vec4 normalize, SSE2 dot4 order (x2+z2)+(y2+w2) */
    float l2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
    float rl = 1.0f / sqrtf(l2);
    return {q.x * rl, q.y * rl, q.z * rl, q.w * rl};
}

/* Quat * Vec3 */
v3 quat_mul_vec3(quat q, v3 r) {
    float w = q.w;
    v3 b = {q.x, q.y, q.z};
    float b2 = dot(b, b);
    v3 t0 = scale(r, w * w - b2);
    v3 t1 = scale(b, dot(r, b) * 2.0f);
    v3 t2 = scale(cross(b, r), w * 2.0f);
    return add(add(t0, t1), t2);
}

void mat4_mul(const float* a, const float* b, float* out) { /* Mat4 * Mat4, column by column */
    for (int c = 0; c < 4; ++c) {
        float col[4];
        mat4_mul_vec4(a, b[c * 4 + 0], b[c * 4 + 1], b[c * 4 + 2], b[c * 4 + 3], col);
        memcpy(out + c * 4, col, sizeof col);
    }
}

/* Camera::{up, direction, get_view, build_view_proj_matrix}: data/camera.rs:97-128;
 * LightAngle::to_vec3: :44-53; Uniforms::new: render/data.rs:44-58.  normal_proj (inverse transpose
 * of the view, camera.rs:130-132) is uploaded but never read by the shaders (render_shader.wgsl:5);
 * the oracle fills it with the transpose of the view's rotation part extended to 4x4, which is what
 * inverse().transpose() of a rigid transform equals up to rounding -- it does not reach any output. */
Uniforms camera_uniforms(const float eye_in[3], float yaw, float pitch, float fov_y, float width,
                         float height, float sun_theta_deg, float sun_phi_deg, int view_mode) {
    Uniforms u;
    memset(&u, 0, sizeof u);
    v3 eye = {eye_in[0], eye_in[1], eye_in[2]};
    v3 up = glam_normalize(eye);
    quat rot = quat_from_rotation_arc({0.0f, -1.0f, 0.0f}, up);
    float x = cosf(yaw) * cosf(pitch), y = sinf(pitch), z = sinf(yaw) * cosf(pitch);
    v3 dir = quat_mul_vec3(rot, {x, y, z});
    /* Mat4::look_to_rh(eye, dir, up) */
    v3 f = dir;
    v3 s = glam_normalize(cross(f, up));
    v3 uu = cross(s, f);
    float view[16] = {s.x, uu.x, -f.x, 0.0f, s.y, uu.y, -f.y, 0.0f, s.z, uu.z, -f.z, 0.0f,
                      -dot(eye, s), -dot(eye, uu), dot(eye, f), 1.0f};
    /* Mat4::perspective_rh(fov_y, aspect, near, far) */
    float aspect = width / height;
    float sf = sinf(0.5f * fov_y), cf = cosf(0.5f * fov_y);
    float hh = cf / sf, ww = hh / aspect, r = FAR_Z / (NEAR_Z - FAR_Z);
    float proj[16] = {ww, 0, 0, 0, 0, hh, 0, 0, 0, 0, r, -1.0f, 0, 0, r * NEAR_Z, 0};
    mat4_mul(proj, view, u.camera_proj);
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) u.normal_proj[c * 4 + rr] = view[c * 4 + rr];
    u.normal_proj[15] = 1.0f;
    u.camera_pos[0] = eye.x; u.camera_pos[1] = eye.y; u.camera_pos[2] = eye.z; u.camera_pos[3] = 0.0f;
    float m3[9];
    mat3_from_euler_xyz_ex(0.0f, to_radians_rs(90.0f - sun_phi_deg), to_radians_rs(sun_theta_deg), m3);
    u.sun_direction[0] = m3[6]; u.sun_direction[1] = m3[7]; u.sun_direction[2] = m3[8]; /* * Vec3::Z */
    u.view_mode = view_mode;
    return u;
}

/* ------------------------------------------------------------------------------------------
 * Vertex stage: render_shader.wgsl:35-73.
 * ------------------------------------------------------------------------------------------ */
struct VSOut {
    float clip[4];
    v3 wpos;
    v3 wnrm;
};

VSOut vs_main(const Tile& t, uint32_t px, uint32_t py, const Uniforms& u) {
    VSOut o;
    float height = t.height((int)px, (int)py);
    /* to_model */
    float mx = ((float)px - t.tu.raster_point[0]) * t.tu.pixel_scale[0] + t.tu.model_point[0];
    float my = ((float)py - t.tu.raster_point[1]) * -t.tu.pixel_scale[1] + t.tu.model_point[1];
    float longitude = radians(mx), latitude = radians(my);
    float R = R0 + height;
    float slat, clat, slon, clon;
    sincos_spec(latitude, &slat, &clat);
    sincos_spec(longitude, &slon, &clon);
    o.wpos = {R * clat * clon, R * clat * slon, R * slat};
    const uint8_t* n8 = &t.nrm[((size_t)py * t.w + px) * 4];
    v3 normal = {2.0f * from_unorm8(n8[0]) - 1.0f, 2.0f * from_unorm8(n8[1]) - 1.0f,
                 2.0f * from_unorm8(n8[2]) - 1.0f};
    o.wnrm = mat4_mul_dir(t.tu.normal_to_world_rot, normal);
    mat4_mul_point(u.camera_proj, o.wpos.x, o.wpos.y, o.wpos.z, o.clip);
    return o;
}

/* ------------------------------------------------------------------------------------------
 * Fragment stage: render_shader.wgsl:75-87 (hash/dither), :96-115 (fs_main).
 * ------------------------------------------------------------------------------------------ */
float hash12n(float sx, float sy) {
    float px = fract(sx * 5.3987f), py = fract(sy * 5.4421f);
    float d = py * (px + 21.5351f) + px * (py + 14.3137f); /* dot(p.yx, p.xy + c) */
    px += d; py += d;
    return fract(px * py * 95.4307f);
}
void fs_main(const Uniforms& u, float frag_x, float frag_y, v3 wpos, v3 wnrm, float out[4]) {
    const float ambient_strength = 0.01f;
    v3 sun = {u.sun_direction[0], u.sun_direction[1], u.sun_direction[2]};
    float d = dot(normalize(wnrm), sun);
    float diffuse_strength = 0.7f * (d > 0.0f ? d : 0.0f);
    float result_lin = ambient_strength + diffuse_strength; /* light_color = in.color = 1 */
    if (u.view_mode == 2) {
        out[0] = 0.5f * (wnrm.x + 1.0f); out[1] = 0.5f * (wnrm.y + 1.0f); out[2] = 0.5f * (wnrm.z + 1.0f);
        out[3] = 1.0f;
    } else if (u.view_mode == 1) {
        out[0] = out[1] = out[2] = result_lin; out[3] = 1.0f;
    } else {
        /* ditherRGB(result_lin, frag.xy + camera_pos.xy - world_pos.xy) */
        float px = frag_x + u.camera_pos[0] - wpos.x, py = frag_y + u.camera_pos[1] - wpos.y;
        float h1[3] = {hash12n(px, py), hash12n(px + 0.07f, py + 0.07f), hash12n(px + 0.11f, py + 0.11f)};
        float qx = px + 0.13f, qy = py + 0.13f;
        float h2[3] = {hash12n(qx, qy), hash12n(qx + 0.07f, qy + 0.07f), hash12n(qx + 0.11f, qy + 0.11f)};
        for (int k = 0; k < 3; ++k) out[k] = result_lin + 1.0f * (h1[k] + h2[k] - 1.0f) / 255.0f;
        out[3] = 1.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * One frame (terrain_renderer.rs:365-452): clear, draws, post.
 * ------------------------------------------------------------------------------------------ */
struct Frame {
    uint32_t W, H;
    std::vector<float> depth;     /* Depth32Float, clear 1.0 (terrain_renderer.rs:392) */
    std::vector<uint8_t> color;   /* render_texture, Rgba8UnormSrgb */
    std::vector<uint8_t> final_;  /* surface, Rgba8UnormSrgb */
    /* bookkeeping for oracle/ray_check.py only (not part of any reference resource): the draw that owns each pixel,
     * (tile rank in draw order) * 2(w-1)(h-1) + index-buffer triangle; 0xFFFFFFFF = cleared */
    std::vector<uint32_t> winner;
    uint32_t cur_draw = 0;
    /* the surface format RenderEngine::new picks (render_engine.rs:77-84: formats[0], with the sRGB suffix when the
     * surface offers it) -- render target and final target share it (terrain_renderer.rs:88-93):
     * 1 Rgba8UnormSrgb, 2 Bgra8UnormSrgb, 3 Rgba8Unorm, 4 Bgra8Unorm.  The Srgb formats encode on store and decode on
     * sample; the plain ones store round(clamp(v) * 255) and sample c / 255.  Bgra only changes the byte order in memory. */
    uint32_t format = 1;
};
inline bool format_is_srgb(uint32_t f) { return f == 1 || f == 2; }
inline bool format_is_bgra(uint32_t f) { return f == 2 || f == 4; }

void store_color(Frame& f, size_t p, const float c[4]) {
    uint8_t* o = &f.color[p * 4];
    if (format_is_srgb(f.format)) { o[0] = srgb().encode(c[0]); o[1] = srgb().encode(c[1]); o[2] = srgb().encode(c[2]); }
    else { o[0] = unorm8(c[0]); o[1] = unorm8(c[1]); o[2] = unorm8(c[2]); }
    o[3] = unorm8(c[3]);
}

struct ScreenVert {
    int64_t X, Y; /* 1/256-pixel fixed point */
    float z, w;   /* z_ndc, w_clip */
};

/* Raster spec: perspective divide by reciprocal, viewport transform as WebGPU framebufferCoords, snap to 1/256 pixel
 * (round-half-even), guard band |coord| <= 2^20 pixels else the triangle is discarded. */
bool to_screen(const VSOut& v, uint32_t W, uint32_t H, ScreenVert* s) {
    float w = v.clip[3];
    float rw = 1.0f / w;                                   /* perspective divide: reciprocal, then multiplies */
    float nx = v.clip[0] * rw, ny = v.clip[1] * rw, nz = v.clip[2] * rw;
    float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
    float xf = fmaf(nx, half_w, half_w);                   /* 0.5 * (ndc.x + 1) * W */
    float yf = fmaf(-ny, half_h, half_h);                  /* 0.5 * (1 - ndc.y) * H */
    if (!(fabsf(xf) <= 1048576.0f) || !(fabsf(yf) <= 1048576.0f)) return false;
    s->X = (int64_t)rintf(xf * 256.0f);
    s->Y = (int64_t)rintf(yf * 256.0f);
    s->z = nz;
    s->w = w;
    return true;
}

static inline int64_t floor_div(int64_t a, int64_t b) { /* b > 0 */
    int64_t q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

/* Fixed-function state: CCW front, back-cull, Depth32F Less + write, REPLACE (pipeline.rs:221-246). */
/* `orig` (non-null for the pieces of a primitive cut by the near plane): the primitive's own three vertices.  Coverage
 * and depth of a piece come from the piece; its varyings are evaluated on the ORIGINAL primitive with homogeneous
 * (clip-space) barycentrics, as clippers hand them to the attribute interpolators -- DESIGN.md "Raster spec" 8. */
void raster_triangle(Frame& f, const Uniforms& u, const VSOut& v0, const VSOut& v1, const VSOut& v2,
                     const ScreenVert& s0, const ScreenVert& s1, const ScreenVert& s2, const VSOut* const* orig = nullptr) {
    /* signed doubled area in y-down framebuffer space; visually counter-clockwise <=> negative */
    int64_t area2 = (s1.X - s0.X) * (s2.Y - s0.Y) - (s1.Y - s0.Y) * (s2.X - s0.X);
    if (area2 >= 0) return; /* back-facing or degenerate */
    const int64_t A = -area2;
    int64_t minX = std::min(s0.X, std::min(s1.X, s2.X)), maxX = std::max(s0.X, std::max(s1.X, s2.X));
    int64_t minY = std::min(s0.Y, std::min(s1.Y, s2.Y)), maxY = std::max(s0.Y, std::max(s1.Y, s2.Y));
    /* pixel centres at (px*256+128, py*256+128) */
    int64_t x0 = std::max<int64_t>(0, floor_div(minX - 128 + 255, 256));
    int64_t x1 = std::min<int64_t>((int64_t)f.W - 1, floor_div(maxX - 128, 256));
    int64_t y0 = std::max<int64_t>(0, floor_div(minY - 128 + 255, 256));
    int64_t y1 = std::min<int64_t>((int64_t)f.H - 1, floor_div(maxY - 128, 256));
    if (x0 > x1 || y0 > y1) return;
    /* F_ab(p) = (by-ay)(px-ax) - (bx-ax)(py-ay): >= 0 inside for this winding.
     * Top-left rule: edge a->b owns its boundary iff it is a left edge (dy > 0) or a top edge
     * (dy == 0 && dx < 0). */
    struct Edge { int64_t ax, ay, dx, dy; int64_t bias; };
    auto mk = [](const ScreenVert& a, const ScreenVert& b) {
        Edge e{a.X, a.Y, b.X - a.X, b.Y - a.Y, 0};
        bool top_left = (e.dy > 0) || (e.dy == 0 && e.dx < 0);
        e.bias = top_left ? 0 : -1;
        return e;
    };
    const Edge e12 = mk(s1, s2), e20 = mk(s2, s0), e01 = mk(s0, s1);
    auto ev = [](const Edge& e, int64_t px, int64_t py) { return e.dy * (px - e.ax) - e.dx * (py - e.ay); };
    const float rA = 1.0f / (float)A;
    const float dz1 = s1.z - s0.z, dz2 = s2.z - s0.z;
    for (int64_t py = y0; py <= y1; ++py)
        for (int64_t px = x0; px <= x1; ++px) {
            const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
            const int64_t F0 = ev(e12, cx, cy), F1 = ev(e20, cx, cy), F2 = ev(e01, cx, cy);
            if (F0 + e12.bias < 0 || F1 + e20.bias < 0 || F2 + e01.bias < 0) continue;
            const float b0 = (float)F0 * rA, b1 = (float)F1 * rA, b2 = (float)F2 * rA;
            float z = fmaf(b1, dz1, fmaf(b2, dz2, s0.z));
            if (!(z < 1.0f)) continue; /* far plane (and NaN) */
            if (z < 0.0f) z = 0.0f;
            const size_t p = (size_t)py * f.W + (size_t)px;
            if (!(z < f.depth[p])) continue; /* CompareFunction::Less */
            /* perspective-correct varyings: weights q_i, a = (a0 q0 + a1 q1 + a2 q2) / (q0 + q1 + q2) */
            const VSOut *a0 = &v0, *a1 = &v1, *a2 = &v2;
            float q0, q1, q2;
            if (!orig) {
                q0 = b0 * (1.0f / s0.w); q1 = b1 * (1.0f / s1.w); q2 = b2 * (1.0f / s2.w);
            } else {
                /* the pixel centre as a clip-space direction (gx, gy, 1); q = adj([x y w columns]) . (gx, gy, 1) */
                a0 = orig[0]; a1 = orig[1]; a2 = orig[2];
                const float gx = fmaf((float)px + 0.5f, 2.0f / (float)f.W, -1.0f);
                const float gy = fmaf(-((float)py + 0.5f), 2.0f / (float)f.H, 1.0f);
                auto det = [](float a, float b, float c, float d) { const float p = a * b, q = c * d; return p - q; };
                const float *c0 = a0->clip, *c1 = a1->clip, *c2 = a2->clip;
                q0 = fmaf(det(c1[1], c2[3], c1[3], c2[1]), gx, fmaf(det(c1[3], c2[0], c1[0], c2[3]), gy, det(c1[0], c2[1], c1[1], c2[0])));
                q1 = fmaf(det(c2[1], c0[3], c2[3], c0[1]), gx, fmaf(det(c2[3], c0[0], c2[0], c0[3]), gy, det(c2[0], c0[1], c2[1], c0[0])));
                q2 = fmaf(det(c0[1], c1[3], c0[3], c1[1]), gx, fmaf(det(c0[3], c1[0], c0[0], c1[3]), gy, det(c0[0], c1[1], c0[1], c1[0])));
            }
            const float rq = 1.0f / ((q0 + q1) + q2);
            v3 wpos, wnrm;
            wpos.x = fmaf(a2->wpos.x, q2, fmaf(a1->wpos.x, q1, a0->wpos.x * q0)) * rq;
            wpos.y = fmaf(a2->wpos.y, q2, fmaf(a1->wpos.y, q1, a0->wpos.y * q0)) * rq;
            wpos.z = fmaf(a2->wpos.z, q2, fmaf(a1->wpos.z, q1, a0->wpos.z * q0)) * rq;
            wnrm.x = fmaf(a2->wnrm.x, q2, fmaf(a1->wnrm.x, q1, a0->wnrm.x * q0)) * rq;
            wnrm.y = fmaf(a2->wnrm.y, q2, fmaf(a1->wnrm.y, q1, a0->wnrm.y * q0)) * rq;
            wnrm.z = fmaf(a2->wnrm.z, q2, fmaf(a1->wnrm.z, q1, a0->wnrm.z * q0)) * rq;
            float c[4];
            fs_main(u, (float)px + 0.5f, (float)py + 0.5f, wpos, wnrm, c);
            f.depth[p] = z;
            if (!f.winner.empty()) f.winner[p] = f.cur_draw;
            store_color(f, p, c);
        }
}

VSOut lerp_vs(const VSOut& in, const VSOut& out, float t) {
    VSOut r;
    for (int k = 0; k < 4; ++k) r.clip[k] = in.clip[k] + t * (out.clip[k] - in.clip[k]);
    r.clip[2] = 0.0f; /* on the near plane by construction */
    r.wpos = add(in.wpos, scale(sub(out.wpos, in.wpos), t));
    r.wnrm = add(in.wnrm, scale(sub(out.wnrm, in.wnrm), t));
    return r;
}

/* Primitive clipping against the near plane z_clip >= 0 (WebGPU clip volume 0 <= z <= w; with
 * perspective_rh z_clip >= 0 implies w >= NEAR > 0).  Sutherland-Hodgman; the intersection is always
 * computed from the inside vertex towards the outside vertex, t = z_in / (z_in - z_out), so the two
 * triangles sharing an edge produce the same point.  Fan-triangulated from the first emitted vertex.
 * The far plane is applied per fragment (z < 1), x/y planes by the viewport scissor. */
struct PV { /* post-transform vertex cache entry */
    VSOut v;
    ScreenVert s;
    bool in;  /* z_clip >= 0 */
    bool ok;  /* inside the guard band (meaningful only when in) */
};
PV make_pv(const VSOut& v, uint32_t W, uint32_t H) {
    PV p;
    p.v = v;
    p.in = v.clip[2] >= 0.0f;
    p.ok = p.in && to_screen(v, W, H, &p.s);
    return p;
}
void draw_triangle(Frame& f, const Uniforms& u, const PV& pa, const PV& pb, const PV& pc) {
    const VSOut &a = pa.v, &b = pb.v, &c = pc.v;
    const VSOut* v[3] = {&a, &b, &c};
    bool in[3] = {pa.in, pb.in, pc.in};
    int nin = (int)in[0] + (int)in[1] + (int)in[2];
    if (nin == 0) return; /* also rejects NaN z */
    if (nin == 3) {
        if (pa.ok && pb.ok && pc.ok) raster_triangle(f, u, a, b, c, pa.s, pb.s, pc.s);
        return;
    }
    VSOut poly[4];
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        if (in[i]) poly[n++] = *v[i];
        if (in[i] != in[j]) {
            const VSOut& I = in[i] ? *v[i] : *v[j];
            const VSOut& O = in[i] ? *v[j] : *v[i];
            float t = I.clip[2] / (I.clip[2] - O.clip[2]);
            poly[n++] = lerp_vs(I, O, t);
        }
    }
    ScreenVert sp[4];
    for (int k = 0; k < n; ++k)
        if (!to_screen(poly[k], f.W, f.H, &sp[k])) return; /* guard band: whole primitive discarded */
    for (int k = 1; k + 1 < n; ++k) raster_triangle(f, u, poly[0], poly[k], poly[k + 1], sp[0], sp[k], sp[k + 1], v);
}

/* generate_indices: render_buffer.rs:191-219 (vertex index = i*h + j <-> texel (x=i, y=j), :185-189) */
void draw_tile(Frame& f, const Uniforms& u, const Tile& t, uint32_t draw_base = 0) {
    /* post-transform vertex cache (vs_main is a pure function of the vertex) */
    std::vector<PV> vs((size_t)t.w * t.h);
    for (uint32_t i = 0; i < t.w; ++i)
        for (uint32_t j = 0; j < t.h; ++j) vs[(size_t)i * t.h + j] = make_pv(vs_main(t, i, j, u), f.W, f.H);
    for (uint32_t i = 0; i + 1 < t.w; ++i)
        for (uint32_t j = 0; j + 1 < t.h; ++j) {
            size_t index = (size_t)i * t.h + j, next = (size_t)(i + 1) * t.h + j;
            f.cur_draw = draw_base + (uint32_t)(((size_t)i * (t.h - 1) + j) * 2);
            if ((i + j) % 2 == 0) {
                draw_triangle(f, u, vs[index], vs[index + 1], vs[next + 1]);
                ++f.cur_draw;
                draw_triangle(f, u, vs[next + 1], vs[next], vs[index]);
            } else {
                draw_triangle(f, u, vs[index], vs[index + 1], vs[next]);
                ++f.cur_draw;
                draw_triangle(f, u, vs[next + 1], vs[next], vs[index + 1]);
            }
        }
}

/* The render target as the post pass samples it (s_render: mag Linear, min Nearest, clamp-to-edge, one mip level:
 * texture.rs:69-78), restated for the pixelise branch (postprocessing_shader.wgsl:70-74), the one place where uv does not
 * address a texel centre.  What WebGPU leaves to the implementation is fixed as follows (DESIGN.md "Raster spec", item 10):
 * derivatives are the differences across the pixel's 2 x 2 quad (quads start at even coordinates; a partner pixel outside the
 * target still has its uv, as a helper invocation has); rho = max(|du/dx| * tw, |dv/dy| * th) (the cross terms are zero: u
 * depends on x only, v on y only); rho > 1 minifies -- Nearest: texel (floor(u tw), floor(v th)) --, otherwise the sample
 * magnifies -- Linear: texel coordinates (u tw - 0.5, v th - 0.5), the four texels around them clamped to the edge, decoded,
 * and weighted with the f32 fractions, mix(mix(c00, c10, fx), mix(c01, c11, fx), fy) with WGSL's mix(a, b, t) = a (1 - t) + b t. */
static void decode_texel(const Frame& f, int x, int y, float out[4]) {
    const int W = (int)f.W, H = (int)f.H;
    x = std::min(std::max(x, 0), W - 1);
    y = std::min(std::max(y, 0), H - 1);
    const uint8_t* c8 = &f.color[((size_t)y * W + x) * 4];
    const bool is_srgb = format_is_srgb(f.format);
    for (int k = 0; k < 3; ++k) out[k] = is_srgb ? srgb().decode[c8[k]] : from_unorm8(c8[k]);
    out[3] = from_unorm8(c8[3]);
}
static inline float pixelized(float frag, float viewport, float n) { return floorf(frag / viewport * n) / n; }
static void sample_pixelized(const Frame& f, const PostUniforms& pu, int px, int py, float out[4]) {
    const float n = pu.pixelize_n, tw = (float)f.W, th = (float)f.H;
    const float u = pixelized((float)px + 0.5f, pu.viewport[0], n), v = pixelized((float)py + 0.5f, pu.viewport[1], n);
    const float ux = pixelized((float)(px ^ 1) + 0.5f, pu.viewport[0], n), vy = pixelized((float)(py ^ 1) + 0.5f, pu.viewport[1], n);
    const float rho = std::max(fabsf(ux - u) * tw, fabsf(vy - v) * th);
    if (rho > 1.0f) {
        decode_texel(f, (int)floorf(u * tw), (int)floorf(v * th), out);
        return;
    }
    const float tx = u * tw - 0.5f, ty = v * th - 0.5f;
    const float x0 = floorf(tx), y0 = floorf(ty), fx = tx - x0, fy = ty - y0;
    float c00[4], c10[4], c01[4], c11[4];
    decode_texel(f, (int)x0, (int)y0, c00);
    decode_texel(f, (int)x0 + 1, (int)y0, c10);
    decode_texel(f, (int)x0, (int)y0 + 1, c01);
    decode_texel(f, (int)x0 + 1, (int)y0 + 1, c11);
    for (int k = 0; k < 4; ++k) out[k] = mix(mix(c00[k], c10[k], fx), mix(c01[k], c11[k], fx), fy);
}

/* Post pass: postprocessing_shader.wgsl:56-96.  uv = frag.xy / viewport addresses exact texel
 * centres (viewport == target size), so every tap is an exact texel fetch; depth taps clamp to the
 * edge (default sampler: texture.rs:113-117).  pixelize_n < 99.99999 (never in the reference, which always passes 100:
 * application_data.rs:31) moves the COLOUR sample to floor(uv n) / n: sample_pixelized() above. */
void post_pass(Frame& f, const PostUniforms& pu) {
    const int W = (int)f.W, H = (int)f.H;
    const bool pixelize = pu.pixelize_n < 99.99999f;
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            const size_t p = (size_t)py * W + px;
            const uint8_t* c8 = &f.color[p * 4];
            const bool is_srgb = format_is_srgb(f.format);
            float rc[4] = {is_srgb ? srgb().decode[c8[0]] : from_unorm8(c8[0]), is_srgb ? srgb().decode[c8[1]] : from_unorm8(c8[1]),
                           is_srgb ? srgb().decode[c8[2]] : from_unorm8(c8[2]), from_unorm8(c8[3])};
            if (pixelize) sample_pixelized(f, pu, px, py, rc);
            float center_linear = dist_from_depth(f.depth[p]);
            float contour = 8.0f * center_linear;
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    if (i == 0 && j == 0) continue;
                    int sx = std::min(std::max(px + i, 0), W - 1), sy = std::min(std::max(py + j, 0), H - 1);
                    contour -= dist_from_depth(f.depth[(size_t)sy * W + sx]);
                }
            float a = smoothstep(0.05f, 0.15f, contour / center_linear);
            const float cc[4] = {0.0f, 0.0f, 0.0f, 1.0f};
            uint8_t* o = &f.final_[p * 4];
            const float m0 = mix(rc[0], cc[0], a), m1 = mix(rc[1], cc[1], a), m2 = mix(rc[2], cc[2], a);
            o[0] = is_srgb ? srgb().encode(m0) : unorm8(m0);
            o[1] = is_srgb ? srgb().encode(m1) : unorm8(m1);
            o[2] = is_srgb ? srgb().encode(m2) : unorm8(m2);
            if (format_is_bgra(f.format)) std::swap(o[0], o[2]);      /* memory order of the surface texel: B G R A */
            o[3] = unorm8(mix(rc[3], cc[3], a));
        }
}

struct Oracle {
    uint32_t W = 0, H = 0;
    uint32_t format = 1;
    std::map<GeoKey, std::unique_ptr<Tile>> tiles;
    Uniforms u{};
    PostUniforms pu{};
    Frame frame;
    std::string err;
    Tile* find(int lat, int lon) {
        auto it = tiles.find(geo_key(lat, lon));
        return it == tiles.end() ? nullptr : it->second.get();
    }
};

void render_frame(const Oracle& o, const Uniforms& u, Frame& f) {
    f.W = o.W; f.H = o.H; f.format = o.format;
    const size_t P = (size_t)o.W * o.H;
    f.depth.assign(P, 1.0f);
    f.color.resize(P * 4);
    f.final_.resize(P * 4);
    const float clear[4] = {(float)0.0, (float)0.71, (float)0.885, (float)1.0}; /* terrain_renderer.rs:379-384 */
    store_color(f, 0, clear);      /* encode the clear colour once, then replicate the texel */
    for (size_t p = 1; p < P; ++p) memcpy(&f.color[p * 4], &f.color[0], 4);
    uint32_t rank = 0;
    for (const auto& kv : o.tiles) {   /* BTreeMap order, :407-420 */
        draw_tile(f, u, *kv.second, rank * 2u * (kv.second->w - 1) * (kv.second->h - 1));
        ++rank;
    }
    post_pass(f, o.pu);
}

void copy_out(const Frame& f, uint8_t* rgba, size_t rgba_pitch, float* depth, size_t depth_pitch,
              uint8_t* pre_post) {
    for (uint32_t y = 0; y < f.H; ++y) {
        if (rgba) memcpy(rgba + (size_t)y * rgba_pitch, &f.final_[(size_t)y * f.W * 4], (size_t)f.W * 4);
        if (depth)
            memcpy((uint8_t*)depth + (size_t)y * depth_pitch, &f.depth[(size_t)y * f.W], (size_t)f.W * 4);
        if (pre_post) memcpy(pre_post + (size_t)y * f.W * 4, &f.color[(size_t)y * f.W * 4], (size_t)f.W * 4);
    }
}

}  // namespace

/* ==========================================================================================
 * C interface for ctypes (tests / smoke / cpu_baseline only).
 * ========================================================================================== */
extern "C" {

void* oracle_create(uint32_t w, uint32_t h) {
    Oracle* o = new Oracle();
    o->W = w; o->H = h;
    return o;
}
void oracle_destroy(void* p) { delete (Oracle*)p; }
const char* oracle_last_error(void* p) { return ((Oracle*)p)->err.c_str(); }

/* TerrainRenderer::add_terrain: terrain_renderer.rs:173-350 */
int oracle_add_terrain(void* p, int32_t lat, int32_t lon, const float* heights, uint32_t w, uint32_t h,
                       const float raster_point[2], const float model_point[2], const float pixel_scale[2]) {
    Oracle& o = *(Oracle*)p;
    if (w < 3 || h < 3) { o.err = "tile must be at least 3x3"; return -1; }
    if (!o.tiles.empty()) {
        const Tile& first = *o.tiles.begin()->second;
        if (first.w != w || first.h != h) { o.err = "mixed tile sizes are rejected (render_buffer.rs:12-15)"; return -1; }
    }
    auto t = std::make_unique<Tile>();
    t->lat = lat; t->lon = lon; t->w = w; t->h = h;
    t->hgt.assign(heights, heights + (size_t)w * h);
    t->nrm.assign((size_t)w * h * 4, 0);
    t->tu = terrain_uniforms_new(raster_point, model_point, pixel_scale, w, h);
    Tile& nt = *t;
    normals_interior(nt);
    Tile* left = o.find(lat, lon - 1);
    Tile* right = o.find(lat, lon + 1);
    Tile* top = o.find(lat + 1, lon);
    Tile* bottom = o.find(lat - 1, lon);
    Tile* top_left = o.find(lat + 1, lon - 1);
    Tile* top_right = o.find(lat + 1, lon + 1);
    Tile* bottom_left = o.find(lat - 1, lon - 1);
    Tile* bottom_right = o.find(lat - 1, lon + 1);
    if (left) normals_edge_lr(*left, nt, nt.tu);
    if (right) normals_edge_lr(nt, *right, nt.tu);
    if (top) normals_edge_tb(*top, nt, nt.tu);
    if (bottom) normals_edge_tb(nt, *bottom, nt.tu);
    if (top_left && top && left) normals_corner(*top_left, *top, *left, nt, nt.tu);
    if (top && top_right && right) normals_corner(*top, *top_right, nt, *right, nt.tu);
    if (left && bottom_left && bottom) normals_corner(*left, nt, *bottom_left, *bottom, nt.tu);
    if (right && bottom && bottom_right) normals_corner(nt, *right, *bottom, *bottom_right, nt.tu);
    o.tiles[geo_key(lat, lon)] = std::move(t); /* BTreeMap::insert replaces */
    return 0;
}

/* unload_terrain: terrain_renderer.rs:361-363 (neighbours' seam normals are left as they are) */
int oracle_unload_terrain(void* p, int32_t lat, int32_t lon) {
    ((Oracle*)p)->tiles.erase(geo_key(lat, lon));
    return 0;
}

/* update: terrain_renderer.rs:151-171 */
int oracle_update(void* p, uint32_t w, uint32_t h, const void* uniforms160, const void* post16) {
    Oracle& o = *(Oracle*)p;
    PostUniforms pu;
    memcpy(&pu, post16, sizeof pu);
    if (pu.pixelize_n < 99.99999f && !(pu.pixelize_n >= 1.0f)) { o.err = "pixelize_n must be at least 1"; return -2; }
    o.W = w; o.H = h;
    memcpy(&o.u, uniforms160, sizeof o.u);
    o.pu = pu;
    return 0;
}

/* render + depth copy: terrain_renderer.rs:365-452, render_engine.rs:219-249 */
int oracle_render(void* p, uint8_t* rgba, size_t rgba_pitch, float* depth, size_t depth_pitch, uint8_t* pre_post) {
    Oracle& o = *(Oracle*)p;
    render_frame(o, o.u, o.frame);
    copy_out(o.frame, rgba, rgba_pitch, depth, depth_pitch, pre_post);
    return 0;
}

/* n independent frames over the same tile set (one per panorama sector), OpenMP over frames.
 * Output v lands at base + v*view_stride; used for the timed CPU baseline. */
int oracle_render_views(void* p, uint32_t n, const void* uniforms160xn, uint8_t* rgba, size_t rgba_view_stride,
                        size_t rgba_pitch, float* depth, size_t depth_view_stride, size_t depth_pitch, int threads) {
    Oracle& o = *(Oracle*)p;
    const Uniforms* us = (const Uniforms*)uniforms160xn;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int v = 0; v < (int)n; ++v) {
        Frame f;
        Uniforms u;
        memcpy(&u, &us[v], sizeof u);
        render_frame(o, u, f);
        copy_out(f, rgba ? rgba + (size_t)v * rgba_view_stride : nullptr, rgba_pitch,
                 depth ? (float*)((uint8_t*)depth + (size_t)v * depth_view_stride) : nullptr, depth_pitch, nullptr);
    }
    return 0;
}

/* TerrainRenderer::new's `format` argument (terrain_renderer.rs:37; chosen at render_engine.rs:77-84): 1 Rgba8UnormSrgb,
 * 2 Bgra8UnormSrgb, 3 Rgba8Unorm, 4 Bgra8Unorm */
int oracle_set_format(void* p, uint32_t format) {
    if (format < 1 || format > 4) return -1;
    ((Oracle*)p)->format = format;
    return 0;
}

/* As oracle_render, additionally reporting which draw owns each pixel (ray_check.py compares it with a ray cast). */
int oracle_render_winners(void* p, float* depth, uint32_t* winner) {
    Oracle& o = *(Oracle*)p;
    o.frame.winner.assign((size_t)o.W * o.H, 0xFFFFFFFFu);
    render_frame(o, o.u, o.frame);
    memcpy(depth, o.frame.depth.data(), (size_t)o.W * o.H * 4);
    memcpy(winner, o.frame.winner.data(), (size_t)o.W * o.H * 4);
    o.frame.winner.clear();
    return 0;
}

/* The same n frames with MORE threads than frames (bench.py's cpu_baseline on a many-core host): each frame's tiles are
 * split into `groups` contiguous runs of the draw order, every (frame, run) job rasterises its tiles into a z-buffer of its
 * own, and the runs are merged per pixel -- smallest depth wins, on equal depth the earlier run, which is what `Less` with
 * in-order draws gives -- before the post pass.  Same bytes as oracle_render_views (tests/test_oracle_kat.py). */
int oracle_render_views_tiled(void* p, uint32_t n, const void* uniforms160xn, uint8_t* rgba, size_t rgba_view_stride,
                              size_t rgba_pitch, float* depth, size_t depth_view_stride, size_t depth_pitch, int threads, int groups) {
    Oracle& o = *(Oracle*)p;
    const Uniforms* us = (const Uniforms*)uniforms160xn;
    if (groups < 1) groups = 1;
    const size_t P = (size_t)o.W * o.H;
    std::vector<const Tile*> order;
    for (const auto& kv : o.tiles) order.push_back(kv.second.get());      /* BTreeMap order */
    const int T = (int)order.size();
    std::vector<Frame> part((size_t)n * groups);
    const float clear[4] = {(float)0.0, (float)0.71, (float)0.885, (float)1.0};
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int job = 0; job < (int)n * groups; ++job) {
        const int v = job / groups, g = job % groups;
        Frame& f = part[job];
        f.W = o.W; f.H = o.H; f.format = o.format;
        f.depth.assign(P, 1.0f);
        f.color.resize(P * 4);
        store_color(f, 0, clear);
        for (size_t px = 1; px < P; ++px) memcpy(&f.color[px * 4], &f.color[0], 4);
        Uniforms u;
        memcpy(&u, &us[v], sizeof u);
        const int lo = (int)((long long)T * g / groups), hi = (int)((long long)T * (g + 1) / groups);
        for (int k = lo; k < hi; ++k) draw_tile(f, u, *order[k]);
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int v = 0; v < (int)n; ++v) {
        Frame& f = part[(size_t)v * groups];
        for (int g = 1; g < groups; ++g) {
            Frame& q = part[(size_t)v * groups + g];
            for (size_t px = 0; px < P; ++px)
                if (q.depth[px] < f.depth[px]) { f.depth[px] = q.depth[px]; memcpy(&f.color[px * 4], &q.color[px * 4], 4); }
            std::vector<float>().swap(q.depth);
            std::vector<uint8_t>().swap(q.color);
        }
        f.final_.resize(P * 4);
        post_pass(f, o.pu);
        copy_out(f, rgba ? rgba + (size_t)v * rgba_view_stride : nullptr, rgba_pitch,
                 depth ? (float*)((uint8_t*)depth + (size_t)v * depth_view_stride) : nullptr, depth_pitch, nullptr);
    }
    return 0;
}

/* ---- overlay pass: LineRenderer::render (line_renderer.rs:200-212) with line_shader.wgsl, drawn into the post pass.
 * Pipeline state: TriangleList, CCW front, back faces culled, no blending (line_renderer.rs:262-276); depth test Greater
 * with write (pipeline.rs:24-32) against the attachment the post quad filled with 1/4096 (postprocessing_shader.wgsl:62).
 * A classic in-order z-buffer: the triangles one after the other, fragment by fragment. */
struct LineVertex { float position[2], normal[2], color[3]; int32_t z_index; };      /* GpuVertex, line_renderer.rs:18-25 */

int oracle_overlay_lines(void* p, const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, float width,
                         uint8_t* rgba, size_t pitch) {
    Oracle& o = *(Oracle*)p;
    const LineVertex* vs = (const LineVertex*)vertices;
    const uint32_t W = o.W, H = o.H;
    std::vector<float> zbuf((size_t)W * H, 1.0f / 4096.0f);
    for (uint32_t t = 0; t + 2 < n_indices; t += 3) {
        ScreenVert sv[3];
        const LineVertex* lv[3];
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            if (indices[t + k] >= n_vertices) { ok = false; break; }
            lv[k] = &vs[indices[t + k]];
            /* vs_main, line_shader.wgsl:27-41 */
            VSOut v{};
            const float z = (float)lv[k]->z_index / 4096.0f;
            const float px = lv[k]->position[0] + lv[k]->normal[0] * width, py = -(lv[k]->position[1] + lv[k]->normal[1] * width);
            v.clip[0] = 2.0f * px / (float)W - 1.0f;
            v.clip[1] = 2.0f * py / (float)H + 1.0f;
            v.clip[2] = z;
            v.clip[3] = 1.0f;
            ok = to_screen(v, W, H, &sv[k]);
        }
        if (!ok) continue;
        const int64_t area2 = (sv[1].X - sv[0].X) * (sv[2].Y - sv[0].Y) - (sv[1].Y - sv[0].Y) * (sv[2].X - sv[0].X);
        if (area2 >= 0) continue;      /* back-facing or degenerate */
        const int64_t minX = std::min(sv[0].X, std::min(sv[1].X, sv[2].X)), maxX = std::max(sv[0].X, std::max(sv[1].X, sv[2].X));
        const int64_t minY = std::min(sv[0].Y, std::min(sv[1].Y, sv[2].Y)), maxY = std::max(sv[0].Y, std::max(sv[1].Y, sv[2].Y));
        const int64_t x0 = std::max<int64_t>(0, floor_div(minX - 128 + 255, 256)), x1 = std::min<int64_t>((int64_t)W - 1, floor_div(maxX - 128, 256));
        const int64_t y0 = std::max<int64_t>(0, floor_div(minY - 128 + 255, 256)), y1 = std::min<int64_t>((int64_t)H - 1, floor_div(maxY - 128, 256));
        const float iA = 1.0f / (float)(-area2);
        for (int64_t py = y0; py <= y1; ++py)
            for (int64_t px = x0; px <= x1; ++px) {
                const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
                int64_t F[3];
                bool in = true;
                for (int e = 0; e < 3; ++e) {      /* edge e: v[e+1] -> v[e+2], the weight of v[e]; top-left rule */
                    const ScreenVert &a = sv[(e + 1) % 3], &b = sv[(e + 2) % 3];
                    const int64_t dx = b.X - a.X, dy = b.Y - a.Y;
                    F[e] = dy * (cx - a.X) - dx * (cy - a.Y);
                    const bool owns = dy > 0 || (dy == 0 && dx < 0);
                    if (F[e] < 0 || (F[e] == 0 && !owns)) in = false;
                }
                if (!in) continue;
                const float b0 = (float)F[0] * iA, b1 = (float)F[1] * iA, b2 = (float)F[2] * iA;
                const float z = fmaf(b1, sv[1].z - sv[0].z, fmaf(b2, sv[2].z - sv[0].z, sv[0].z));
                if (!(z >= 0.0f && z <= 1.0f)) continue;
                const size_t pix = (size_t)py * W + (size_t)px;
                if (!(z > zbuf[pix])) continue;      /* CompareFunction::Greater */
                zbuf[pix] = z;
                const float iq = 1.0f / ((b0 + b1) + b2);      /* w = 1 for every vertex */
                float c[3];
                for (int k = 0; k < 3; ++k) c[k] = fmaf(lv[2]->color[k], b2, fmaf(lv[1]->color[k], b1, lv[0]->color[k] * b0)) * iq;
                uint8_t* out = rgba + (size_t)py * pitch + (size_t)px * 4;
                const bool is_srgb = format_is_srgb(o.format);
                uint8_t r8 = is_srgb ? srgb().encode(c[0]) : unorm8(c[0]), g8 = is_srgb ? srgb().encode(c[1]) : unorm8(c[1]),
                        b8 = is_srgb ? srgb().encode(c[2]) : unorm8(c[2]);
                if (format_is_bgra(o.format)) std::swap(r8, b8);
                out[0] = r8; out[1] = g8; out[2] = b8; out[3] = unorm8(1.0f);
            }
    }
    return 0;
}

/* ---- overlay pass, text: TextRenderer::render (text_renderer.rs:198-204, :259-291) draws glyphon 0.10.0's glyph quads into
 * the same post pass after the lines (render_engine.rs:215-216).  glyphon is a third-party crate absent from the tree
 * (Cargo.lock: glyphon 0.10.0); what is restated here is its published pipeline (src/text_render.rs, src/shader.wgsl):
 *   - one instance per glyph, `GlyphToRender` (#[repr(C)], 28 bytes): pos [i32; 2] (top-left pixel), dim [u16; 2], uv [u16; 2]
 *     (top-left atlas texel), color u32 (a << 24 | r << 16 | g << 8 | b), content_type_with_srgb [u16; 2], depth f32;
 *   - vs_main: the quad pos .. pos + dim in pixels, uv .. uv + dim in atlas texels (1 : 1), z = depth; the colour's r, g, b go
 *     through the sRGB decode when content_type_with_srgb[1] == 1 (ColorMode::Accurate on an *Srgb surface), a = a8 / 255;
 *   - fs_main, mask content (content type 1; colour glyphs -- emoji -- are not restated): vec4(color.rgb, color.a * mask)
 *     with mask = the R8Unorm atlas texel, nearest;
 *   - blending BlendState::ALPHA_BLENDING: rgb = src.rgb * src.a + dst.rgb * (1 - src.a), a = src.a + dst.a * (1 - src.a),
 *     in linear light on an *Srgb surface (decode -> blend -> encode);
 *   - depth: the pass's depth-stencil state, CompareFunction::Greater with write (pipeline.rs:24-32); the reference gives
 *     every glyph the depth 100 / 4096 (prepare_with_depth(.., |_| 100.0 / 4096.0), text_renderer.rs:291) -- above the post
 *     quad's 1/4096 and the lines' 2/4096 and 3/4096 -- so of two overlapping quads the one drawn FIRST keeps its pixels,
 *     transparent ones included (the depth is written wherever the quad covers, whatever the mask says).
 * `depth` is that one value for the whole call. */
struct GlyphInst { int32_t pos[2]; uint16_t dim[2], uv[2]; uint32_t color; uint16_t content_type_with_srgb[2]; float depth; };
static_assert(sizeof(GlyphInst) == 28, "GlyphToRender is 28 bytes");

int oracle_overlay_glyphs(void* p, const void* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t aw, uint32_t ah,
                          uint8_t* rgba, size_t pitch) {
    Oracle& o = *(Oracle*)p;
    const GlyphInst* gs = (const GlyphInst*)glyphs;
    const int64_t W = o.W, H = o.H;
    std::vector<float> zbuf((size_t)W * H, 1.0f / 4096.0f);
    const bool is_srgb = format_is_srgb(o.format), bgra = format_is_bgra(o.format);
    for (uint32_t g = 0; g < n_glyphs; ++g) {
        const GlyphInst& gi = gs[g];
        if (gi.content_type_with_srgb[0] != 1) { o.err = "only mask glyphs (content type 1) are restated"; return -1; }
        const uint32_t c = gi.color;
        const uint8_t c8[3] = {(uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
        float src[3];
        for (int k = 0; k < 3; ++k) src[k] = gi.content_type_with_srgb[1] == 1 ? srgb().decode[c8[k]] : from_unorm8(c8[k]);
        const float ca = from_unorm8((uint8_t)(c >> 24));
        for (int64_t dy = 0; dy < gi.dim[1]; ++dy)
            for (int64_t dx = 0; dx < gi.dim[0]; ++dx) {
                const int64_t px = (int64_t)gi.pos[0] + dx, py = (int64_t)gi.pos[1] + dy;
                if (px < 0 || py < 0 || px >= W || py >= H) continue;
                const size_t pix = (size_t)py * W + (size_t)px;
                if (!(depth > zbuf[pix])) continue;      /* CompareFunction::Greater */
                zbuf[pix] = depth;
                const uint32_t ax = gi.uv[0] + (uint32_t)dx, ay = gi.uv[1] + (uint32_t)dy;
                const float mask = ax < aw && ay < ah ? from_unorm8(atlas[(size_t)ay * aw + ax]) : 0.0f;      /* (outside the atlas: transparent) */
                const float sa = ca * mask;
                uint8_t* out = rgba + (size_t)py * pitch + (size_t)px * 4;
                uint8_t d8[4] = {out[0], out[1], out[2], out[3]};
                if (bgra) std::swap(d8[0], d8[2]);
                uint8_t r8[4];
                for (int k = 0; k < 3; ++k) {
                    const float dst = is_srgb ? srgb().decode[d8[k]] : from_unorm8(d8[k]);
                    const float v = src[k] * sa + dst * (1.0f - sa);
                    r8[k] = is_srgb ? srgb().encode(v) : unorm8(v);
                }
                r8[3] = unorm8(sa + from_unorm8(d8[3]) * (1.0f - sa));
                if (bgra) std::swap(r8[0], r8[2]);
                out[0] = r8[0]; out[1] = r8[1]; out[2] = r8[2]; out[3] = r8[3];
            }
    }
    return 0;
}

int oracle_read_normals(void* p, int32_t lat, int32_t lon, uint8_t* out) {
    Oracle& o = *(Oracle*)p;
    Tile* t = o.find(lat, lon);
    if (!t) { o.err = "no such tile"; return -1; }
    memcpy(out, t->nrm.data(), t->nrm.size());
    return 0;
}

/* RenderEngine::get_visible_labels: render_engine.rs:338-396, over the depth of the last oracle_render, indexed with
 * the reference's pad_256 row pitch (:364-370).  glam Mat4::project_point3 = ((x_axis*x, y_axis*y + ., z_axis*z + .),
 * w_axis + .) / w in f32 without fma. */
int oracle_visible_peaks(void* p, uint32_t n, const float* peaks_xyz, uint8_t* visible, uint32_t* xy) {
    Oracle& o = *(Oracle*)p;
    const Frame& f = o.frame;
    if (f.W != o.W || f.H != o.H || f.depth.empty()) { o.err = "render first"; return -1; }
    /* the depth read buffer as the reference lays it out */
    const uint32_t pitch = ((o.W * 4 - 1) / 256 + 1) * 256;
    std::vector<uint8_t> buf((size_t)pitch * o.H, 0);
    for (uint32_t y = 0; y < o.H; ++y) memcpy(&buf[(size_t)y * pitch], &f.depth[(size_t)y * o.W], (size_t)o.W * 4);
    const float* m = o.u.camera_proj;
    for (uint32_t i = 0; i < n; ++i) {
        const float X = peaks_xyz[3 * i], Y = peaks_xyz[3 * i + 1], Z = peaks_xyz[3 * i + 2];
        float res[4];
        for (int r = 0; r < 4; ++r) res[r] = m[r] * X;
        for (int r = 0; r < 4; ++r) res[r] = m[4 + r] * Y + res[r];
        for (int r = 0; r < 4; ++r) res[r] = m[8 + r] * Z + res[r];
        for (int r = 0; r < 4; ++r) res[r] = m[12 + r] + res[r];
        const float px = res[0] / res[3], py = res[1] / res[3], pz = res[2] / res[3];
        visible[i] = 0; xy[2 * i] = 0; xy[2 * i + 1] = 0;
        if (px > -1.0f && px < 1.0f && py > -1.0f && py < 1.0f && pz < 1.0f) {
            const uint32_t x_pos = (uint32_t)(0.5f * (px + 1.0f) * (float)o.W);
            const uint32_t y_pos = (uint32_t)(-0.5f * (py - 1.0f) * (float)o.H);
            const size_t pos = (size_t)x_pos * 4 + (size_t)y_pos * pitch;
            if (x_pos >= o.W || pos + 4 > buf.size()) continue;    /* the reference would panic here */
            float depth_value;
            memcpy(&depth_value, &buf[pos], 4);
            const float terrain_distance = dist_from_depth(depth_value);
            const float peak_distance = dist_from_depth(pz);
            if (peak_distance - 10.0f < terrain_distance) { visible[i] = 1; xy[2 * i] = x_pos; xy[2 * i + 1] = y_pos; }
        }
    }
    return 0;
}

/* UiController::get_locations_range: control/ui_controller.rs:61-83, restated line by line (incl. the
 * `.min(-90).max(89)` that pins the sort centre's latitude to 89). */
uint32_t oracle_locations_range(float latitude, float longitude, float range_dist, int32_t* out, uint32_t cap) {
    int center0 = std::max(std::min((int)floorf(latitude), -90), 89);
    int center1 = ((int)(floorf(longitude) + 540.0f)) % 360 - 180;
    float lat_cos = cosf(to_radians_rs(latitude));
    float arc_factor = 0.5f * range_dist / R0;
    float arc_factor_sin = sinf(arc_factor);
    float afs_sq = arc_factor_sin * arc_factor_sin;
    const float to_deg = 180.0f / 3.14159265358979323846f;
    float dlon = acosf(1.0f - afs_sq / lat_cos / lat_cos) * to_deg;
    float dlat = acosf(1.0f - afs_sq) * to_deg;
    int lat_start = std::max((int)floorf(latitude - dlat), -90);
    int lat_end = std::min((int)floorf(latitude + dlat), 89);
    int lon_start = (int)floorf(longitude - dlon);
    int lon_end = (int)floorf(longitude + dlon);
    std::vector<std::pair<int, int>> v;
    for (int lat = lat_start; lat <= lat_end; ++lat)
        for (int lon = lon_start; lon <= lon_end; ++lon) v.push_back({lat, lon});
    std::stable_sort(v.begin(), v.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        return std::make_pair(std::abs(a.first - center0), std::abs(a.second - center1)) <
               std::make_pair(std::abs(b.first - center0), std::abs(b.second - center1));
    });
    uint32_t n = 0;
    for (auto& p : v) {
        if (n < cap) { out[2 * n] = p.first; out[2 * n + 1] = (p.second + 540) % 360 - 180; }
        ++n;
    }
    return n;
}

/* host-side helpers (glam restatement) */
void oracle_camera_uniforms(const float eye[3], float yaw, float pitch, float fov_y, float width, float height,
                            float sun_theta_deg, float sun_phi_deg, int32_t view_mode, void* out160) {
    Uniforms u = camera_uniforms(eye, yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode);
    memcpy(out160, &u, sizeof u);
}
void oracle_terrain_uniforms(const float raster_point[2], const float model_point[2], const float pixel_scale[2],
                             uint32_t w, uint32_t h, void* out96) {
    TerrainUniforms t = terrain_uniforms_new(raster_point, model_point, pixel_scale, w, h);
    memcpy(out96, &t, sizeof t);
}
void oracle_geometry_transform(float h, float lon_deg, float lat_deg, float out[3]) {
    v3 r = geometry_transform(h, lon_deg, lat_deg);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float oracle_dist_from_depth(float d) { return dist_from_depth(d); }
uint32_t oracle_pad_256(uint32_t size) { return ((size - 1) / 256 + 1) * 256; } /* data/mod.rs:9-11 */

/* Coverage probe for known-answer tests: rasterise ONE triangle given directly in framebuffer coordinates
 * (pixels, f32) with z_ndc = z and w = 1 into a WxH target; out[p] = 1 + number of times pixel p was shaded
 * minus 1, i.e. the count of fragments that passed coverage (depth test disabled by clearing between calls is
 * the caller's business: counts accumulate across calls so shared edges can be checked for double hits). */
void oracle_coverage_probe(uint32_t W, uint32_t H, const float xy[6], uint32_t* counts) {
    ScreenVert s[3];
    for (int k = 0; k < 3; ++k) {
        s[k].X = (int64_t)rintf(xy[2 * k] * 256.0f);
        s[k].Y = (int64_t)rintf(xy[2 * k + 1] * 256.0f);
        s[k].z = 0.5f;
        s[k].w = 1.0f;
    }
    Frame f;
    f.W = W; f.H = H;
    f.depth.assign((size_t)W * H, 1.0f);
    f.color.assign((size_t)W * H * 4, 0);
    Uniforms u;
    memset(&u, 0, sizeof u);
    u.view_mode = 2;
    VSOut v;
    memset(&v, 0, sizeof v);
    v.clip[3] = 1.0f;
    raster_triangle(f, u, v, v, v, s[0], s[1], s[2]);
    for (size_t p = 0; p < (size_t)W * H; ++p)
        if (f.depth[p] < 1.0f) counts[p] += 1;
}

/* math probes for cross-checking the product's device header */
void oracle_sincos(const float* x, float* s, float* c, size_t n) {
    for (size_t i = 0; i < n; ++i) sincos_spec(x[i], &s[i], &c[i]);
}
void oracle_srgb_tables(float decode[256], float thresh[255]) {
    memcpy(decode, srgb().decode, sizeof(float) * 256);
    memcpy(thresh, srgb().thresh, sizeof(float) * 255);
}
uint8_t oracle_srgb_encode(float l) { return srgb().encode(l); }
void oracle_vs_main_probe(void* p, int32_t lat, int32_t lon, uint32_t x, uint32_t y, float out[10]) {
    Oracle& o = *(Oracle*)p;
    Tile* t = o.find(lat, lon);
    if (!t) return;
    VSOut v = vs_main(*t, x, y, o.u);
    memcpy(out, v.clip, 16);
    out[4] = v.wpos.x; out[5] = v.wpos.y; out[6] = v.wpos.z;
    out[7] = v.wnrm.x; out[8] = v.wnrm.y; out[9] = v.wnrm.z;
}

}  // extern "C"
