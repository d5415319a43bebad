// topo_capi.cpp -- the extern "C" boundary (include/topo_hip.h) over topo::TerrainRenderer.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/topo_hip_test.h"      // the test hooks are defined here too (their own header: not part of the boundary)
#include "geotiff.hpp"
#include "terrain_renderer.hpp"

struct topo_ctx {
    topo::TerrainRenderer* r = nullptr;
};

namespace {
thread_local std::string g_create_error;   // topo_last_error(NULL) reports why topo_create failed
}

#define TOPO_GUARD(ctx)                       \
    if (!(ctx) || !(ctx)->r) return TOPO_ERR_INVALID

extern "C" {

int topo_create(topo_ctx** out, int hip_device, uint32_t width, uint32_t height, uint32_t color_format) {
    if (!out) return TOPO_ERR_INVALID;
    *out = nullptr;
    topo::TerrainRenderer* r = nullptr;
    int rc;
    try {
        rc = topo::TerrainRenderer::create(&r, hip_device, width, height, color_format, &g_create_error);
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return TOPO_ERR_HIP;
    }
    if (rc != TOPO_OK) return rc;
    topo_ctx* c = new (std::nothrow) topo_ctx();
    if (!c) { delete r; return TOPO_ERR_HIP; }
    c->r = r;
    *out = c;
    return TOPO_OK;
}

void topo_destroy(topo_ctx* ctx) {
    if (!ctx) return;
    delete ctx->r;
    delete ctx;
}

const char* topo_last_error(topo_ctx* ctx) {
    if (!ctx || !ctx->r) return g_create_error.c_str();
    return ctx->r->last_error();
}

#define TOPO_CALL(expr)                                         \
    try {                                                       \
        return (expr);                                          \
    } catch (const std::exception&) {                           \
        return TOPO_ERR_HIP;                                    \
    }

int topo_add_terrain(topo_ctx* ctx, int32_t lat, int32_t lon, const float* heights, uint32_t w, uint32_t h,
                     const float rp[2], const float mp[2], const float ps[2]) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->add_terrain(lat, lon, heights, false, w, h, rp, mp, ps));
}

int topo_add_terrain_device(topo_ctx* ctx, int32_t lat, int32_t lon, const float* heights_dev, uint32_t w, uint32_t h,
                            const float rp[2], const float mp[2], const float ps[2]) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->add_terrain(lat, lon, heights_dev, true, w, h, rp, mp, ps));
}

int topo_unload_terrain(topo_ctx* ctx, int32_t lat, int32_t lon) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->unload_terrain(lat, lon));
}

int topo_update(topo_ctx* ctx, uint32_t w, uint32_t h, const topo_uniforms* u, const topo_post_uniforms* pu) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->update(w, h, u, pu));
}

int topo_render(topo_ctx* ctx, uint8_t* rgba, size_t rgba_pitch, float* depth, size_t depth_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->render(rgba, rgba_pitch, depth, depth_pitch));
}

int topo_recompute_normals(topo_ctx* ctx) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->recompute_normals());
}

int topo_render_views_device(topo_ctx* ctx, uint32_t n_views, const topo_uniforms* views, uint32_t width, uint32_t height,
                             uint8_t* rgba_dev, size_t rgba_view_stride, size_t rgba_pitch, float* depth_dev,
                             size_t depth_view_stride, size_t depth_pitch) {
    TOPO_GUARD(ctx);
    topo::OutputParams o{};
    o.rgba = rgba_dev;
    o.rgba_view_stride = rgba_view_stride;
    o.rgba_pitch = rgba_pitch;
    o.depth = depth_dev;
    o.depth_view_stride = depth_view_stride;
    o.depth_pitch = depth_pitch;
    TOPO_CALL(ctx->r->render_views_device(n_views, views, width, height, o));
}

int topo_set_stream(topo_ctx* ctx, void* hip_stream) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->set_stream((hipStream_t)hip_stream));
}

int topo_synchronize(topo_ctx* ctx) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->synchronize());
}

int topo_set_normals_lds_rows(topo_ctx* ctx, int rows) {
    TOPO_GUARD(ctx);
    return ctx->r->set_normals_lds_rows(rows);
}

int topo_debug_set_queue_caps(topo_ctx* ctx, uint32_t big_cap, uint32_t rare_cap) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->set_queue_caps(big_cap, rare_cap));
}

int topo_debug_far_phase_launched(topo_ctx* ctx, int32_t* out) {
    TOPO_GUARD(ctx);
    if (!out) return TOPO_ERR_INVALID;
    *out = ctx->r->last_far_phase() ? 1 : 0;
    return TOPO_OK;
}

int topo_get_timings(topo_ctx* ctx, float out_ms[TOPO_TIMING_SLOTS]) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->get_timings(out_ms));
}

int topo_pin_host_buffer(topo_ctx* ctx, void* buffer, size_t bytes) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->pin_host_buffer(buffer, bytes));
}

int topo_unpin_host_buffer(topo_ctx* ctx, void* buffer) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->unpin_host_buffer(buffer));
}

int topo_get_timing_history(topo_ctx* ctx, uint32_t n_frames, float* out_ms, uint32_t* n_out) {
    TOPO_GUARD(ctx);
    if (!n_out) return TOPO_ERR_INVALID;
    TOPO_CALL(ctx->r->get_timing_history(n_frames, out_ms, n_out));
}

int topo_set_timing_slots(topo_ctx* ctx, uint32_t slot_mask) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->set_timing_slots(slot_mask));
}

int topo_set_occlusion_split(topo_ctx* ctx, float metres) {
    TOPO_GUARD(ctx);
    return ctx->r->set_occlusion_split(metres);
}

int topo_get_counters(topo_ctx* ctx, uint32_t out[6]) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->get_counters(out));
}

int topo_read_normals(topo_ctx* ctx, int32_t lat, int32_t lon, uint8_t* out) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->read_normals(lat, lon, out));
}

int topo_read_tile_tables(topo_ctx* ctx, int32_t lat, int32_t lon, float* minmax_out, float* trig_out, double* bounds_out, uint32_t* n_blocks_out) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->read_tile_tables(lat, lon, minmax_out, trig_out, bounds_out, n_blocks_out));
}

int topo_visible_peaks(topo_ctx* ctx, uint32_t n_peaks, const float* peaks_xyz, uint8_t* visible_out, uint32_t* xy_out) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->visible_peaks(n_peaks, peaks_xyz, visible_out, xy_out));
}

int topo_visible_peaks_device(topo_ctx* ctx, const topo_uniforms* view, uint32_t width, uint32_t height, const float* depth_dev,
                              size_t depth_pitch, uint32_t n_peaks, const float* peaks_xyz_dev, uint8_t* visible_dev,
                              uint32_t* xy_dev) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->visible_peaks_device(view, width, height, depth_dev, depth_pitch, n_peaks, peaks_xyz_dev, visible_dev, xy_dev));
}

int topo_probe_sincos(topo_ctx* ctx, const float* x, float* s, float* c, size_t n) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->probe_sincos(x, s, c, n));
}

int topo_set_pipeline_depth(topo_ctx* ctx, int32_t depth) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->set_pipeline_depth(depth));
}

int topo_join(topo_ctx* ctx) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->join_frames());
}

// ---- multi-GPU panorama / viewpoint batch (panorama.cpp) -----------------------------------------------------
struct topo_comm {
    topo::Comm* c = nullptr;
};

int topo_comm_unique_id(uint8_t out_id[TOPO_COMM_ID_BYTES]) {
    if (!out_id) return TOPO_ERR_INVALID;
    try {
        return topo::comm_unique_id(out_id, &g_create_error);
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return TOPO_ERR_HIP;
    }
}

int topo_comm_init(topo_comm** out, int hip_device, const uint8_t id[TOPO_COMM_ID_BYTES], int rank, int world) {
    if (!out || (world > 1 && !id)) return TOPO_ERR_INVALID;
    *out = nullptr;
    topo::Comm* c = nullptr;
    static const uint8_t zero[TOPO_COMM_ID_BYTES] = {};
    int rc;
    try {
        rc = topo::comm_init(&c, hip_device, id ? id : zero, rank, world, &g_create_error);
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return TOPO_ERR_HIP;
    }
    if (rc != TOPO_OK) return rc;
    *out = new topo_comm();
    (*out)->c = c;
    return TOPO_OK;
}

int topo_comm_from_nccl(topo_comm** out, void* nccl_comm, int rank, int world) {
    if (!out) return TOPO_ERR_INVALID;
    *out = nullptr;
    topo::Comm* c = nullptr;
    if (int rc = topo::comm_from_nccl(&c, nccl_comm, rank, world, &g_create_error)) return rc;
    *out = new topo_comm();
    (*out)->c = c;
    return TOPO_OK;
}

void topo_comm_destroy(topo_comm* comm) {
    if (!comm) return;
    topo::comm_destroy(comm->c);
    delete comm;
}

void topo_panorama_sector_range(int rank, int world, uint32_t* first, uint32_t* count) {
    if (first && count && world >= 1 && rank >= 0 && rank < world) topo::panorama_sector_range(rank, world, first, count);
}

uint32_t topo_panorama_slots(int world, uint32_t sector_w, uint32_t sector_h, topo_panorama_slot* out, uint32_t cap) {
    if (world < 1 || topo::kPanoramaSectors % (uint32_t)world != 0 || sector_w == 0 || sector_h == 0) return 0;
    return topo::panorama_slots(world, sector_w, sector_h, out, cap);
}

int topo_render_panorama(topo_ctx* ctx, topo_comm* comm, const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h,
                         float sun_theta_deg, float sun_phi_deg, int32_t view_mode, uint8_t* strip_dev, float* depth_dev) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->render_panorama(comm ? comm->c : nullptr, eye, yaw0, pitch, sector_w, sector_h, sun_theta_deg, sun_phi_deg, view_mode,
                                      strip_dev, depth_dev));
}

int topo_render_batch(topo_ctx* ctx, uint32_t n_viewpoints, const float* eyes_xyz, const float* yaw0, const float* sun_theta_phi_deg,
                      float pitch, uint32_t sector_w, uint32_t sector_h, int32_t view_mode, uint8_t* rgba_dev, float* depth_dev) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->render_batch(n_viewpoints, eyes_xyz, yaw0, sun_theta_phi_deg, pitch, sector_w, sector_h, view_mode, rgba_dev, depth_dev));
}

static uint32_t put_pairs(const std::vector<std::pair<int32_t, int32_t>>& v, int32_t* out, uint32_t cap) {
    for (uint32_t i = 0; i < v.size() && i < cap && out; ++i) { out[2 * i] = v[i].first; out[2 * i + 1] = v[i].second; }
    return (uint32_t)v.size();
}

void topo_change_location_plan(float latitude, float longitude, float range_dist, const int32_t* loaded_lat_lon, uint32_t n_loaded,
                               int32_t* unload_out, uint32_t unload_cap, uint32_t* n_unload, int32_t* request_out, uint32_t request_cap,
                               uint32_t* n_request) {
    std::vector<std::pair<int32_t, int32_t>> unload, request;
    topo::change_location_plan(latitude, longitude, range_dist, loaded_lat_lon, loaded_lat_lon ? n_loaded : 0, unload, request);
    const uint32_t nu = put_pairs(unload, unload_out, unload_cap), nr = put_pairs(request, request_out, request_cap);
    if (n_unload) *n_unload = nu;
    if (n_request) *n_request = nr;
}

int topo_change_location(topo_ctx* ctx, float latitude, float longitude, float range_dist, int32_t* request_out, uint32_t request_cap,
                         uint32_t* n_request, uint32_t* n_unloaded) {
    TOPO_GUARD(ctx);
    try {
        std::vector<std::pair<int32_t, int32_t>> request;
        const int rc = ctx->r->change_location(latitude, longitude, range_dist, request, n_unloaded);
        if (rc != TOPO_OK) return rc;
        const uint32_t nr = put_pairs(request, request_out, request_cap);
        if (n_request) *n_request = nr;
        return TOPO_OK;
    } catch (const std::exception&) {
        return TOPO_ERR_HIP;
    }
}

int topo_overlay_lines(topo_ctx* ctx, const topo_overlay_vertex* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices,
                       float line_width, uint8_t* rgba, size_t rgba_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->overlay_lines(vertices, n_vertices, indices, n_indices, line_width, rgba, rgba_pitch));
}

int topo_overlay_lines_device(topo_ctx* ctx, const topo_overlay_vertex* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices,
                              float line_width, uint8_t* rgba_dev, size_t rgba_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->overlay_lines_device(vertices, n_vertices, indices, n_indices, line_width, rgba_dev, rgba_pitch));
}

int topo_overlay_glyphs(topo_ctx* ctx, const topo_glyph* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas_mask, uint32_t atlas_w, uint32_t atlas_h,
                        uint8_t* rgba, size_t rgba_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->overlay_glyphs(glyphs, n_glyphs, depth, atlas_mask, atlas_w, atlas_h, rgba, rgba_pitch));
}

int topo_overlay_glyphs_device(topo_ctx* ctx, const topo_glyph* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas_mask, uint32_t atlas_w,
                               uint32_t atlas_h, uint8_t* rgba_dev, size_t rgba_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->overlay_glyphs_device(glyphs, n_glyphs, depth, atlas_mask, atlas_w, atlas_h, rgba_dev, rgba_pitch));
}

int topo_frame_status(topo_ctx* ctx, uint32_t out[4]) {
    TOPO_GUARD(ctx);
    if (!out) return TOPO_ERR_INVALID;
    TOPO_CALL(ctx->r->frame_status(out));
}

int topo_probe_div(topo_ctx* ctx, int32_t kind, const float* x, const float* y, float* out, size_t n) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->probe_div(kind, x, y, out, n));
}

void topo_camera_uniforms(const float eye[3], float yaw, float pitch, float fov_y, float width, float height,
                          float sun_theta_deg, float sun_phi_deg, int32_t view_mode, topo_uniforms* out) {
    topo::camera_uniforms(eye, yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode, out);
}

float topo_sector_fov_y(uint32_t sector_w, uint32_t sector_h, uint32_t n_sectors) {
    return (float)(2.0 * atan(tan(3.14159265358979323846 / (double)n_sectors) * (double)sector_h / (double)sector_w));
}

void topo_panorama_uniforms(const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h, float sun_theta_deg,
                            float sun_phi_deg, int32_t view_mode, uint32_t n_sectors, topo_uniforms* out) {
    topo::panorama_uniforms(eye, yaw0, pitch, sector_w, sector_h, sun_theta_deg, sun_phi_deg, view_mode, n_sectors, out);
}

int topo_render_device(topo_ctx* ctx, uint8_t* rgba_dev, size_t rgba_pitch, float* depth_dev, size_t depth_pitch) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->render_device(rgba_dev, rgba_pitch, depth_dev, depth_pitch));
}

void topo_terrain_uniforms(const float rp[2], const float mp[2], const float ps[2], uint32_t w, uint32_t h, float out[24]) {
    memset(out, 0, 24 * sizeof(float));
    out[0] = rp[0]; out[1] = rp[1]; out[2] = mp[0]; out[3] = mp[1]; out[4] = ps[0]; out[5] = ps[1];
    out[6] = (float)w; out[7] = (float)h;
    float rot[9];
    topo::terrain_rotation(mp[0], mp[1], rot);
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out[8 + c * 4 + r] = rot[c * 3 + r];   // Mat4::from_mat3
    out[8 + 15] = 1.0f;
}

void topo_geometry_transform(float h, float lon_deg, float lat_deg, float out[3]) {
    topo::geometry_transform(h, lon_deg, lat_deg, out);
}

float topo_dist_from_depth(float depth) { return topo::kFar * topo::kNear / (topo::kFar - depth * (topo::kFar - topo::kNear)); }

uint32_t topo_pad_256(uint32_t size) { return ((size - 1) / 256 + 1) * 256; }

int topo_coordinate_transform(const double* pixel_scale, uint32_t n_pixel_scale, const double* tie_points, uint32_t n_tie_points,
                              const double* model_transformation, float rp[2], float mp[2], float ps[2]) {
    if (!rp || !mp || !ps) return TOPO_ERR_INVALID;
    if (model_transformation) return TOPO_ERR_UNSUPPORTED;            // IncorrectGeoTags
    if (!pixel_scale || !tie_points) return TOPO_ERR_UNSUPPORTED;     // IncorrectGeoTags
    if (n_pixel_scale != 3 || n_tie_points != 6) return TOPO_ERR_INVALID;   // IncorrectGeoTagData
    rp[0] = (float)tie_points[0]; rp[1] = (float)tie_points[1];
    mp[0] = (float)tie_points[3]; mp[1] = (float)tie_points[4];
    ps[0] = (float)pixel_scale[0]; ps[1] = (float)pixel_scale[1];
    return TOPO_OK;
}

void topo_to_model(const float rp[2], const float mp[2], const float ps[2], float x, float y, float out[2]) {
    out[0] = (x - rp[0]) * ps[0] + mp[0];
    out[1] = (y - rp[1]) * -ps[1] + mp[1];
}

void topo_to_raster(const float rp[2], const float mp[2], const float ps[2], float lon, float lat, float out[2]) {
    out[0] = (lon - mp[0]) / ps[0] + rp[0];
    out[1] = (lat - mp[1]) / -ps[1] + rp[1];
}

namespace {
size_t rust_f32_as_usize(float v) {   // `as usize`: NaN -> 0, saturating at both ends, truncation toward zero
    if (!(v > 0.0f)) return 0;
    if (v >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)v;
}
}  // namespace

int topo_height_value_at(const float* heights, uint32_t w, uint32_t h, const float rp[2], const float mp[2], const float ps[2],
                         double longitude, double latitude, float* out) {
    if (!heights || !rp || !mp || !ps || !out) return TOPO_ERR_INVALID;
    float r[2];
    topo_to_raster(rp, mp, ps, (float)longitude, (float)latitude, r);
    const size_t ry = rust_f32_as_usize(r[1]), rx = rust_f32_as_usize(r[0]);
    const unsigned __int128 index = (unsigned __int128)ry * w + rx;   // usize arithmetic cannot overflow for any real tile
    if (index >= (unsigned __int128)w * h) return TOPO_ERR_NOT_FOUND;
    *out = heights[(size_t)index];
    return TOPO_OK;
}

int topo_geotiff_info(const uint8_t* bytes, size_t n, uint32_t* w, uint32_t* h, float rp[2], float mp[2], float ps[2]) {
    if (!bytes || !w || !h || !rp || !mp || !ps) return TOPO_ERR_INVALID;
    topo::TiffInfo ti;
    std::string e;
    if (int rc = topo::tiff_parse(bytes, n, ti, e)) return rc;
    *w = ti.width;
    *h = ti.height;
    return topo::geotiff_transform(ti, rp, mp, ps);
}

int topo_geotiff_decode(topo_ctx* ctx, const uint8_t* bytes, size_t n, float* heights_out, size_t capacity) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->geotiff_decode(bytes, n, heights_out, capacity));
}

int topo_add_terrain_geotiff(topo_ctx* ctx, int32_t lat, int32_t lon, const uint8_t* bytes, size_t n) {
    TOPO_GUARD(ctx);
    TOPO_CALL(ctx->r->add_terrain_geotiff(lat, lon, bytes, n));
}

uint32_t topo_locations_range(float latitude, float longitude, float range_dist, int32_t* out, uint32_t cap) {
    return topo::locations_range(latitude, longitude, range_dist, out, cap);
}

void topo_synth_tile(int32_t lat, int32_t lon, uint32_t w, uint32_t h, uint32_t seed, float* out) {
    topo::synth_tile(lat, lon, w, h, seed, out);
}

}  // extern "C"
