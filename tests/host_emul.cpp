// tests/host_emul.cpp -- TEST-ONLY g++ build of the product's TOPO_HD pipeline functions
// (topo-renderer_amd/csrc/topo_math.h, topo_pipeline.h), driven by sequential loops that mirror the
// HIP kernels' structure.  It lets the CPU test-suite check the product's arithmetic against the oracle
// bit for bit without a GPU.  It is NOT a fallback: nothing in the product links or loads it, and
// libtopo_hip.so refuses to work without a HIP device.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../topo-renderer_amd/csrc/topo_pipeline.h"

using namespace topo;

extern "C" {

struct EmulTile {
    const float* heights;
    uint32_t* normals;
    float tu[24];   // TerrainUniforms, 96 B
};

static TileDev make_dev(const EmulTile& e) {
    TileDev t{};
    t.heights = e.heights; t.normals = e.normals; t.block_minmax = nullptr;
    t.raster_x = e.tu[0]; t.raster_y = e.tu[1]; t.model_x = e.tu[2]; t.model_y = e.tu[3];
    t.scale_x = e.tu[4]; t.scale_y = e.tu[5];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) t.rot[c * 3 + r] = e.tu[8 + c * 4 + r];
    return t;
}

uint32_t emul_fastdiv(uint32_t n, uint32_t d) { return fastdiv(n, fastdiv_make(d)); }
void emul_markstein_div(const float* x, float C, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = markstein_div(x[i], C, 1.0f / C); }
void emul_srgb_encode_both(const float* x, uint8_t* by_probes, uint8_t* by_lut, size_t n) {
    float thresh[258];
    for (int i = 0; i < 256; ++i) thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]);
    thresh[255] = thresh[256] = thresh[257] = NAN;   // as k_resolve pads its LDS copy
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(TOPO_SRGB_LUT12_WORDS);   // little-endian host
    for (size_t i = 0; i < n; ++i) {
        by_probes[i] = (uint8_t)srgb_encode(thresh, x[i]);
        by_lut[i] = (uint8_t)srgb_encode_lut(thresh, lut, x[i]);
    }
}
// normal_texel_fast against normal_texel over n stencils: returns the number of texels where the fast route claimed a
// result that differs from the spec's (must be 0); *n_fast = how many it claimed at all
uint64_t emul_normal_fast_check(const float* xs, const float* ys, const float* h4, size_t n, uint64_t* n_fast) {
    uint64_t bad = 0, fast = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t t = 0;
        if (normal_texel_fast(xs[i], ys[i], h4[4 * i], h4[4 * i + 1], h4[4 * i + 2], h4[4 * i + 3], t)) {
            ++fast;
            if (t != normal_texel(xs[i], ys[i], h4[4 * i], h4[4 * i + 1], h4[4 * i + 2], h4[4 * i + 3])) ++bad;
        }
    }
    *n_fast = fast;
    return bad;
}
void emul_sincos(const float* x, float* s, float* c, size_t n) { for (size_t i = 0; i < n; ++i) sincos_f(x[i], s[i], c[i]); }

void emul_normals_interior(const EmulTile* e, int W, int H) {
    TileDev t = make_dev(*e);
    const float xs = deg2rad(t.scale_x) * kR0, ys0 = deg2rad(t.scale_y) * kR0;
    for (int gy = 1; gy < H - 1; ++gy) for (int gx = 1; gx < W - 1; ++gx) {
        const float latitude = ((float)gy - t.raster_y) * -t.scale_y + t.model_y;
        const float ys = ys0 * cos_f(deg2rad(latitude));
        const float* h = t.heights;
        t.normals[(size_t)gy * W + gx] = normal_texel(xs, ys, h[(size_t)(gy - 1) * W + gx], h[(size_t)gy * W + gx - 1],
                                                      h[(size_t)gy * W + gx + 1], h[(size_t)(gy + 1) * W + gx]);
    }
}

void emul_normals_edge(const EmulTile* elt, const EmulTile* erb, const EmulTile* euni, int W, int H, int top_bottom) {
    TileDev lt = make_dev(*elt), rb = make_dev(*erb), u = make_dev(*euni);
    const float xs = deg2rad(fabsf(u.scale_x)) * kR0, ys0 = deg2rad(fabsf(u.scale_y)) * kR0;
    for (int id = 1; id < W - 1; ++id) {
        if (!top_bottom) {
            if (id >= H - 1) continue;
            const float latitude = ((float)id - u.raster_y) * -u.scale_y + u.model_y;
            const float ys = ys0 * cos_f(deg2rad(latitude));
            const int lx = W - 1, ly = id;
            const uint32_t tx = normal_texel(xs, ys, lt.heights[(size_t)(ly - 1) * W + lx], lt.heights[(size_t)ly * W + lx - 1],
                                             rb.heights[(size_t)id * W + 1], lt.heights[(size_t)(ly + 1) * W + lx]);
            lt.normals[(size_t)ly * W + lx] = tx; rb.normals[(size_t)id * W] = tx;
        } else {
            const float latitude = ((float)(H - 1) - u.raster_y) * -u.scale_y + u.model_y;
            const float ys = ys0 * cos_f(deg2rad(latitude));
            const int ty = H - 1;
            const uint32_t tx = normal_texel(xs, ys, lt.heights[(size_t)(ty - 1) * W + id], lt.heights[(size_t)ty * W + id - 1],
                                             lt.heights[(size_t)ty * W + id + 1], rb.heights[(size_t)W + id]);
            lt.normals[(size_t)ty * W + id] = tx; rb.normals[id] = tx;
        }
    }
}

void emul_normals_corner(const EmulTile* elt, const EmulTile* ert, const EmulTile* elb, const EmulTile* erb,
                         const EmulTile* euni, int W, int H) {
    TileDev lt = make_dev(*elt), rt = make_dev(*ert), lb = make_dev(*elb), rb = make_dev(*erb), u = make_dev(*euni);
    const float latitude = ((float)(H - 1) - u.raster_y) * -u.scale_y + u.model_y;
    const float xs = deg2rad(fabsf(u.scale_x)) * kR0;
    const float ys = deg2rad(fabsf(u.scale_y)) * kR0 * cos_f(deg2rad(latitude));
    const uint32_t tx = normal_texel(xs, ys, rb.heights[(size_t)(H - 2) * W], lt.heights[(size_t)(H - 1) * W + W - 2],
                                     rt.heights[(size_t)(H - 1) * W + 1], lb.heights[(size_t)W + W - 1]);
    lt.normals[(size_t)(H - 1) * W + W - 1] = tx; rt.normals[(size_t)(H - 1) * W] = tx; lb.normals[W - 1] = tx; rb.normals[0] = tx;
}

static void emul_emit(const TriSetup& ts, uint64_t* vis, int W, uint32_t id) {
    for (int py = ts.py0; py <= ts.py1; ++py) for (int px = ts.px0; px <= ts.px1; ++px) {
        float z, b[3];
        if (triangle_pixel(ts, px, py, z, b)) {
            const uint64_t key = vis_key(z, id);
            if (key < vis[(size_t)py * W + px]) vis[(size_t)py * W + px] = key;
        }
    }
}

static int g_split = 0;
void emul_set_split(int on) { g_split = on; }
static float g_pixelize[3] = {0.0f, 0.0f, 100.0f};      // PostprocessingUniforms: viewport, pixelize_n (100: the branch off)
void emul_set_post(float vw, float vh, float n) { g_pixelize[0] = vw; g_pixelize[1] = vh; g_pixelize[2] = n; }

// tiles must be given in draw (BTreeMap) order
int emul_render(const EmulTile* tiles, uint32_t n_tiles, uint32_t tile_w, uint32_t tile_h, const float* uniforms40,
                int W, int H, uint8_t* rgba, float* depth) {
    ViewDev view{};
    memcpy(view.proj, uniforms40, 64);
    view.cam_x = uniforms40[32]; view.cam_y = uniforms40[33];
    memcpy(view.sun, uniforms40 + 36, 12);
    memcpy(&view.view_mode, uniforms40 + 39, 4);
    std::vector<TileDev> td;
    std::vector<std::vector<float>> trig(n_tiles);          // what k_block_minmax tabulates in the load phase
    float ndec[256];                                         // what k_resolve tabulates in LDS
    for (uint32_t c = 0; c < 256; ++c) ndec[c] = normal_channel(c);
    for (uint32_t i = 0; i < n_tiles; ++i) {
        TileDev t = make_dev(tiles[i]);
        trig[i].resize(2 * (size_t)(tile_w + tile_h));
        for (uint32_t x = 0; x < tile_w; ++x) sincos_f(vertex_lon(t, x), trig[i][2 * x], trig[i][2 * x + 1]);
        for (uint32_t y = 0; y < tile_h; ++y) sincos_f(vertex_lat(t, y), trig[i][2 * (tile_w + y)], trig[i][2 * (tile_w + y) + 1]);
        t.trig_lon = trig[i].data();
        t.trig_lat = trig[i].data() + 2 * (size_t)tile_w;
        td.push_back(t);
    }
    const FastDiv div_hm1 = fastdiv_make(tile_h - 1), div_tris = fastdiv_make(2u * (tile_w - 1) * (tile_h - 1));
    std::vector<uint64_t> vis((size_t)W * H, kVisClear);
    const uint32_t tris_per_tile = 2u * (tile_w - 1) * (tile_h - 1);
    for (uint32_t rank = 0; rank < n_tiles; ++rank) {
        const TileDev& t = td[rank];
        std::vector<SVert> sv((size_t)tile_w * tile_h);
        for (uint32_t vy = 0; vy < tile_h; ++vy) for (uint32_t vx = 0; vx < tile_w; ++vx) {
            float clip[4];
            vertex_clip(t, view, vx, vy, t.heights[(size_t)vy * tile_w + vx], clip);
            clip_to_screen(clip, (float)W, (float)H, sv[(size_t)vy * tile_w + vx]);
        }
        for (uint32_t i = 0; i + 1 < tile_w; ++i) for (uint32_t j = 0; j + 1 < tile_h; ++j) {
            const SVert a = sv[(size_t)j * tile_w + i], b = sv[(size_t)(j + 1) * tile_w + i];
            const SVert c = sv[(size_t)j * tile_w + i + 1], d = sv[(size_t)(j + 1) * tile_w + i + 1];
            const bool even = ((i + j) & 1u) == 0;
            for (uint32_t k = 0; k < 2; ++k) {
                const SVert& s0 = k == 0 ? a : d;
                const SVert& s1 = k == 0 ? b : c;
                const SVert& s2 = k == 0 ? (even ? d : c) : (even ? a : b);
                const uint32_t tri = (i * (tile_h - 1) + j) * 2 + k, draw = rank * tris_per_tile + tri;
                const int nnear = (s0.flag == kVtxNear) + (s1.flag == kVtxNear) + (s2.flag == kVtxNear);
                if (nnear == 3) continue;
                if (nnear != 0) {
                    for (uint32_t fan = 0; fan < 2; ++fan) {
                        ResolvedTri r;
                        if (resolve_triangle(t, tile_w, div_hm1, tile_h - 1, view, W, H, tri, fan, r)) emul_emit(r.ts, vis.data(), W, (draw << 1) | fan);
                    }
                    continue;
                }
                if ((s0.flag | s1.flag | s2.flag) != kVtxOk) continue;
                TriSetup ts;
                if (triangle_setup(s0, s1, s2, W, H, ts)) emul_emit(ts, vis.data(), W, draw << 1);
            }
        }
    }
    float thresh[256], decode[256];
    for (int i = 0; i < 256; ++i) { thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]); decode[i] = bits_f(TOPO_SRGB_DECODE_BITS[i]); }
    auto vdepth = [&](int x, int y) {
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x); y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        return bits_f((uint32_t)(vis[(size_t)y * W + x] >> 32));
    };
    for (int py = 0; py < H; ++py) for (int px = 0; px < W; ++px) {
        const uint64_t key = vis[(size_t)py * W + px];
        const float dc = bits_f((uint32_t)(key >> 32));
        const uint32_t id = (uint32_t)key;
        float lin[4] = {0.0f, 0.71f, 0.885f, 1.0f};
        if (id != kNoTri) {
            const uint32_t draw = id >> 1, fan = id & 1u, rank = fastdiv(draw, div_tris), tri = draw - rank * tris_per_tile;
            f3 wpos = {0.0f, 0.0f, 0.0f}, wnrm;
            if (g_split) {      // the two-step route k_resolve takes for waves whose pixels share few winners
                TriRecord rec;
                // the reference pixel k_resolve uses: the origin of the 64 x 4 px strip the pixel lies in
                const int32_t ox = px & ~63, oy = py & ~3;
                resolve_setup(td[rank], tile_w, div_hm1, tile_h - 1, view, W, H, tri, fan, ndec, ox, oy, rec);
                if (!resolve_pixel(rec, pixel_at(W, H, px, py, ox, oy), wpos.x, wpos.y, wnrm)) return -1;
            } else if (!resolve_varyings(td[rank], tile_w, div_hm1, tile_h - 1, view, W, H, tri, fan, ndec, px, py, wpos, wnrm)) return -1;
            shade_fragment(view.view_mode, {view.sun[0], view.sun[1], view.sun[2]}, view.cam_x, view.cam_y, (float)px + 0.5f,
                           (float)py + 0.5f, wpos, wnrm, lin);
        }
        const uint32_t c8 = srgb_encode(thresh, lin[0]) | (srgb_encode(thresh, lin[1]) << 8) | (srgb_encode(thresh, lin[2]) << 16) |
                            (to_unorm8(lin[3]) << 24);
        memcpy(rgba + ((size_t)py * W + px) * 4, &c8, 4);      // the render target; the post pass below reads it
        depth[(size_t)py * W + px] = dc;
    }
    std::vector<uint32_t> pre((size_t)W * H);
    memcpy(pre.data(), rgba, (size_t)W * H * 4);
    const bool pixelize = g_pixelize[2] < 99.99999f;
    for (int py = 0; py < H; ++py) for (int px = 0; px < W; ++px) {
        float dn[8]; int k = 0;
        for (int i = -1; i <= 1; ++i) for (int j = -1; j <= 1; ++j) { if (i == 0 && j == 0) continue; dn[k++] = linear_depth(vdepth(px + i, py + j)); }
        const float dc = bits_f((uint32_t)(vis[(size_t)py * W + px] >> 32));
        uint32_t o;
        if (pixelize) {
            float rc[4];
            sample_pixelized(px, py, g_pixelize[0], g_pixelize[1], g_pixelize[2], W, H, [&](int32_t x, int32_t y, float out[4]) {
                const uint32_t c = pre[(size_t)y * W + x];
                out[0] = decode[c & 255u]; out[1] = decode[(c >> 8) & 255u]; out[2] = decode[(c >> 16) & 255u]; out[3] = from_unorm8(c >> 24);
            }, rc);
            o = post_mix(thresh, rc, linear_depth(dc), dn, true);
        } else {
            o = post_pixel(thresh, decode, pre[(size_t)py * W + px], linear_depth(dc), dn);
        }
        memcpy(rgba + ((size_t)py * W + px) * 4, &o, 4);
    }
    return 0;
}

// coverage of one triangle given in framebuffer pixels (same contract as oracle_coverage_probe)
void emul_coverage_probe(uint32_t W, uint32_t H, const float xy[6], uint32_t* counts) {
    SVert s[3];
    for (int k = 0; k < 3; ++k) {
        s[k].X = (int32_t)rintf(xy[2 * k] * 256.0f); s[k].Y = (int32_t)rintf(xy[2 * k + 1] * 256.0f);
        s[k].z = 0.5f; s[k].flag = kVtxOk;
    }
    TriSetup ts;
    if (!triangle_setup(s[0], s[1], s[2], (int32_t)W, (int32_t)H, ts)) return;
    for (int py = ts.py0; py <= ts.py1; ++py) for (int px = ts.px0; px <= ts.px1; ++px) {
        float z, b[3];
        if (triangle_pixel(ts, px, py, z, b)) counts[(size_t)py * W + px] += 1;
    }
}

// product's project_peak + the depth comparison, over a tightly packed depth image
void emul_visible_peaks(const float* proj, uint32_t W, uint32_t H, const float* depth, uint32_t n, const float* peaks,
                        uint8_t* visible, uint32_t* xy) {
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t x_pos = 0, y_pos = 0; float pd = 0.0f; bool vis = false;
        if (project_peak(proj, peaks[3 * i], peaks[3 * i + 1], peaks[3 * i + 2], (float)W, (float)H, x_pos, y_pos, pd) && x_pos < W && y_pos < H)
            vis = pd - 10.0f < linear_depth(depth[(size_t)y_pos * W + x_pos]);
        visible[i] = vis; xy[2 * i] = vis ? x_pos : 0; xy[2 * i + 1] = vis ? y_pos : 0;
    }
}

// ---- lane bodies of the raster kernels under a bounds-checking sink ---------------------------------------------------
// The sink the kernels use is `atomicMin(vis + pix)` + `dirty[(view base + pix) >> 6] = 1`: an index outside the target
// is a wild write on the GPU.  Here every fragment is checked instead: inside the target, inside the item's region (big
// items), emitted at most once per (item, pixel).  Returns the number of violations; keys[] (W*H, caller-initialised to
// kVisClear) receives the min of the emitted keys, n_frag their count.
struct CheckSink {
    uint64_t* keys; std::vector<uint8_t> seen; int W, H, rx, ry; int violations = 0; uint32_t n = 0;
    CheckSink(uint64_t* k, int w, int h, int rx_, int ry_) : keys(k), seen((size_t)w * h, 0), W(w), H(h), rx(rx_), ry(ry_) {}
    void operator()(size_t pix, uint64_t key) {
        if (pix >= (size_t)W * H) { ++violations; return; }
        const int px = (int)(pix % W), py = (int)(pix / W);
        if (rx >= 0 && (px < rx * 64 || px > rx * 64 + 63 || py < ry * 64 || py > ry * 64 + 63)) ++violations;
        if (seen[pix]) ++violations;
        seen[pix] = 1;
        if (key < keys[pix]) keys[pix] = key;
        ++n;
    }
};

int emul_big_item(const int32_t* X, const int32_t* Y, const float* z, uint32_t id, int W, int H, int rx, int ry, uint64_t* keys,
                  uint32_t* n_frag, int* was_medium) {
    CheckSink sink(keys, W, H, rx, ry);
    const bool medium = spans_fit_int32(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
    *was_medium = medium ? 1 : 0;
    for (uint32_t lane = 0; lane < 64; ++lane) {
        if (medium)
            big_medium_lane(X, Y, z, id, W, H, rx, ry, lane, [&](const uint32_t pix[4], const uint64_t key[4], const int32_t py[4]) {
                for (int k = 0; k < 4; ++k)
                    if (key[k] != kVisClear) {      // the contract: pix[k], py[k] are only meaningful with a key
                        if (py[k] < 0 || py[k] >= H || (size_t)py[k] != pix[k] / (size_t)W) ++sink.violations;      // the row the kernel marks
                        sink(pix[k], key[k]);
                    }
            });
        else
            big_giant_lane(X, Y, z, id, W, H, rx, ry, lane, [&](size_t pix, uint64_t key, int32_t py) {
                if (py < 0 || py >= H || (size_t)py != pix / (size_t)W) ++sink.violations;
                sink(pix, key);
            });
    }
    *n_frag = sink.n;
    return sink.violations;
}

int emul_raster_rows(const int32_t* X, const int32_t* Y, const float* z, uint32_t id, int W, int H, uint64_t* keys, uint32_t* n_frag) {
    CheckSink sink(keys, W, H, -1, -1);
    if (!spans_fit_int32(X[0], Y[0], X[1], Y[1], X[2], Y[2])) return -1;
    const int32_t area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
    if (area2 < 0)      // classify_small only lists front-facing triangles
        raster_rows(W, H, X[0], Y[0], X[1], Y[1], X[2], Y[2], z[0], z[1], z[2], id, [&](uint32_t pix, uint64_t key) { sink(pix, key); });
    *n_frag = sink.n;
    return sink.violations;
}

// the same triangle through triangle_setup / triangle_pixel (the generic path the oracle-parity tests pin); rx < 0: no region
void emul_reference_triangle(const int32_t* X, const int32_t* Y, const float* z, uint32_t id, int W, int H, int rx, int ry, uint64_t* keys,
                             uint32_t* n_frag) {
    SVert s[3];
    for (int k = 0; k < 3; ++k) { s[k].X = X[k]; s[k].Y = Y[k]; s[k].z = z[k]; s[k].flag = kVtxOk; }
    TriSetup ts;
    *n_frag = 0;
    if (!triangle_setup(s[0], s[1], s[2], W, H, ts)) return;
    for (int py = ts.py0; py <= ts.py1; ++py) for (int px = ts.px0; px <= ts.px1; ++px) {
        if (rx >= 0 && (px < rx * 64 || px > rx * 64 + 63 || py < ry * 64 || py > ry * 64 + 63)) continue;
        float zz, b[3];
        if (triangle_pixel(ts, px, py, zz, b)) {
            const uint64_t key = vis_key(zz, id);
            if (key < keys[(size_t)py * W + px]) keys[(size_t)py * W + px] = key;
            ++*n_frag;
        }
    }
}

// the overlay pass through the product's header functions, sequentially: keys raised by max (as k_overlay_raster's atomic
// does), then coloured (k_overlay_resolve); srgb != 0: *Srgb target
void emul_overlay_lines(const void* vertices, uint32_t n_vertices, const uint32_t* idx, uint32_t n_indices, float width, int W, int H, int srgb,
                        int bgra, uint8_t* rgba, size_t pitch) {
    const OverlayVertex* vs = (const OverlayVertex*)vertices;
    std::vector<uint64_t> keys((size_t)W * H, kOverlayClear);
    for (uint32_t t = 0; 3 * t + 2 < n_indices; ++t) {
        const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
        if (i0 >= n_vertices || i1 >= n_vertices || i2 >= n_vertices) continue;
        SVert s0, s1, s2;
        if (overlay_vertex(vs[i0], width, (float)W, (float)H, s0) != kVtxOk || overlay_vertex(vs[i1], width, (float)W, (float)H, s1) != kVtxOk ||
            overlay_vertex(vs[i2], width, (float)W, (float)H, s2) != kVtxOk)
            continue;
        TriSetup ts;
        if (!triangle_setup(s0, s1, s2, W, H, ts)) continue;
        for (int py = ts.py0; py <= ts.py1; ++py) for (int px = ts.px0; px <= ts.px1; ++px) {
            const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
            int64_t F[3]; bool in = true;
            for (int e = 0; e < 3; ++e) { F[e] = ts.dy[e] * (cx - ts.ax[e]) - ts.dx[e] * (cy - ts.ay[e]); in = in && F[e] + ts.bias[e] >= 0; }
            if (!in) continue;
            const float z = fmaf((float)F[1] * ts.iA, ts.dz1, fmaf((float)F[2] * ts.iA, ts.dz2, ts.z0));
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            const uint64_t key = overlay_key(z, t);
            if (key > keys[(size_t)py * W + px]) keys[(size_t)py * W + px] = key;
        }
    }
    float thresh[256];
    for (int i = 0; i < 256; ++i) thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]);
    for (int py = 0; py < H; ++py) for (int px = 0; px < W; ++px) {
        const uint64_t key = keys[(size_t)py * W + px];
        if (key <= kOverlayClear) continue;
        const uint32_t t = 0xFFFFFFFFu - (uint32_t)key;
        float rgb[3];
        if (!overlay_color(vs[idx[3 * t]], vs[idx[3 * t + 1]], vs[idx[3 * t + 2]], width, W, H, px, py, rgb)) continue;
        uint32_t out = srgb ? srgb_encode(thresh, rgb[0]) | (srgb_encode(thresh, rgb[1]) << 8) | (srgb_encode(thresh, rgb[2]) << 16)
                            : to_unorm8(rgb[0]) | (to_unorm8(rgb[1]) << 8) | (to_unorm8(rgb[2]) << 16);
        out |= to_unorm8(1.0f) << 24;
        if (bgra) out = (out & 0xFF00FF00u) | ((out >> 16) & 0xFFu) | ((out & 0xFFu) << 16);
        memcpy(rgba + (size_t)py * pitch + (size_t)px * 4, &out, 4);
    }
}

float emul_from_unorm8(uint32_t c) { return from_unorm8(c); }

uint64_t emul_vis_key(float z, uint32_t id) { return vis_key(z, id); }

void emul_srgb_tables(float* decode, float* thresh) {
    for (int i = 0; i < 256; ++i) decode[i] = bits_f(TOPO_SRGB_DECODE_BITS[i]);
    for (int i = 0; i < 255; ++i) thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]);
}


// the text overlay through the product's header function glyph_blend, sequentially, with the kernels' ownership rule (the
// first quad over a pixel keeps it: max of depth << 32 | ~index at one depth per call)
void emul_overlay_glyphs(const void* glyphs, uint32_t n_glyphs, const uint8_t* atlas, uint32_t aw, uint32_t ah, int W, int H, int srgb, int bgra,
                         uint8_t* rgba, size_t pitch) {
    const GlyphInstance* gs = (const GlyphInstance*)glyphs;
    float thresh[256], decode[256];
    for (int i = 0; i < 256; ++i) { thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]); decode[i] = bits_f(TOPO_SRGB_DECODE_BITS[i]); }
    std::vector<uint32_t> owner((size_t)W * H, 0xFFFFFFFFu);
    for (uint32_t g = 0; g < n_glyphs; ++g)
        for (uint32_t dy = 0; dy < gs[g].dim[1]; ++dy) for (uint32_t dx = 0; dx < gs[g].dim[0]; ++dx) {
            const int64_t px = (int64_t)gs[g].pos[0] + dx, py = (int64_t)gs[g].pos[1] + dy;
            if (px < 0 || py < 0 || px >= W || py >= H) continue;
            uint32_t& o = owner[(size_t)py * W + px];
            if (o != 0xFFFFFFFFu) continue;
            o = g;
            const uint32_t ax = gs[g].uv[0] + dx, ay = gs[g].uv[1] + dy;
            const uint32_t mask = ax < aw && ay < ah ? atlas[(size_t)ay * aw + ax] : 0u;
            uint32_t* out = (uint32_t*)(rgba + (size_t)py * pitch + (size_t)px * 4);
            *out = glyph_blend(gs[g], mask, *out, srgb != 0, bgra != 0, thresh, decode);
        }
}
}  // extern "C"
