// terrain_renderer.cpp -- host orchestration of the HIP terrain path (see terrain_renderer.hpp).
#include "terrain_renderer.hpp"

#include "geotiff.hpp"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace topo {

#define TOPO_HIP_TRY(expr)                                   \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);    \
    } while (0)

int TerrainRenderer::fail(int code, const std::string& msg) {
    err_ = msg;
    return code;
}

int TerrainRenderer::hip_fail(hipError_t e, const char* what) {
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return TOPO_ERR_HIP;
}

int TerrainRenderer::bind_device() {
    TOPO_HIP_TRY(hipSetDevice(device_));
    return TOPO_OK;
}

// TerrainRenderer::new (terrain_renderer.rs:37-69): targets are allocated lazily at the first render.
int TerrainRenderer::create(TerrainRenderer** out, int device, uint32_t w, uint32_t h, uint32_t format, std::string* err) {
    *out = nullptr;
    if (w == 0 || h == 0) { *err = "target size must be non-zero"; return TOPO_ERR_INVALID; }
    if (format < TOPO_FORMAT_RGBA8_UNORM_SRGB || format > TOPO_FORMAT_BGRA8_UNORM) {
        *err = "colour format must be Rgba8UnormSrgb, Bgra8UnormSrgb, Rgba8Unorm or Bgra8Unorm";
        return TOPO_ERR_UNSUPPORTED;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        *err = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
               " (this library has no CPU fallback)";
        return TOPO_ERR_HIP;
    }
    if (device < 0 || device >= count) { *err = "hip_device out of range"; return TOPO_ERR_INVALID; }
    TerrainRenderer* r = new TerrainRenderer();
    r->device_ = device;
    r->W_ = w;
    r->H_ = h;
    r->format_ = format;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreate(&r->own_stream_);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&r->load_ev_[i]);
    for (int k = 0; k < kEvRing && e == hipSuccess; ++k)
        for (int i = 0; i < kNumEvents && e == hipSuccess; ++i) e = hipEventCreate(&r->ctx_[0].evr[k][i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ctx_[0].done, hipEventDisableTiming);
    if (e != hipSuccess) {
        *err = std::string("HIP initialisation failed: ") + hipGetErrorString(e);
        delete r;
        return TOPO_ERR_HIP;
    }
    r->stream_ = r->own_stream_;
    *out = r;
    return TOPO_OK;
}

TerrainRenderer::~TerrainRenderer() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& c : ctx_)
        if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (auto& kv : tiles_) {
        (void)hipFree(kv.second.d_pool);
    }
    void* bufs[] = {d_tiles_, d_views_, d_out_rgba_, d_out_depth_, d_pre_rgba_, d_pre_depth_, d_edge_jobs_, d_corner_jobs_, d_peaks_, d_proj_, d_overlay_geo_, d_overlay_keys_};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    for (auto& c : ctx_) {
        void* cb[] = {c.d_vis, c.d_dirty, c.d_work, c.d_work2, c.d_far, c.d_big, c.d_rare, c.d_counters};
        for (void* p : cb)
            if (p) (void)hipFree(p);
        for (auto& set : c.evr)
            for (auto& e : set)
                if (e) (void)hipEventDestroy(e);
        if (c.done) (void)hipEventDestroy(c.done);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        if (c.h_status) (void)hipHostFree(c.h_status);
    }
    for (auto& e : load_ev_)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : view_ev_)
        if (e) (void)hipEventDestroy(e);
    if (h_views_) (void)hipHostFree(h_views_);
    if (h_stage_) (void)hipHostFree(h_stage_);
    for (auto& e : stage_ev_)
        if (e) (void)hipEventDestroy(e);
    for (const auto& pin : pinned_) (void)hipHostUnregister(pin.first);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

int TerrainRenderer::ensure(void** p, size_t* cap, size_t need) { return ensure_on(stream_, p, cap, need); }

int TerrainRenderer::ensure_on(hipStream_t s, void** p, size_t* cap, size_t need) {
    if (need <= *cap) return TOPO_OK;
    if (*p) {
        TOPO_HIP_TRY(hipStreamSynchronize(s));
        TOPO_HIP_TRY(hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    TOPO_HIP_TRY(hipMalloc(p, need));
    *cap = need;
    return TOPO_OK;
}

Tile* TerrainRenderer::find(int lat, int lon) {
    auto it = tiles_.find(geo_key(lat, lon));
    return it == tiles_.end() ? nullptr : &it->second;
}

// The compute-pass orchestration of add_terrain (terrain_renderer.rs:192-347) for tile `nt`: a seam pass for each
// loaded edge neighbour and a corner pass for each complete 2x2 block, every one of them fed the NEW tile's
// uniforms (:275, :344).  "Loaded" = inserted before `nt` (all other tiles, except while replaying).  Jobs name
// tiles by rank (their index in the device tile table).
void TerrainRenderer::collect_jobs(const Tile& nt, const std::map<GeoKey, uint32_t>& rank, std::vector<EdgeJob>& edges,
                                   std::vector<CornerJob>& corners) {
    const int lat = nt.lat, lon = nt.lon;
    const uint32_t NONE = 0xFFFFFFFFu;
    auto loaded = [&](int la, int lo) -> uint32_t {
        Tile* t = find(la, lo);
        return (t && t != &nt && t->seq < nt.seq) ? rank.at(geo_key(la, lo)) : NONE;
    };
    const uint32_t me = rank.at(geo_key(lat, lon));
    const uint32_t left = loaded(lat, lon - 1), right = loaded(lat, lon + 1);
    const uint32_t top = loaded(lat + 1, lon), bottom = loaded(lat - 1, lon);
    const uint32_t top_left = loaded(lat + 1, lon - 1), top_right = loaded(lat + 1, lon + 1);
    const uint32_t bottom_left = loaded(lat - 1, lon - 1), bottom_right = loaded(lat - 1, lon + 1);
    if (left != NONE) edges.push_back(EdgeJob{left, me, me, 0});
    if (right != NONE) edges.push_back(EdgeJob{me, right, me, 0});
    if (top != NONE) edges.push_back(EdgeJob{top, me, me, 1});
    if (bottom != NONE) edges.push_back(EdgeJob{me, bottom, me, 1});
    if (top_left != NONE && top != NONE && left != NONE) corners.push_back(CornerJob{top_left, top, left, me, me});
    if (top != NONE && top_right != NONE && right != NONE) corners.push_back(CornerJob{top, top_right, me, right, me});
    if (left != NONE && bottom_left != NONE && bottom != NONE) corners.push_back(CornerJob{left, me, bottom_left, bottom, me});
    if (right != NONE && bottom != NONE && bottom_right != NONE) corners.push_back(CornerJob{me, right, bottom, bottom_right, me});
}

// Uploads the job lists and launches the seam and corner passes (each writes a disjoint set of texels, so one
// launch per kind covers any number of jobs).
int TerrainRenderer::upload_seam_jobs(const std::vector<EdgeJob>& edges, const std::vector<CornerJob>& corners) {
    if (!edges.empty()) {
        if (int rc = ensure(&d_edge_jobs_, &cap_edge_jobs_, edges.size() * sizeof(EdgeJob))) return rc;
        TOPO_HIP_TRY(hipMemcpyAsync(d_edge_jobs_, edges.data(), edges.size() * sizeof(EdgeJob), hipMemcpyHostToDevice, stream_));
    }
    if (!corners.empty()) {
        if (int rc = ensure(&d_corner_jobs_, &cap_corner_jobs_, corners.size() * sizeof(CornerJob))) return rc;
        TOPO_HIP_TRY(hipMemcpyAsync(d_corner_jobs_, corners.data(), corners.size() * sizeof(CornerJob), hipMemcpyHostToDevice, stream_));
    }
    // the job vectors are pageable host memory: hipMemcpyAsync has staged them before returning
    return TOPO_OK;
}
void TerrainRenderer::launch_seam_jobs(size_t n_edges, size_t n_corners) {
    launch_normals_border((const TileDev*)d_tiles_, (const EdgeJob*)d_edge_jobs_, (uint32_t)n_edges, (const CornerJob*)d_corner_jobs_, (uint32_t)n_corners,
                          tile_w_, tile_h_, stream_);
}
int TerrainRenderer::run_seam_jobs(const std::vector<EdgeJob>& edges, const std::vector<CornerJob>& corners) {
    if (int rc = upload_seam_jobs(edges, corners)) return rc;
    launch_seam_jobs(edges.size(), corners.size());
    return TOPO_OK;
}

std::map<GeoKey, uint32_t> TerrainRenderer::ranks() const {
    std::map<GeoKey, uint32_t> r;
    uint32_t i = 0;
    for (const auto& kv : tiles_) r[kv.first] = i++;
    return r;
}

int TerrainRenderer::add_terrain(int32_t lat, int32_t lon, const float* heights, bool on_device, uint32_t w, uint32_t h,
                                 const float rp[2], const float mp[2], const float ps[2]) {
    if (!heights || !rp || !mp || !ps) return fail(TOPO_ERR_INVALID, "null argument");
    if (w < 3 || h < 3) return fail(TOPO_ERR_INVALID, "tile must be at least 3x3");
    if (w > 32768 || h > 32768) return fail(TOPO_ERR_INVALID, "tile too large");
    if (!tiles_.empty() && (w != tile_w_ || h != tile_h_))
        return fail(TOPO_ERR_INVALID, "mixed tile sizes are rejected (the reference caches one mesh: render_buffer.rs:12-15)");
    if (int rc = join()) return rc;
    const uint64_t tris = 2ull * (w - 1) * (h - 1);
    const size_t n_after = tiles_.size() + (find(lat, lon) ? 0 : 1);
    if (tris * n_after >= (1ull << 31) || n_after > 0xFFFFu) return fail(TOPO_ERR_CAPACITY, "draw-order id space exhausted");
    tile_w_ = w;
    tile_h_ = h;
    const size_t texels = (size_t)w * h;
    const uint32_t bxc = (w - 1 + kBCX - 1) / kBCX, byc = (h - 1 + kBCY - 1) / kBCY;
    Tile t;
    t.lat = lat;
    t.lon = lon;
    t.seq = next_seq_++;
    // ONE allocation per tile: heights, normals, then block min/max (2 floats per block), the sin/cos tables of the w columns
    // and the h rows, and the f64 cull bounds (sphere 4 + corners 12 + sagitta 1 doubles per block); every part 256-byte aligned
    const size_t tile_floats = (size_t)bxc * byc * 2 + 2 * ((size_t)w + h);
    auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t off_normals = up256(texels * 4), off_tables = off_normals + up256(texels * 4), off_bounds = off_tables + up256(tile_floats * sizeof(float));
    TOPO_HIP_TRY(hipMalloc(&t.d_pool, off_bounds + (size_t)bxc * byc * 17 * sizeof(double)));
    t.d_heights = reinterpret_cast<float*>(t.d_pool);
    t.d_normals = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(t.d_pool) + off_normals);
    t.d_minmax = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(t.d_pool) + off_tables);
    const hipError_t e = hipMemcpyAsync(t.d_heights, heights, texels * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_);
    // (the zero-initialised normal texture: k_normals_interior writes the untouched border ring as zero)
    if (e != hipSuccess) {
        (void)hipFree(t.d_pool);
        return hip_fail(e, "tile upload");
    }
    // TerrainUniforms::new (render/data.rs:124-151)
    t.dev.heights = t.d_heights;
    t.dev.normals = t.d_normals;
    t.dev.block_minmax = t.d_minmax;
    t.dev.trig_lon = t.d_minmax + (size_t)bxc * byc * 2;
    t.dev.trig_lat = t.dev.trig_lon + 2 * (size_t)w;
    t.dev.block_bounds = reinterpret_cast<const double*>(reinterpret_cast<uint8_t*>(t.d_pool) + off_bounds);
    t.dev.raster_x = rp[0]; t.dev.raster_y = rp[1];
    t.dev.model_x = mp[0]; t.dev.model_y = mp[1];
    t.dev.scale_x = ps[0]; t.dev.scale_y = ps[1];
    terrain_rotation(mp[0], mp[1], t.dev.rot);
    // BTreeMap::insert replaces an existing entry; its GPU resources are dropped after the passes below
    Tile old{};
    bool had_old = false;
    if (Tile* ex = find(lat, lon)) { old = *ex; had_old = true; tiles_.erase(geo_key(lat, lon)); }
    Tile& nt = tiles_[geo_key(lat, lon)] = t;
    table_dirty_ = true;
    if (int rc = upload_tile_table()) return rc;
    {
        const std::map<GeoKey, uint32_t> rk = ranks();
        std::vector<EdgeJob> edges;
        std::vector<CornerJob> corners;
        collect_jobs(nt, rk, edges, corners);
        launch_load_kernels(rk.at(geo_key(lat, lon)), 1, nullptr);
        if (int rc = run_seam_jobs(edges, corners)) return rc;
    }
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));   // `heights` (and the job lists) are only borrowed for the call
    {   // the sphere around the block spheres' centres (see Tile::centres); any non-finite entry leaves it unknown
        std::vector<double> sph((size_t)bxc * byc * 4);
        TOPO_HIP_TRY(hipMemcpy(sph.data(), nt.dev.block_bounds, sph.size() * sizeof(double), hipMemcpyDeviceToHost));
        double c[3] = {0.0, 0.0, 0.0}, r2 = 0.0;
        bool finite = true;
        for (size_t i = 0; i < sph.size(); i += 4) {
            finite = finite && std::isfinite(sph[i]) && std::isfinite(sph[i + 1]) && std::isfinite(sph[i + 2]) && std::isfinite(sph[i + 3]);
            for (int k = 0; k < 3; ++k) c[k] += sph[i + k];
        }
        for (int k = 0; k < 3; ++k) c[k] /= (double)(sph.size() / 4);
        for (size_t i = 0; i < sph.size() && finite; i += 4) {
            const double dx = sph[i] - c[0], dy = sph[i + 1] - c[1], dz = sph[i + 2] - c[2];
            r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
        }
        nt.centres[0] = c[0]; nt.centres[1] = c[1]; nt.centres[2] = c[2];
        nt.centres[3] = finite ? std::sqrt(r2) : -1.0;
    }
    if (had_old) {
        TOPO_HIP_TRY(hipStreamSynchronize(stream_));
        (void)hipFree(old.d_pool);
    }
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

// Every load-time kernel of tiles [first, first + count) of the device table except the seam passes.  Two forms: the DEM read
// ONCE (k_trig_tables -> normals + block minima / maxima in one pass -> k_block_bounds; COP90 / COP30 widths) or the tables kernel
// followed by the normals kernel (any size, any LDS tile size).  `mid`: recorded between the part that precedes the normals and
// the rest (the load phase's timing bracket).
void TerrainRenderer::launch_load_kernels(uint32_t first, uint32_t count, hipEvent_t mid) {
    const TileDev* tiles = (const TileDev*)d_tiles_;
    if (normals_tables_fused(tile_w_, tile_h_, lds_rows_)) {
        launch_trig_tables(tiles, first, count, tile_w_, tile_h_, stream_);
        if (mid) (void)hipEventRecord(mid, stream_);
        launch_normals_tables(tiles, first, count, tile_w_, tile_h_, stream_);
        launch_block_bounds(tiles, first, count, tile_w_, tile_h_, stream_);
    } else {
        launch_block_tables(tiles, first, count, tile_w_, tile_h_, stream_);
        if (mid) (void)hipEventRecord(mid, stream_);
        launch_normals_interior(tiles, first, count, tile_w_, tile_h_, lds_rows_, stream_);
    }
}

// unload_terrain (terrain_renderer.rs:361-363): neighbours keep whatever seam normals they have.
int TerrainRenderer::unload_terrain(int32_t lat, int32_t lon) {
    Tile* t = find(lat, lon);
    if (!t) return TOPO_OK;   // BTreeMap::remove of a missing key is a no-op
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    (void)hipFree(t->d_pool);
    tiles_.erase(geo_key(lat, lon));
    table_dirty_ = true;
    return TOPO_OK;
}

int TerrainRenderer::recompute_normals() {
    if (int rc = join()) return rc;
    std::vector<Tile*> order;
    for (auto& kv : tiles_) order.push_back(&kv.second);
    std::sort(order.begin(), order.end(), [](Tile* a, Tile* b) { return a->seq < b->seq; });
    if (int rc = upload_tile_table()) return rc;
    const std::map<GeoKey, uint32_t> rk = ranks();
    std::vector<EdgeJob> edges;
    std::vector<CornerJob> corners;
    for (Tile* t : order) collect_jobs(*t, rk, edges, corners);
    if (int rc = ensure(&d_edge_jobs_, &cap_edge_jobs_, (edges.size() + 1) * sizeof(EdgeJob))) return rc;
    if (int rc = ensure(&d_corner_jobs_, &cap_corner_jobs_, (corners.size() + 1) * sizeof(CornerJob))) return rc;
    if (int rc = upload_seam_jobs(edges, corners)) return rc;      // (the job lists: host -> device, ahead of the kernels that are timed)
    // the whole load phase of the resident tiles, every load-time kernel inside the bracket: the tables of the frame phase
    // (ev 0 -> 2), then the normals K1-K3 (ev 2 -> 1)
    TOPO_HIP_TRY(hipEventRecord(load_ev_[0], stream_));
    launch_load_kernels(0, (uint32_t)order.size(), load_ev_[2]);
    launch_seam_jobs(edges.size(), corners.size());
    TOPO_HIP_TRY(hipEventRecord(load_ev_[1], stream_));
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));   // the job lists are locals
    load_timed_ = true;
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

// update (terrain_renderer.rs:151-171)
int TerrainRenderer::update(uint32_t w, uint32_t h, const topo_uniforms* u, const topo_post_uniforms* pu) {
    if (!u || !pu) return fail(TOPO_ERR_INVALID, "null argument");
    if (w == 0 || h == 0) return fail(TOPO_ERR_INVALID, "target size must be non-zero");
    if (pu->pixelize_n < 99.99999f && !(pu->pixelize_n >= 1.0f)) return fail(TOPO_ERR_INVALID, "pixelize_n must be at least 1");
    if (pu->pixelize_n < 99.99999f && !(pu->viewport[0] >= 1.0f && pu->viewport[1] >= 1.0f)) return fail(TOPO_ERR_INVALID, "viewport must be at least 1 x 1");
    W_ = w;
    H_ = h;
    uniforms_ = *u;
    post_ = *pu;
    have_uniforms_ = true;
    return TOPO_OK;
}

int TerrainRenderer::upload_tile_table() {
    if (!table_dirty_) return TOPO_OK;
    std::vector<TileDev> table;
    for (auto& kv : tiles_) table.push_back(kv.second.dev);   // std::map iterates in BTreeMap order
    if (!table.empty()) {
        if (int rc = ensure(&d_tiles_, &cap_tiles_, table.size() * sizeof(TileDev))) return rc;
        TOPO_HIP_TRY(hipStreamSynchronize(stream_));
        TOPO_HIP_TRY(hipMemcpy(d_tiles_, table.data(), table.size() * sizeof(TileDev), hipMemcpyHostToDevice));
    }
    table_dirty_ = false;
    return TOPO_OK;
}

int TerrainRenderer::init_ctx(FrameCtx& c, bool own_stream) {
    if (!c.done) {
        for (auto& set : c.evr)
            for (auto& e : set)
                if (!e) TOPO_HIP_TRY(hipEventCreate(&e));
        TOPO_HIP_TRY(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
    }
    if (own_stream && !c.stream) TOPO_HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    return TOPO_OK;
}

// Frames in flight on the contexts' own streams are not ordered with stream_: everything that frees or rewrites what a
// frame reads (tiles, the tile table), and every consumer of a frame's outputs, joins them first.
int TerrainRenderer::join() {
    if (int rc = bind_device()) return rc;
    for (auto& c : ctx_)
        if (c.pending && c.stream) {
            TOPO_HIP_TRY(hipStreamSynchronize(c.stream));
            c.pending = false;
        }
    return TOPO_OK;
}

// The status word of a frame is per frame (k_clear resets it, render_frame copies the counters to pinned memory behind
// k_resolve).  Called once the frames' streams have been waited for: a frame whose rare-triangle queue overflowed has
// dropped triangles -- its outputs are incomplete -- and that is an error of the call that waited for it
// (topo_join / topo_synchronize / topo_render / topo_get_counters), reported once per frame.
bool TerrainRenderer::fold_frames(FrameCtx& c) {
    bool overflow = false;
    for (; c.checked < c.submitted; ++c.checked) {
        const uint32_t* w = c.h_status + (c.checked % kStatusRing) * 16;
        // status bits accumulate over the frames folded since the last topo_frame_status (which clears them): a burst of frames
        // cannot hide an earlier frame's overflow or bounds violation behind a clean last frame
        last_status_[0] |= w[2];
        if (w[2] & kStatusBounds) { last_status_[1] = w[8]; last_status_[2] = w[9]; last_status_[3] = w[10]; }
        overflow |= (w[2] & kStatusRareOverflow) != 0;
    }
    return overflow;
}

int TerrainRenderer::check_frames() {
    bool overflow = overflow_pending_;
    overflow_pending_ = false;
    for (auto& c : ctx_)
        if (!c.pending && c.h_status) overflow |= fold_frames(c);
    if (overflow) return fail(TOPO_ERR_CAPACITY, "rare-triangle queue overflowed: a frame is incomplete (raise the queue capacity or render fewer views per submission)");
    return TOPO_OK;
}

int TerrainRenderer::set_pipeline_depth(int depth) {
    if (depth < 1 || depth > kMaxPipeline) return fail(TOPO_ERR_INVALID, "pipeline depth must be 1..4");
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    for (int i = 0; i < depth; ++i)
        if (int rc = init_ctx(ctx_[i], depth > 1)) return rc;
    pipeline_depth_ = depth;
    next_ctx_ = last_ctx_ = 0;
    return TOPO_OK;
}

int TerrainRenderer::render_views_device(uint32_t n, const topo_uniforms* views, uint32_t w, uint32_t h, const OutputParams& out) {
    if (n == 0 || !views || !out.rgba) return fail(TOPO_ERR_INVALID, "null/empty argument");
    if (n > 0xFFFFu) return fail(TOPO_ERR_INVALID, "too many views");
    if (w == 0 || h == 0 || w > 65536 || h > 65536) return fail(TOPO_ERR_INVALID, "bad target size");
    if (int rc = bind_device()) return rc;
    if (table_dirty_)
        if (int rc = join()) return rc;
    if (int rc = upload_tile_table()) return rc;
    FrameCtx& c = ctx_[next_ctx_];
    last_ctx_ = next_ctx_;
    next_ctx_ = (next_ctx_ + 1) % pipeline_depth_;
    if (pipeline_depth_ == 1) return render_frame(c, stream_, n, views, w, h, out);
    // the tile table (and whatever else the caller queued) was produced on stream_: order the frame after it
    TOPO_HIP_TRY(hipEventRecord(c.done, stream_));
    TOPO_HIP_TRY(hipStreamWaitEvent(c.stream, c.done, 0));
    c.pending = true;
    return render_frame(c, c.stream, n, views, w, h, out);
}

int TerrainRenderer::render_frame(FrameCtx& c, hipStream_t stream, uint32_t n, const topo_uniforms* views, uint32_t w, uint32_t h,
                                  const OutputParams& out, const ResolveSlot* slots, uint32_t n_slots,
                                  const std::function<int(uint32_t, hipStream_t)>* after_slot) {
    const uint32_t n_tiles = (uint32_t)tiles_.size();
    const uint32_t bxc = n_tiles ? (tile_w_ - 1 + kBCX - 1) / kBCX : 0, byc = n_tiles ? (tile_h_ - 1 + kBCY - 1) / kBCY : 0;
    const size_t pixels = (size_t)n * w * h;
    const size_t work_cap = (size_t)n * n_tiles * bxc * byc;
    const size_t big_cap = big_cap_cfg_ ? big_cap_cfg_ : (1u << 22);
    const size_t rare_cap = rare_cap_cfg_ ? rare_cap_cfg_ : (rare_cap_auto_ ? (size_t)rare_cap_auto_ : (1u << 22));
    if (work_cap >= (1ull << 30)) return fail(TOPO_ERR_CAPACITY, "too many raster blocks in one submission");
    if (pixels >= (1ull << 32)) return fail(TOPO_ERR_CAPACITY, "more than 2^32 pixels in one submission");
    const size_t vis_keys = (pixels + 63) & ~(size_t)63;   // whole 64-key segments: k_clear rewrites segments, not keys
    if (vis_keys * 8 > c.cap_vis) {
        // a fresh buffer holds garbage: mark every segment so that the first k_clear initialises all of it
        if (int rc = ensure_on(stream, &c.d_vis, &c.cap_vis, vis_keys * 8)) return rc;
        if (int rc = ensure_on(stream, &c.d_dirty, &c.cap_dirty, vis_keys / 64 + 64)) return rc;
        TOPO_HIP_TRY(hipMemsetAsync(c.d_dirty, 1, c.cap_dirty, stream));
    }
    if (int rc = ensure(&d_views_, &cap_views_, sizeof(ViewDev) * kMaxViewsPerSlot * kViewSlots)) return rc;
    // A near block (and a far survivor) is cut into strips of near_strip cell rows, one wave each: the raster phase is as long
    // as its longest strip, and a strip's vertex rows cost less than the pixels of its triangles.  Measured (tools/exp_strip.sh,
    // ms per frame at 1 / 2 / 4 rows): c1 0.146 / 0.148 / 0.157, c2 0.186 / 0.184 / 0.194, c3 0.382 / 0.372 / 0.390,
    // c4 0.964 / 0.933 / 0.942.  TOPO_NEAR_STRIP overrides (experiments).
    static const int strip_env = getenv("TOPO_NEAR_STRIP") ? atoi(getenv("TOPO_NEAR_STRIP")) : 0;
    const uint32_t near_strip = strip_env >= 1 && strip_env <= 15 ? (uint32_t)strip_env : (work_cap <= 64 * 1024 ? 1u : 2u);
    const size_t near_cap = (size_t)((kBCY + near_strip - 1) / near_strip) * work_cap;
    if (int rc = ensure_on(stream, &c.d_work, &c.cap_work, (near_cap ? near_cap : 1) * sizeof(WorkItem))) return rc;
    if (int rc = ensure_on(stream, &c.d_work2, &c.cap_work2, (near_cap ? near_cap : 1) * sizeof(WorkItem))) return rc;   // far survivors, in strips too
    // the far-candidate list: kFarLists sub-lists, cull workgroup b (256 blocks) appending to sub-list b % kFarLists
    const size_t far_sub_cap = ((work_cap + 255) / 256 + kFarLists - 1) / kFarLists * 256;
    if (int rc = ensure_on(stream, &c.d_far, &c.cap_far, (far_sub_cap ? far_sub_cap * kFarLists : 1) * sizeof(FarItem))) return rc;
    if (int rc = ensure_on(stream, &c.d_big, &c.cap_big, big_cap * sizeof(BigItem))) return rc;
    if (int rc = ensure_on(stream, &c.d_rare, &c.cap_rare, rare_cap * sizeof(RareItem))) return rc;
    if (!c.d_counters) {
        if (int rc = ensure_on(stream, &c.d_counters, &c.cap_counters, 2 * kCounterWords * sizeof(uint32_t))) return rc;      // two sets, alternating
        TOPO_HIP_TRY(hipMemsetAsync(c.d_counters, 0, 2 * kCounterWords * sizeof(uint32_t), stream));
    }
    if (!c.h_status) TOPO_HIP_TRY(hipHostMalloc((void**)&c.h_status, kStatusRing * 16 * sizeof(uint32_t)));
    if (c.submitted - c.checked == kStatusRing) {      // nobody has waited for this context's frames for a whole ring: fold them now
        TOPO_HIP_TRY(hipStreamSynchronize(stream));
        overflow_pending_ |= fold_frames(c);
    }
    // View constants.  Up to kPackViews views (a panorama's eight sectors) travel as the argument of a one-workgroup kernel
    // (k_put_views: the launch copies them; a copy-engine operation and the event guarding its pinned source cost the GPU 13 us
    // per frame and the host two more calls).  Larger submissions go through a small ring of pinned staging slots, each guarded
    // by an event, so a submission never has to wait for the stream (pageable sources would force a synchronous staging copy).
    // Either way each slot has its own device copy, so a later submission cannot overwrite constants a running frame reads.
    if (n > kMaxViewsPerSlot) return fail(TOPO_ERR_INVALID, "too many views in one submission");
    static const bool pack_off = getenv("TOPO_VIEWS_BY_COPY") && atoi(getenv("TOPO_VIEWS_BY_COPY")) != 0;
    static_assert(kViewSlots % kMaxPipeline == 0, "every frame context has its own share of the slots");
    // a slot belongs to one frame context, so whatever used it before is ahead of this submission in the same stream
    const int slot = (int)(&c - ctx_) * (kViewSlots / kMaxPipeline) + (int)(c.frames % (kViewSlots / kMaxPipeline));
    ViewDev* d_slot = (ViewDev*)d_views_ + (size_t)slot * kMaxViewsPerSlot;
    auto fill_views = [&](ViewDev* vd) {
        for (uint32_t i = 0; i < n; ++i) {
            memcpy(vd[i].proj, views[i].camera_proj, sizeof vd[i].proj);
            vd[i].cam_x = views[i].camera_pos[0];
            vd[i].cam_y = views[i].camera_pos[1];
            memcpy(vd[i].sun, views[i].sun_direction, sizeof vd[i].sun);
            vd[i].view_mode = views[i].view_mode;
        }
    };
    // (with the frame's first kernel being k_clear_cull -- there are tiles to cull -- the pack rides in THAT launch's argument
    // segment: no upload kernel either; TOPO_VIEWS_IN_CULL=0: always k_put_views)
    static const bool fuse_off = getenv("TOPO_FUSE_CLEAR_CULL") && atoi(getenv("TOPO_FUSE_CLEAR_CULL")) == 0;
    static const bool in_cull_off = getenv("TOPO_VIEWS_IN_CULL") && atoi(getenv("TOPO_VIEWS_IN_CULL")) == 0;
    const bool fuse = !fuse_off && n_tiles != 0;
    ViewPack pack{};
    const bool packed = n <= kPackViews && !pack_off, pack_in_cull = packed && fuse && !in_cull_off;
    if (packed) {
        fill_views(pack.v);
        if (!pack_in_cull) launch_put_views(pack, n, d_slot, stream);
    } else {
        if (!h_views_) {
            TOPO_HIP_TRY(hipHostMalloc((void**)&h_views_, sizeof(ViewDev) * kMaxViewsPerSlot * kViewSlots));
            for (int i = 0; i < kViewSlots; ++i) TOPO_HIP_TRY(hipEventCreateWithFlags(&view_ev_[i], hipEventDisableTiming));
        }
        if (view_used_[slot]) TOPO_HIP_TRY(hipEventSynchronize(view_ev_[slot]));
        ViewDev* vd = h_views_ + (size_t)slot * kMaxViewsPerSlot;
        fill_views(vd);
        TOPO_HIP_TRY(hipMemcpyAsync(d_slot, vd, n * sizeof(ViewDev), hipMemcpyHostToDevice, stream));
        TOPO_HIP_TRY(hipEventRecord(view_ev_[slot], stream));
        view_used_[slot] = true;
    }

    FrameParams p{};
    p.tiles = (const TileDev*)d_tiles_;
    p.views = d_slot;
    p.vis = (uint64_t*)c.d_vis;
    p.dirty = (uint8_t*)c.d_dirty;
    p.work = (WorkItem*)c.d_work;
    // this frame's counter set; the other one is zeroed by this frame's clear for the next frame of the context
    p.counters = (uint32_t*)c.d_counters + (c.frames & 1u) * kCounterWords;
    // this frame's counters (queue fills, status bits), for whoever waits for the frame (check_frames, get_counters): stored by
    // k_resolve into the pinned ring.  The bounds-checking build, whose k_resolve may still set a status bit, copies them
    // behind the frame instead.
    uint32_t* const h_status_slot = c.h_status + (c.submitted % kStatusRing) * 16;
#if defined(TOPO_BOUNDS_CHECK) || defined(TOPO_RESOLVE_PROF)
    p.status_out = nullptr;
#else
    static const bool status_copy = getenv("TOPO_STATUS_BY_COPY") && atoi(getenv("TOPO_STATUS_BY_COPY")) != 0;
    p.status_out = status_copy ? nullptr : h_status_slot;
#endif
    uint32_t* const counters_next = (uint32_t*)c.d_counters + ((c.frames & 1u) ^ 1u) * kCounterWords;
    p.big = (BigItem*)c.d_big;
    p.rare = (RareItem*)c.d_rare;
    p.far = (FarItem*)c.d_far;
    p.work2 = (WorkItem*)c.d_work2;
    p.split_m = occlusion_split_m_;
    p.rare_cap = (uint32_t)rare_cap;
    p.work_cap = (uint32_t)work_cap;
    p.far_sub_cap = (uint32_t)far_sub_cap;
    p.near_cap = (uint32_t)near_cap;
    p.near_strip = near_strip;
    p.big_cap = (uint32_t)big_cap;
    p.n_views = n;
    p.n_tiles = n_tiles;
    p.W = (int32_t)w;
    p.H = (int32_t)h;
    p.tile_w = tile_w_;
    p.tile_h = tile_h_;
    p.bx_count = bxc;
    p.by_count = byc;
    p.tris_per_tile = n_tiles ? 2u * (tile_w_ - 1) * (tile_h_ - 1) : 8u;
    p.div_tris = fastdiv_make(p.tris_per_tile);
    p.div_hm1 = fastdiv_make(n_tiles ? tile_h_ - 1 : 2u);
    p.rblocks_x = (w + kResolveBlockW - 1) / kResolveBlockW;
    p.rblocks_view = p.rblocks_x * ((h + kResolveBlockH - 1) / kResolveBlockH);
    p.div_rblocks_x = fastdiv_make(p.rblocks_x > 1 ? p.rblocks_x : 2u);
    p.div_rblocks_view = fastdiv_make(p.rblocks_view > 1 ? p.rblocks_view : 2u);
    {   // the cleared render target texel: Color{0, 0.71, 0.885, 1} (terrain_renderer.rs:379-384) stored as Rgba8UnormSrgb
        float thresh[256];
        for (int i = 0; i < 256; ++i) thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]);
        p.linear_target = (format_ == TOPO_FORMAT_RGBA8_UNORM || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
        p.bgra = (format_ == TOPO_FORMAT_BGRA8_UNORM_SRGB || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
        p.sky_c8 = (p.linear_target ? to_unorm8(0.0f) | (to_unorm8(0.71f) << 8) | (to_unorm8(0.885f) << 16)
                                    : srgb_encode(thresh, 0.0f) | (srgb_encode(thresh, 0.71f) << 8) | (srgb_encode(thresh, 0.885f) << 16)) |
                   (to_unorm8(1.0f) << 24);
    }
    last_blocks_tested_ = (uint32_t)work_cap;
    // Is the far phase worth its four launches (each ~4 us of GPU and ~9 us of host time)?  k_cull makes an occlusion-test
    // candidate of a block whose nearest possible view depth, w(centre) - radius |w row|, exceeds the split;
    // w(centre) <= w(C) + R |w row| for the sphere (C, R) around the tile's block centres (Tile::centres).  The split is a
    // performance knob with a flat optimum (60..120 km at c4; results do not depend on it): when that bound stays below 4/3 of
    // it for every view and tile -- a lone tile around the viewpoint -- the frame's split is raised above the bound, no block
    // becomes a candidate and the far phase is not launched.
    bool far_phase = p.split_m > 0.0f;
    static const bool far_skip_off = getenv("TOPO_FAR_SKIP") && atoi(getenv("TOPO_FAR_SKIP")) == 0;
    if (far_phase && !far_skip_off) {
        double bound = 0.0;
        for (uint32_t i = 0; i < n && bound >= 0.0; ++i) {
            const float* m = views[i].camera_proj;
            const double wn = std::sqrt((double)m[3] * m[3] + (double)m[7] * m[7] + (double)m[11] * m[11]);
            for (const auto& kv : tiles_) {
                const double* s = kv.second.centres;
                const double w_far = (double)m[3] * s[0] + (double)m[7] * s[1] + (double)m[11] * s[2] + (double)m[15] + s[3] * wn;
                if (!(s[3] >= 0.0) || !(w_far < 1e30)) { bound = -1.0; break; }      // unknown sphere, NaN or huge: keep the far phase
                bound = std::max(bound, w_far);
            }
        }
        if (bound >= 0.0 && bound + 2.0 < (double)p.split_m * (4.0 / 3.0)) {
            p.split_m = std::max(p.split_m, (float)(bound + 2.0));      // (>= bound + 1 after the rounding to f32: no candidates)
            far_phase = false;
        }
    }
    last_far_phase_ = far_phase;

    // clear -> cull -> [near blocks: raster, rare, big] -> occlusion test of the far blocks -> [survivors: raster,
    // rare, big] -> resolve.  Event slots: 0 clear, 1 cull, 2 raster(near), 3 rare+big(near), 4 occlusion,
    // 5 raster(far), 6 rare+big(far), 7 resolve.
    // A timing event between two kernels costs ~6 us of idle GPU (the next kernel waits for the marker), so only the
    // events the selected timing slots need are recorded (topo_set_timing_slots); slot -> stages: 0:{0} 1:{1} 2:{2,5} 3:{4}
    // 4:{3,6} 5:{7}, stage i = ev[i]..ev[i+1]; the total (ev[0], ev[8]) is always kept.
    uint32_t ev_need = timing_total_ ? 0x101u : 0u;
    {
        static const uint32_t stages_of_slot[6] = {1u << 0, 1u << 1, (1u << 2) | (1u << 5), 1u << 4, (1u << 3) | (1u << 6), 1u << 7};
        for (int sl = 0; sl < 6; ++sl)
            if (timing_slots_ & (1u << sl))
                for (int st = 0; st < 8; ++st)
                    if (stages_of_slot[sl] & (1u << st)) ev_need |= (3u << st);
    }
    // With nothing but k_resolve's duration and / or the total selected (bench.py's timed region) the events are not markers between
    // the kernels but the kernels' own start and end times (hipExtLaunchKernel: ev[0] = start of the frame's first kernel, ev[7] /
    // ev[8] = start / end of k_resolve): a pair of markers costs a frame 8-10 us, these next to nothing.
    const bool pixelize = post_.pixelize_n < 99.99999f;
    static const bool markers_only = getenv("TOPO_EVENTS_BY_MARKER") && atoi(getenv("TOPO_EVENTS_BY_MARKER")) != 0;
    const bool own_times = !markers_only && !pixelize && (timing_slots_ & ~(1u << 5)) == 0;
    const int ring = (int)(c.frames % kEvRing);
    hipEvent_t* const ev = c.evr[ring];
    c.evr_recorded[ring] = ev_need;
    c.evr_slots[ring] = timing_slots_;
    c.evr_frame[ring] = ++frame_seq_;
    ++c.frames;
    if ((ev_need & (1u << 0)) && !own_times) TOPO_HIP_TRY(hipEventRecord(ev[0], stream));
    const hipEvent_t ev_first = (ev_need & (1u << 0)) && own_times ? ev[0] : nullptr;
    // clear and cull side by side in one launch (timing slot "clear" then holds both, "cull" nothing); TOPO_FUSE_CLEAR_CULL=0 or
    // an empty tile set: one after the other
    if (fuse) launch_clear_cull(p, counters_next, stream, ev_first, pack_in_cull ? &pack : nullptr, n);
    else launch_clear(p, counters_next, stream, ev_first);
    if (ev_need & (1u << 1)) TOPO_HIP_TRY(hipEventRecord(ev[1], stream));
    if (!fuse) launch_cull(p, stream);
    if (ev_need & (1u << 2)) TOPO_HIP_TRY(hipEventRecord(ev[2], stream));
    launch_raster(p, 0, stream);
    if (ev_need & (1u << 3)) TOPO_HIP_TRY(hipEventRecord(ev[3], stream));
    launch_raster_rare(p, stream);
    launch_raster_big(p, stream);
    if (ev_need & (1u << 4)) TOPO_HIP_TRY(hipEventRecord(ev[4], stream));
    if (far_phase) {
        launch_occlusion(p, stream);
    }
    if (ev_need & (1u << 5)) TOPO_HIP_TRY(hipEventRecord(ev[5], stream));
    if (far_phase) launch_raster(p, 1, stream);
    if (ev_need & (1u << 6)) TOPO_HIP_TRY(hipEventRecord(ev[6], stream));
    if (far_phase) {
        launch_raster_rare(p, stream);
        launch_raster_big(p, stream);
    }
    if ((ev_need & (1u << 7)) && !own_times) TOPO_HIP_TRY(hipEventRecord(ev[7], stream));
    const hipEvent_t ev_rstart = (ev_need & (1u << 7)) && own_times ? ev[7] : nullptr, ev_rstop = (ev_need & (1u << 8)) && own_times ? ev[8] : nullptr;
    // The pixelise branch of the post shader (pixelize_n < 99.99999; the reference never takes it) samples the render target
    // away from the pixel's own texel: k_resolve then stores the render-target texels into an image of the context's
    // (post_off) and k_post_pixelize makes the surface image from it and the depth image.
    OutputParams kout = out;
    if (pixelize) {
        if (n_slots) return fail(TOPO_ERR_UNSUPPORTED, "the pixelise branch is not available on the slot-by-slot (multi-GPU) path");
        const size_t img = (size_t)w * h * 4;
        if (int rc = ensure_on(stream, &d_pre_rgba_, &cap_pre_rgba_, img * n)) return rc;
        if (!out.depth)
            if (int rc = ensure_on(stream, &d_pre_depth_, &cap_pre_depth_, img * n)) return rc;
        kout.rgba = (uint8_t*)d_pre_rgba_;
        kout.rgba_view_stride = img;
        kout.rgba_pitch = (size_t)w * 4;
        if (!out.depth) { kout.depth = (float*)d_pre_depth_; kout.depth_view_stride = img; kout.depth_pitch = (size_t)w * 4; }
        p.post_off = 1;
    }
    if (n_slots == 0) {
        p.rblock_first = 0;
        p.rblock_count = p.rblocks_view * n;
        launch_resolve(p, kout, stream, ev_rstart, ev_rstop);
        if (pixelize)
            launch_post_pixelize(n, (int32_t)w, (int32_t)h, post_.viewport[0] >= 1.0f ? post_.viewport[0] : (float)w, post_.viewport[1] >= 1.0f ? post_.viewport[1] : (float)h,
                                 post_.pixelize_n, (const uint8_t*)d_pre_rgba_, out, kout.depth, kout.depth_view_stride, kout.depth_pitch, p.linear_target, p.bgra, stream);
    } else {
        for (uint32_t i = 0; i < n_slots; ++i) {
            if ((uint64_t)slots[i].block_first + slots[i].block_count > (uint64_t)p.rblocks_view * n) return fail(TOPO_ERR_INVALID, "resolve slot outside the frame");
            p.rblock_first = slots[i].block_first;
            p.rblock_count = slots[i].block_count;
            launch_resolve(p, out, stream, i == 0 ? ev_rstart : nullptr, i + 1 == n_slots ? ev_rstop : nullptr);
            if (after_slot)
                if (int rc = (*after_slot)(i, stream)) return rc;
        }
    }
    if ((ev_need & (1u << 8)) && !own_times) TOPO_HIP_TRY(hipEventRecord(ev[8], stream));
    if (!p.status_out) TOPO_HIP_TRY(hipMemcpyAsync(h_status_slot, p.counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ++c.submitted;
    c.timed = true;
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

int TerrainRenderer::render_device(uint8_t* rgba_dev, size_t rgba_pitch, float* depth_dev, size_t depth_pitch) {
    if (!rgba_dev) return fail(TOPO_ERR_INVALID, "rgba_dev is null");
    if (!have_uniforms_) return fail(TOPO_ERR_INVALID, "topo_update has not been called");
    if (rgba_pitch < (size_t)W_ * 4 || (depth_dev && depth_pitch < (size_t)W_ * 4)) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    OutputParams o{};
    o.rgba = rgba_dev;
    o.rgba_view_stride = rgba_pitch * H_;
    o.rgba_pitch = rgba_pitch;
    o.depth = depth_dev;
    o.depth_view_stride = depth_pitch * H_;
    o.depth_pitch = depth_pitch;
    return render_views_device(1, &uniforms_, W_, H_, o);
}

// render (terrain_renderer.rs:365-452) + depth copy (render_engine.rs:219-249), host outputs.
int TerrainRenderer::render(uint8_t* rgba, size_t rgba_pitch, float* depth, size_t depth_pitch) {
    if (!rgba) return fail(TOPO_ERR_INVALID, "rgba_out is null");
    if (!have_uniforms_) return fail(TOPO_ERR_INVALID, "topo_update has not been called");
    if (rgba_pitch < (size_t)W_ * 4 || (depth && depth_pitch < (size_t)W_ * 4)) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    if (int rc = bind_device()) return rc;
    const size_t row = (size_t)W_ * 4;
    if (int rc = ensure(&d_out_rgba_, &cap_out_rgba_, row * H_)) return rc;
    if (depth)
        if (int rc = ensure(&d_out_depth_, &cap_out_depth_, row * H_)) return rc;
    OutputParams o{};
    o.rgba = (uint8_t*)d_out_rgba_;
    o.rgba_view_stride = row * H_;
    o.rgba_pitch = row;
    o.depth = depth ? (float*)d_out_depth_ : nullptr;
    o.depth_view_stride = row * H_;
    o.depth_pitch = row;
    // A frame whose rare-triangle queue overflowed is incomplete.  The synchronous entry point does not hand such a frame
    // out: it grows the queue to what the frame asked for and renders it again (an explicit topo_debug_set_queue_caps
    // setting is a test hook and is left alone: then the call fails with TOPO_ERR_CAPACITY).
    // Frames queued earlier through the asynchronous entry points are waited for, but an overflow of one of THEM is not this
    // call's error (the caller could not tell which frame failed, and this frame would go unrendered): it stays pending and
    // is reported, once, by the next topo_join / topo_synchronize -- the calls that wait for those frames.
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    for (auto& fc : ctx_)
        if (!fc.pending && fc.h_status) overflow_pending_ |= fold_frames(fc);
    for (int attempt = 0;; ++attempt) {
        if (int rc = render_views_device(1, &uniforms_, W_, H_, o)) return rc;
        if (int rc = join()) return rc;       // (pipelined contexts run on their own streams)
        TOPO_HIP_TRY(hipStreamSynchronize(stream_));
        FrameCtx& fc = ctx_[last_ctx_];
        const uint32_t* words = fc.h_status + ((fc.submitted - 1) % kStatusRing) * 16;
        const uint32_t status = words[2], wanted = words[3];
        if (!(status & kStatusRareOverflow) || rare_cap_cfg_ != 0 || attempt >= 3) {
            (void)check_frames();          // (folds this frame's words; its status is answered for right here)
            if (status & kStatusRareOverflow) return fail(TOPO_ERR_CAPACITY, "rare-triangle queue overflowed: frame incomplete");
            break;
        }
        (void)check_frames();
        rare_cap_auto_ = (uint64_t)wanted + wanted / 4u + 1024u;      // the overflowed frame counted what it needs
        if (rare_cap_auto_ > (1ull << 28)) return fail(TOPO_ERR_CAPACITY, "rare-triangle queue would exceed 2^28 entries");
    }
    if (int rc = download(rgba, rgba_pitch, (const uint8_t*)d_out_rgba_, row)) return rc;
    if (depth)
        if (int rc = download((uint8_t*)depth, depth_pitch, (const uint8_t*)d_out_depth_, row)) return rc;
    have_depth_ = depth != nullptr;
    depth_w_ = W_;
    depth_h_ = H_;
    return TOPO_OK;
}

// One H_-row image from device memory into the caller's HOST buffer.  A copy into pageable memory makes the runtime stage it
// through its own small pinned buffers, synchronously: ~10 GB/s, 6.8 ms for the RGBA + depth of a 2048 x 4096 frame, fifty
// times the frame's render time.  So: buffers the caller has pinned (topo_pin_host_buffer) are written directly, at the
// link's rate; any other buffer is filled through a pinned staging buffer of the context's own, in slices -- the device
// copies slice k + 1 while a few host threads move slice k on to the caller's rows.
int TerrainRenderer::download(uint8_t* dst, size_t dst_pitch, const uint8_t* src_dev, size_t row) {
    const size_t span = dst_pitch * (H_ - 1) + row;
    for (const auto& pin : pinned_)
        if (dst >= pin.first && dst + span <= pin.first + pin.second) {
            TOPO_HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src_dev, row, row, H_, hipMemcpyDeviceToHost, stream_));
            TOPO_HIP_TRY(hipStreamSynchronize(stream_));
            return TOPO_OK;
        }
    const size_t total = row * H_;
    if (total > cap_stage_) {
        if (h_stage_) (void)hipHostFree(h_stage_);
        h_stage_ = nullptr;
        cap_stage_ = 0;
        TOPO_HIP_TRY(hipHostMalloc((void**)&h_stage_, total));
        cap_stage_ = total;
    }
    constexpr int kSlices = 8;
    if (!stage_ev_[0])
        for (auto& e : stage_ev_) TOPO_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint32_t rows_per = (H_ + kSlices - 1) / kSlices;
    int n_slices = 0;
    for (uint32_t r0 = 0; r0 < H_; r0 += rows_per, ++n_slices) {
        const uint32_t rows = std::min(rows_per, H_ - r0);
        TOPO_HIP_TRY(hipMemcpyAsync(h_stage_ + (size_t)r0 * row, src_dev + (size_t)r0 * row, (size_t)rows * row, hipMemcpyDeviceToHost, stream_));
        TOPO_HIP_TRY(hipEventRecord(stage_ev_[n_slices], stream_));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = (int)std::min<size_t>(std::max(1u, std::min(hw ? hw : 4u, 8u)), std::max<size_t>(1, total >> 20));      // one per MiB, at most 8
    std::atomic<int> failed{0};
    auto worker = [&](int t) {
        (void)hipSetDevice(device_);
        for (int k = 0; k < n_slices; ++k) {
            if (hipEventSynchronize(stage_ev_[k]) != hipSuccess) { failed = 1; return; }
            const uint32_t r0 = (uint32_t)k * rows_per, rows = std::min(rows_per, H_ - r0);
            const uint32_t a = r0 + (uint32_t)((uint64_t)rows * t / n_threads), b = r0 + (uint32_t)((uint64_t)rows * (t + 1) / n_threads);
            if (dst_pitch == row) memcpy(dst + (size_t)a * row, h_stage_ + (size_t)a * row, (size_t)(b - a) * row);
            else
                for (uint32_t r = a; r < b; ++r) memcpy(dst + (size_t)r * dst_pitch, h_stage_ + (size_t)r * row, row);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    if (failed) return fail(TOPO_ERR_HIP, "device-to-host copy failed");
    return TOPO_OK;
}

// The caller's own output buffers, pinned once (hipHostRegister): topo_render then copies into them directly.
int TerrainRenderer::pin_host_buffer(void* p, size_t bytes) {
    if (!p || bytes == 0) return fail(TOPO_ERR_INVALID, "null/empty buffer");
    if (int rc = bind_device()) return rc;
    for (const auto& pin : pinned_)
        if (pin.first == (uint8_t*)p) return pin.second == bytes ? TOPO_OK : fail(TOPO_ERR_INVALID, "buffer already pinned with another size");
    TOPO_HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    pinned_.emplace_back((uint8_t*)p, bytes);
    return TOPO_OK;
}

int TerrainRenderer::unpin_host_buffer(void* p) {
    for (size_t i = 0; i < pinned_.size(); ++i)
        if (pinned_[i].first == (uint8_t*)p) {
            if (int rc = bind_device()) return rc;
            TOPO_HIP_TRY(hipStreamSynchronize(stream_));
            TOPO_HIP_TRY(hipHostUnregister(p));
            pinned_.erase(pinned_.begin() + (long)i);
            return TOPO_OK;
        }
    return fail(TOPO_ERR_NOT_FOUND, "buffer was not pinned by topo_pin_host_buffer");
}

// LineRenderer::render (line_renderer.rs:200-212) over an image this context produced: the overlay triangles are drawn
// on top of the post pass's output with the reference's layering (depth Greater against the post quad's 1/4096).
int TerrainRenderer::overlay_lines_device(const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, float line_width,
                                          uint8_t* rgba_dev, size_t rgba_pitch) {
    if ((n_vertices && !vertices) || (n_indices && !indices) || !rgba_dev) return fail(TOPO_ERR_INVALID, "null argument");
    if (n_indices % 3 != 0) return fail(TOPO_ERR_INVALID, "the overlay is a triangle list: index count must be a multiple of 3");
    if (rgba_pitch < (size_t)W_ * 4) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    if (int rc = bind_device()) return rc;
    const size_t vb = (size_t)n_vertices * sizeof(OverlayVertex), ib = (size_t)n_indices * 4, keys_b = (size_t)W_ * H_ * 8;
    if (int rc = ensure(&d_overlay_geo_, &cap_overlay_geo_, ((vb + 15) & ~(size_t)15) + ib + 16)) return rc;
    const bool fresh = keys_b > cap_overlay_keys_ || overlay_w_ != W_ || overlay_h_ != H_;
    if (int rc = ensure(&d_overlay_keys_, &cap_overlay_keys_, keys_b)) return rc;
    overlay_w_ = W_; overlay_h_ = H_;
    uint8_t* geo = (uint8_t*)d_overlay_geo_;
    uint32_t* d_idx = (uint32_t*)(geo + ((vb + 15) & ~(size_t)15));
    if (vb) TOPO_HIP_TRY(hipMemcpyAsync(geo, vertices, vb, hipMemcpyHostToDevice, stream_));
    if (ib) TOPO_HIP_TRY(hipMemcpyAsync(d_idx, indices, ib, hipMemcpyHostToDevice, stream_));
    const uint32_t linear = (format_ == TOPO_FORMAT_RGBA8_UNORM || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
    const uint32_t bgra = (format_ == TOPO_FORMAT_BGRA8_UNORM_SRGB || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
    launch_overlay((const OverlayVertex*)geo, d_idx, n_indices / 3, n_vertices, line_width, (int32_t)W_, (int32_t)H_, (uint64_t*)d_overlay_keys_, fresh,
                   rgba_dev, rgba_pitch, linear, bgra, stream_);
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));      // the geometry is only borrowed for the call
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

// Host image in, host image out (the frame topo_render returned, or any W x H image in the context's format).
int TerrainRenderer::overlay_lines(const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, float line_width,
                                   uint8_t* rgba, size_t rgba_pitch) {
    if (!rgba) return fail(TOPO_ERR_INVALID, "null argument");
    if (rgba_pitch < (size_t)W_ * 4) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    if (int rc = bind_device()) return rc;
    const size_t row = (size_t)W_ * 4;
    if (int rc = ensure(&d_out_rgba_, &cap_out_rgba_, row * H_)) return rc;
    TOPO_HIP_TRY(hipMemcpy2DAsync(d_out_rgba_, row, rgba, rgba_pitch, row, H_, hipMemcpyHostToDevice, stream_));
    if (int rc = overlay_lines_device(vertices, n_vertices, indices, n_indices, line_width, (uint8_t*)d_out_rgba_, row)) return rc;
    TOPO_HIP_TRY(hipMemcpy2DAsync(rgba, rgba_pitch, d_out_rgba_, row, row, H_, hipMemcpyDeviceToHost, stream_));
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    return TOPO_OK;
}

// TextRenderer::render (text_renderer.rs:198-204) over an image this context produced: glyphon's glyph quads, alpha-blended,
// the first quad over a pixel keeping it (depth Greater with write at one depth).
int TerrainRenderer::overlay_glyphs_device(const void* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t atlas_w, uint32_t atlas_h,
                                           uint8_t* rgba_dev, size_t rgba_pitch) {
    if ((n_glyphs && (!glyphs || !atlas)) || !rgba_dev) return fail(TOPO_ERR_INVALID, "null argument");
    if (rgba_pitch < (size_t)W_ * 4) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    if (!(depth > 1.0f / 4096.0f) || !(depth <= 1.0f)) return fail(TOPO_ERR_INVALID, "glyph depth must lie above the post quad's 1/4096 (the reference: 100/4096)");
    const GlyphInstance* gs = (const GlyphInstance*)glyphs;
    for (uint32_t i = 0; i < n_glyphs; ++i)
        if (gs[i].content_type_with_srgb[0] != 1) return fail(TOPO_ERR_UNSUPPORTED, "only mask glyphs (glyphon content type 1) are drawn; colour glyphs are not");
    if (int rc = bind_device()) return rc;
    const size_t gb = (size_t)n_glyphs * sizeof(GlyphInstance), ab = (size_t)atlas_w * atlas_h, keys_b = (size_t)W_ * H_ * 8;
    if (int rc = ensure(&d_overlay_geo_, &cap_overlay_geo_, ((gb + 15) & ~(size_t)15) + ab + 16)) return rc;
    const bool fresh = keys_b > cap_overlay_keys_ || overlay_w_ != W_ || overlay_h_ != H_;
    if (int rc = ensure(&d_overlay_keys_, &cap_overlay_keys_, keys_b)) return rc;
    overlay_w_ = W_; overlay_h_ = H_;
    uint8_t* geo = (uint8_t*)d_overlay_geo_;
    uint8_t* d_atlas = geo + ((gb + 15) & ~(size_t)15);
    if (gb) TOPO_HIP_TRY(hipMemcpyAsync(geo, glyphs, gb, hipMemcpyHostToDevice, stream_));
    if (ab) TOPO_HIP_TRY(hipMemcpyAsync(d_atlas, atlas, ab, hipMemcpyHostToDevice, stream_));
    const uint32_t linear = (format_ == TOPO_FORMAT_RGBA8_UNORM || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
    const uint32_t bgra = (format_ == TOPO_FORMAT_BGRA8_UNORM_SRGB || format_ == TOPO_FORMAT_BGRA8_UNORM) ? 1u : 0u;
    launch_overlay_glyphs((const GlyphInstance*)geo, n_glyphs, depth, d_atlas, atlas_w, atlas_h, (int32_t)W_, (int32_t)H_, (uint64_t*)d_overlay_keys_, fresh, rgba_dev,
                          rgba_pitch, linear, bgra, stream_);
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));      // glyphs and atlas are only borrowed for the call
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

int TerrainRenderer::overlay_glyphs(const void* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t atlas_w, uint32_t atlas_h, uint8_t* rgba,
                                    size_t rgba_pitch) {
    if (!rgba) return fail(TOPO_ERR_INVALID, "null argument");
    if (rgba_pitch < (size_t)W_ * 4) return fail(TOPO_ERR_INVALID, "pitch smaller than a row");
    if (int rc = bind_device()) return rc;
    const size_t row = (size_t)W_ * 4;
    if (int rc = ensure(&d_out_rgba_, &cap_out_rgba_, row * H_)) return rc;
    TOPO_HIP_TRY(hipMemcpy2DAsync(d_out_rgba_, row, rgba, rgba_pitch, row, H_, hipMemcpyHostToDevice, stream_));
    if (int rc = overlay_glyphs_device(glyphs, n_glyphs, depth, atlas, atlas_w, atlas_h, (uint8_t*)d_out_rgba_, row)) return rc;
    TOPO_HIP_TRY(hipMemcpy2DAsync(rgba, rgba_pitch, d_out_rgba_, row, row, H_, hipMemcpyDeviceToHost, stream_));
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    return TOPO_OK;
}

// RenderEngine::get_visible_labels over the depth the context already holds on the device.
int TerrainRenderer::visible_peaks_device(const topo_uniforms* view, uint32_t w, uint32_t h, const float* depth_dev, size_t depth_pitch,
                                          uint32_t n, const float* peaks_dev, uint8_t* visible_dev, uint32_t* xy_dev) {
    if (!view || !depth_dev || (n && (!peaks_dev || !visible_dev || !xy_dev))) return fail(TOPO_ERR_INVALID, "null argument");
    if (int rc = bind_device()) return rc;
    if (int rc = ensure(&d_proj_, &cap_proj_, 16 * sizeof(float))) return rc;
    TOPO_HIP_TRY(hipMemcpyAsync(d_proj_, view->camera_proj, 16 * sizeof(float), hipMemcpyHostToDevice, stream_));
    launch_visible_peaks((const float*)d_proj_, w, h, depth_dev, depth_pitch, n, peaks_dev, visible_dev, xy_dev, stream_);
    TOPO_HIP_TRY(hipGetLastError());
    return TOPO_OK;
}

int TerrainRenderer::visible_peaks(uint32_t n, const float* peaks, uint8_t* visible, uint32_t* xy) {
    if (n && (!peaks || !visible || !xy)) return fail(TOPO_ERR_INVALID, "null argument");
    if (!have_depth_ || depth_w_ != W_ || depth_h_ != H_)
        return fail(TOPO_ERR_INVALID, "topo_visible_peaks needs the depth of a preceding topo_render(.., depth_out, ..) at the current size");
    if (n == 0) return TOPO_OK;
    if (int rc = bind_device()) return rc;
    const size_t in_b = (size_t)n * 12, xy_b = (size_t)n * 8, vis_b = ((size_t)n + 15) & ~(size_t)15;
    if (int rc = ensure(&d_peaks_, &cap_peaks_, in_b + xy_b + vis_b)) return rc;
    uint8_t* base = (uint8_t*)d_peaks_;
    TOPO_HIP_TRY(hipMemcpyAsync(base, peaks, in_b, hipMemcpyHostToDevice, stream_));
    if (int rc = visible_peaks_device(&uniforms_, W_, H_, (const float*)d_out_depth_, (size_t)W_ * 4, n, (const float*)base,
                                      base + in_b + xy_b, (uint32_t*)(base + in_b)))
        return rc;
    TOPO_HIP_TRY(hipMemcpyAsync(xy, base + in_b, xy_b, hipMemcpyDeviceToHost, stream_));
    TOPO_HIP_TRY(hipMemcpyAsync(visible, base + in_b + xy_b, n, hipMemcpyDeviceToHost, stream_));
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    return TOPO_OK;
}

int TerrainRenderer::set_stream(hipStream_t s) {
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    stream_ = s ? s : own_stream_;
    return TOPO_OK;
}

int TerrainRenderer::synchronize() {
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    return check_frames();
}

int TerrainRenderer::join_frames() {
    if (int rc = join()) return rc;
    if (pipeline_depth_ == 1) TOPO_HIP_TRY(hipStreamSynchronize(stream_));      // one frame in flight: it runs on stream_
    return check_frames();
}

int TerrainRenderer::frame_status(uint32_t out[4]) {
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    // (folds the finished frames into last_status_; an overflow is reported through out[0] here, and stays pending as the error
    // of the next call that waits for frames)
    for (auto& fc : ctx_)
        if (!fc.pending && fc.h_status) overflow_pending_ |= fold_frames(fc);
    for (int i = 0; i < 4; ++i) out[i] = last_status_[i];
    last_status_[0] = last_status_[1] = last_status_[2] = last_status_[3] = 0;
    return TOPO_OK;
}

int TerrainRenderer::set_normals_lds_rows(int rows) {
    if (rows != 0 && rows != 4 && rows != 8 && rows != 16 && rows != 32 && rows != 64) return fail(TOPO_ERR_INVALID, "lds rows must be 4, 8, 16, 32 or 64 (0: the form without an LDS tile)");
    lds_rows_ = rows;
    return TOPO_OK;
}

int TerrainRenderer::set_timing_slots(uint32_t mask) {
    timing_slots_ = mask & 0x3Fu;
    timing_total_ = !(mask & TOPO_TIMING_NO_TOTAL);
    return TOPO_OK;
}

int TerrainRenderer::set_occlusion_split(float metres) {
    if (!(metres >= 0.0f)) return fail(TOPO_ERR_INVALID, "split must be >= 0");
    occlusion_split_m_ = metres;
    return TOPO_OK;
}

int TerrainRenderer::set_queue_caps(uint32_t big_cap, uint32_t rare_cap) {
    big_cap_cfg_ = big_cap;
    // bit 31 of rare_cap: "start at this capacity but grow on demand" (what the default does from 4 Mi entries)
    rare_cap_cfg_ = (rare_cap & 0x80000000u) ? 0u : rare_cap;
    rare_cap_auto_ = (rare_cap & 0x80000000u) ? (rare_cap & 0x7FFFFFFFu) : 0u;
    return join();
}

// Durations of the frame whose events are set `ring` of context c (out[0..6]; the frame must have completed).
int TerrainRenderer::frame_durations(FrameCtx& c, int ring, float out[7]) {
    hipEvent_t* ev = c.evr[ring];
    float d[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = 0.0f;
        if ((c.evr_recorded[ring] >> i & 3u) == 3u) TOPO_HIP_TRY(hipEventElapsedTime(&d[i], ev[i], ev[i + 1]));
    }
    out[0] = d[0];                // clear
    out[1] = d[1];                // cull
    out[2] = d[2] + d[5];         // raster: near blocks + far survivors
    out[3] = d[4];                // occlusion test
    out[4] = d[3] + d[6];         // rare + big (both phases)
    out[5] = d[7];                // resolve
    for (int sl = 0; sl < 6; ++sl)
        if (!(c.evr_slots[ring] & (1u << sl))) out[sl] = 0.0f;      // (a neighbour's events may have bracketed it by chance)
    out[6] = 0.0f;
    if ((c.evr_recorded[ring] & 0x101u) == 0x101u) TOPO_HIP_TRY(hipEventElapsedTime(&out[6], ev[0], ev[8]));
    return TOPO_OK;
}

int TerrainRenderer::get_timings(float out[TOPO_TIMING_SLOTS]) {
    for (int i = 0; i < TOPO_TIMING_SLOTS; ++i) out[i] = 0.0f;
    if (int rc = bind_device()) return rc;
    // depth 1: the last frame.  Pipelined: the OLDEST frame in flight (the context the next submission will reuse), so
    // that reading timings every frame does not wait for the frame just submitted
    FrameCtx& c = ctx_[pipeline_depth_ > 1 ? next_ctx_ : last_ctx_];
    if (c.timed && c.frames) {
        const int ring = (int)((c.frames - 1) % kEvRing);
        if (c.evr_recorded[ring] & 0x100u) TOPO_HIP_TRY(hipEventSynchronize(c.evr[ring][8]));
        else TOPO_HIP_TRY(hipStreamSynchronize(pipeline_depth_ > 1 ? c.stream : stream_));      // (TOPO_TIMING_NO_TOTAL: no event behind the frame)
        if (int rc = frame_durations(c, ring, out)) return rc;
    }
    if (load_timed_) {
        TOPO_HIP_TRY(hipEventSynchronize(load_ev_[1]));
        TOPO_HIP_TRY(hipEventElapsedTime(&out[7], load_ev_[0], load_ev_[1]));      // the whole load phase
        TOPO_HIP_TRY(hipEventElapsedTime(&out[8], load_ev_[0], load_ev_[2]));      // its tables part
    }
    return TOPO_OK;
}

// The last n_frames frames (at most kEvRing per context), oldest first, 7 durations each (slots [0]..[6] of topo_get_timings).
// Waits for the frames in flight: meant to be called after a timed region, not inside it.
int TerrainRenderer::get_timing_history(uint32_t n_frames, float* out_ms, uint32_t* n_out) {
    *n_out = 0;
    if (!out_ms && n_frames) return fail(TOPO_ERR_INVALID, "null argument");
    if (int rc = join()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    struct Ref { uint64_t frame; int ctx, ring; };
    std::vector<Ref> refs;
    for (int ci = 0; ci < kMaxPipeline; ++ci) {
        FrameCtx& c = ctx_[ci];
        const uint64_t have = c.frames < (uint64_t)kEvRing ? c.frames : (uint64_t)kEvRing;
        for (uint64_t k = 0; k < have; ++k) {
            const int ring = (int)((c.frames - 1 - k) % kEvRing);
            refs.push_back(Ref{c.evr_frame[ring], ci, ring});
        }
    }
    std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.frame < b.frame; });
    const size_t n = std::min<size_t>(n_frames, refs.size());
    for (size_t i = 0; i < n; ++i) {
        const Ref& r = refs[refs.size() - n + i];
        if (int rc = frame_durations(ctx_[r.ctx], r.ring, out_ms + 7 * i)) return rc;
    }
    *n_out = (uint32_t)n;
    return TOPO_OK;
}

int TerrainRenderer::get_counters(uint32_t out[6]) {
    for (int i = 0; i < 6; ++i) out[i] = 0;
    FrameCtx& fc = ctx_[last_ctx_];
    if (!fc.h_status) return TOPO_OK;
    if (int rc = bind_device()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(pipeline_depth_ > 1 ? fc.stream : stream_));
    uint32_t c[16] = {};
    if (fc.submitted) memcpy(c, fc.h_status + ((fc.submitted - 1) % kStatusRing) * 16, sizeof c);
    if (getenv("TOPO_DEBUG_COUNTERS")) {   // raw queue counters, for kernel experiments
        fprintf(stderr, "[topo] counters:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", c[i]);
        fprintf(stderr, "\n");
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3]; out[4] = c[4]; out[5] = c[5];
    return TOPO_OK;
}

int TerrainRenderer::read_normals(int32_t lat, int32_t lon, uint8_t* out) {
    Tile* t = find(lat, lon);
    if (!t) return fail(TOPO_ERR_NOT_FOUND, "no such tile");
    if (int rc = bind_device()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    TOPO_HIP_TRY(hipMemcpy(out, t->d_normals, (size_t)tile_w_ * tile_h_ * 4, hipMemcpyDeviceToHost));
    return TOPO_OK;
}

int TerrainRenderer::read_tile_tables(int32_t lat, int32_t lon, float* minmax_out, float* trig_out, double* bounds_out, uint32_t* n_blocks_out) {
    Tile* t = find(lat, lon);
    if (!t) return fail(TOPO_ERR_NOT_FOUND, "no such tile");
    if (int rc = bind_device()) return rc;
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    const uint32_t bxc = (tile_w_ - 1 + kBCX - 1) / kBCX, byc = (tile_h_ - 1 + kBCY - 1) / kBCY;
    const size_t nb = (size_t)bxc * byc;
    if (n_blocks_out) *n_blocks_out = (uint32_t)nb;
    if (minmax_out) TOPO_HIP_TRY(hipMemcpy(minmax_out, t->dev.block_minmax, nb * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (trig_out) TOPO_HIP_TRY(hipMemcpy(trig_out, t->dev.trig_lon, 2 * ((size_t)tile_w_ + tile_h_) * sizeof(float), hipMemcpyDeviceToHost));
    if (bounds_out) TOPO_HIP_TRY(hipMemcpy(bounds_out, t->dev.block_bounds, nb * 17 * sizeof(double), hipMemcpyDeviceToHost));
    return TOPO_OK;
}

// ---- GeoTIFF (fetch_terrain's decode step, background_runner.rs:113-136) ---------------------------------------
int geotiff_transform(const TiffInfo& ti, float rp[2], float mp[2], float ps[2]) {
    // CoordinateTransform::from_geo_tag_data (coordinate_transform.rs:23-57)
    if (ti.has_model_transformation) return TOPO_ERR_UNSUPPORTED;
    if (!ti.has_pixel_scale || !ti.has_tie_points) return TOPO_ERR_UNSUPPORTED;
    if (ti.pixel_scale.size() != 3 || ti.tie_points.size() != 6) return TOPO_ERR_INVALID;
    rp[0] = (float)ti.tie_points[0]; rp[1] = (float)ti.tie_points[1];
    mp[0] = (float)ti.tie_points[3]; mp[1] = (float)ti.tie_points[4];
    ps[0] = (float)ti.pixel_scale[0]; ps[1] = (float)ti.pixel_scale[1];
    return TOPO_OK;
}

// Decodes the first image of the file into a fresh device raster (caller frees *d_heights with hipFree).
int TerrainRenderer::geotiff_to_device(const uint8_t* bytes, size_t n, float** d_heights, uint32_t* w, uint32_t* h, float rp[2],
                                       float mp[2], float ps[2]) {
    *d_heights = nullptr;
    TiffInfo ti;
    std::string e;
    if (int rc = tiff_parse(bytes, n, ti, e)) return fail(rc, "GeoTIFF: " + e);
    if (int rc = geotiff_transform(ti, rp, mp, ps))
        return fail(rc, rc == TOPO_ERR_UNSUPPORTED ? "GeoTIFF: IncorrectGeoTags (ModelPixelScale + ModelTiepoint without ModelTransformation required)"
                                                    : "GeoTIFF: IncorrectGeoTagData (ModelPixelScale needs 3 and ModelTiepoint 6 values)");
    if (int rc = bind_device()) return rc;
    // host: the byte streams of all strips/tiles, decompressed back to back (4-byte aligned: sizes are multiples of 4)
    std::vector<TiffSegDev> segs;
    std::vector<uint32_t> row_seg;
    size_t total = 0;
    for (const TiffSegment& s : ti.segments) {
        TiffSegDev d{};
        d.byte_off = total; d.row0 = (uint32_t)row_seg.size();
        d.x0 = s.x0; d.y0 = s.y0; d.w = s.w; d.h = s.h;
        for (uint32_t r = 0; r < s.h; ++r) row_seg.push_back((uint32_t)segs.size());
        segs.push_back(d);
        total += (size_t)s.w * s.h * 4;
    }
    std::vector<uint8_t> staged(total);
    {   // strips/tiles are independent byte streams: inflate them on a few host threads
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t n_thr = std::min<size_t>(segs.size(), std::min<unsigned>(hw ? hw : 1, 8));
        std::vector<int> rcs(n_thr, TOPO_OK);
        std::vector<std::string> errs(n_thr);
        std::atomic<size_t> next{0};
        auto work = [&](size_t tid) {
            for (size_t k = next++; k < segs.size(); k = next++)
                if (int rc = tiff_segment_bytes(bytes, n, ti, ti.segments[k], staged.data() + segs[k].byte_off, errs[tid])) { rcs[tid] = rc; return; }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
        for (size_t t = 0; t < n_thr; ++t)
            if (rcs[t]) return fail(rcs[t], "GeoTIFF: " + errs[t]);
    }
    uint8_t* d_bytes = nullptr;
    TiffSegDev* d_segs = nullptr;
    uint32_t* d_rows = nullptr;
    float* d_out = nullptr;
    hipError_t he = hipMalloc((void**)&d_bytes, total ? total : 4);
    if (he == hipSuccess) he = hipMalloc((void**)&d_segs, segs.size() * sizeof(TiffSegDev));
    if (he == hipSuccess) he = hipMalloc((void**)&d_rows, row_seg.size() * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMalloc((void**)&d_out, (size_t)ti.width * ti.height * sizeof(float));
    if (he == hipSuccess) he = hipMemcpyAsync(d_bytes, staged.data(), total, hipMemcpyHostToDevice, stream_);
    if (he == hipSuccess) he = hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(TiffSegDev), hipMemcpyHostToDevice, stream_);
    if (he == hipSuccess) he = hipMemcpyAsync(d_rows, row_seg.data(), row_seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream_);
    if (he == hipSuccess) {
        launch_tiff_rows(d_bytes, d_segs, d_rows, (uint32_t)row_seg.size(), d_out, ti.width, ti.height, ti.predictor, ti.big_endian, stream_);
        he = hipStreamSynchronize(stream_);          // the staging vectors are only borrowed for the call
    }
    (void)hipFree(d_bytes); (void)hipFree(d_segs); (void)hipFree(d_rows);
    if (he != hipSuccess) {
        (void)hipFree(d_out);
        return hip_fail(he, "GeoTIFF decode");
    }
    *d_heights = d_out;
    *w = ti.width;
    *h = ti.height;
    return TOPO_OK;
}

int TerrainRenderer::geotiff_decode(const uint8_t* bytes, size_t n, float* heights_out, size_t capacity) {
    if (!bytes || !heights_out) return fail(TOPO_ERR_INVALID, "null argument");
    float* d = nullptr;
    uint32_t w = 0, h = 0;
    float rp[2], mp[2], ps[2];
    if (int rc = geotiff_to_device(bytes, n, &d, &w, &h, rp, mp, ps)) return rc;
    int rc = TOPO_OK;
    if ((size_t)w * h > capacity) rc = fail(TOPO_ERR_CAPACITY, "heights_out is smaller than the image");
    else if (hipMemcpy(heights_out, d, (size_t)w * h * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(TOPO_ERR_HIP, "copy of the decoded raster failed");
    (void)hipFree(d);
    return rc;
}

int TerrainRenderer::add_terrain_geotiff(int32_t lat, int32_t lon, const uint8_t* bytes, size_t n) {
    if (!bytes) return fail(TOPO_ERR_INVALID, "null argument");
    float* d = nullptr;
    uint32_t w = 0, h = 0;
    float rp[2], mp[2], ps[2];
    if (int rc = geotiff_to_device(bytes, n, &d, &w, &h, rp, mp, ps)) return rc;
    const int rc = add_terrain(lat, lon, d, true, w, h, rp, mp, ps);      // copies device-to-device
    (void)hipFree(d);
    return rc;
}

int TerrainRenderer::probe_sincos(const float* x, float* s, float* c, size_t n) {
    if (int rc = bind_device()) return rc;
    float *dx = nullptr, *ds = nullptr, *dc = nullptr;
    TOPO_HIP_TRY(hipMalloc((void**)&dx, n * 4));
    TOPO_HIP_TRY(hipMalloc((void**)&ds, n * 4));
    TOPO_HIP_TRY(hipMalloc((void**)&dc, n * 4));
    TOPO_HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    launch_probe_sincos(dx, ds, dc, n, stream_);
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    TOPO_HIP_TRY(hipMemcpy(s, ds, n * 4, hipMemcpyDeviceToHost));
    TOPO_HIP_TRY(hipMemcpy(c, dc, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dc);
    return TOPO_OK;
}

int TerrainRenderer::probe_div(int32_t kind, const float* x, const float* y, float* out, size_t n) {
    if (kind < 0 || kind > 8 || !x || !y || !out) return fail(TOPO_ERR_INVALID, "probe_div: bad argument");
    if (int rc = bind_device()) return rc;
    float *dx = nullptr, *dy = nullptr, *dq = nullptr;
    TOPO_HIP_TRY(hipMalloc((void**)&dx, n * 4));
    TOPO_HIP_TRY(hipMalloc((void**)&dy, n * 4));
    TOPO_HIP_TRY(hipMalloc((void**)&dq, n * 4));
    TOPO_HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    TOPO_HIP_TRY(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
    launch_probe_div(kind, dx, dy, dq, n, stream_);
    TOPO_HIP_TRY(hipStreamSynchronize(stream_));
    TOPO_HIP_TRY(hipMemcpy(out, dq, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dq);
    return TOPO_OK;
}

// =========================================================================================================
// Host-side CPU math of the reference (glam 0.31.0, Cargo.lock:1272-1273), restated in f32.
// =========================================================================================================
namespace {

inline float rs_to_radians(float d) { return d * 0.017453292519943295f; }   // f32::to_radians

struct V3 { float x, y, z; };
inline float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 vcross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 vnormalize(V3 v) {   // Vec3::normalize: self * length_recip()
    const float r = 1.0f / sqrtf(vdot(v, v));
    return {v.x * r, v.y * r, v.z * r};
}

// Mat3::from_euler(EulerRot::XYZEx, 0, b, c) = Rz(c) * Ry(b); with a = 0 every entry is one product.
void euler_xyz_ex_a0(float b, float c, float m[9] /*column-major*/) {
    const float si = sinf(0.0f), ci = cosf(0.0f);
    const float sj = sinf(b), cj = cosf(b), sh = sinf(c), ch = cosf(c);
    const float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
    m[0] = cj * ch;       m[1] = cj * sh;       m[2] = -sj;
    m[3] = sj * sc - cs;  m[4] = sj * ss + cc;  m[5] = cj * si;
    m[6] = sj * cc + ss;  m[7] = sj * cs - sc;  m[8] = cj * ci;
}

}  // namespace

// TerrainUniforms::new's normal_to_world_rot (render/data.rs:125-133)
void terrain_rotation(float model_lon_deg, float model_lat_deg, float rot[9]) {
    euler_xyz_ex_a0(rs_to_radians(90.0f - model_lat_deg), rs_to_radians(model_lon_deg), rot);
}

// geometry::transform (render/geometry.rs:12-20)
void geometry_transform(float h, float lon_deg, float lat_deg, float out[3]) {
    const float r = kR0 + h;
    const float lon = rs_to_radians(lon_deg), lat = rs_to_radians(lat_deg);
    out[0] = r * cosf(lat) * cosf(lon);
    out[1] = r * cosf(lat) * sinf(lon);
    out[2] = r * sinf(lat);
}

// Uniforms::new (render/data.rs:44-58) over Camera::{up,direction,get_view,build_view_proj_matrix}
// (data/camera.rs:97-128) and LightAngle::to_vec3 (:44-53).
void camera_uniforms(const float eye_in[3], float yaw, float pitch, float fov_y, float width, float height,
                     float sun_theta_deg, float sun_phi_deg, int32_t view_mode, topo_uniforms* out) {
    memset(out, 0, sizeof *out);
    const V3 eye = {eye_in[0], eye_in[1], eye_in[2]};
    const V3 up = vnormalize(eye);
    // Quat::from_rotation_arc(-Y, up)
    const V3 from = {0.0f, -1.0f, 0.0f};
    float qx, qy, qz, qw;
    const float d = vdot(from, up);
    const float one_minus_eps = 1.0f - 2.0f * 1.1920929e-7f;
    if (d > one_minus_eps) {
        qx = qy = qz = 0.0f; qw = 1.0f;
    } else if (d < -one_minus_eps) {   // any_orthonormal_vector(from), half-turn
        const float sign = copysignf(1.0f, from.z);
        const float a = -1.0f / (sign + from.z);
        const float b = from.x * from.y * a;
        const V3 axis = {b, sign + from.y * from.y * a, -from.y};
        const float s = sinf(3.14159265358979323846f * 0.5f), c = cosf(3.14159265358979323846f * 0.5f);
        qx = axis.x * s; qy = axis.y * s; qz = axis.z * s; qw = c;
    } else {
        const V3 c = vcross(from, up);
        const float w = 1.0f + d;
        const float l2 = (c.x * c.x + c.z * c.z) + (c.y * c.y + w * w);   // SSE2 dot4 order
        const float r = 1.0f / sqrtf(l2);
        qx = c.x * r; qy = c.y * r; qz = c.z * r; qw = w * r;
    }
    // direction = rot * (cos yaw cos pitch, sin pitch, sin yaw cos pitch)   (Quat * Vec3)
    const V3 v = {cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)};
    const V3 b = {qx, qy, qz};
    const float b2 = vdot(b, b);
    const float k0 = qw * qw - b2, k1 = vdot(v, b) * 2.0f, k2 = qw * 2.0f;
    const V3 bxv = vcross(b, v);
    const V3 f = {(v.x * k0 + b.x * k1) + bxv.x * k2, (v.y * k0 + b.y * k1) + bxv.y * k2, (v.z * k0 + b.z * k1) + bxv.z * k2};
    // Mat4::look_to_rh(eye, f, up)
    const V3 s = vnormalize(vcross(f, up));
    const V3 u = vcross(s, f);
    const float view[16] = {s.x, u.x, -f.x, 0.0f, s.y, u.y, -f.y, 0.0f, s.z, u.z, -f.z, 0.0f,
                            -vdot(eye, s), -vdot(eye, u), vdot(eye, f), 1.0f};
    // Mat4::perspective_rh(fov_y, aspect, NEAR, FAR)
    const float aspect = width / height;
    const float sf = sinf(0.5f * fov_y), cf = cosf(0.5f * fov_y);
    const float hh = cf / sf, ww = hh / aspect, r = kFar / (kNear - kFar);
    const float proj[16] = {ww, 0, 0, 0, 0, hh, 0, 0, 0, 0, r, -1.0f, 0, 0, r * kNear, 0};
    for (int c = 0; c < 4; ++c)        // proj * view, column by column: ((c0*x + c1*y) + c2*z) + c3*w
        for (int rr = 0; rr < 4; ++rr) {
            float t = proj[rr] * view[c * 4 + 0];
            t = t + proj[4 + rr] * view[c * 4 + 1];
            t = t + proj[8 + rr] * view[c * 4 + 2];
            t = t + proj[12 + rr] * view[c * 4 + 3];
            out->camera_proj[c * 4 + rr] = t;
        }
    // normal_proj = view.inverse().transpose(): no shader reads it (render_shader.wgsl:5); filled with the
    // rotation block of the view, which is what it equals for a rigid transform up to rounding.
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) out->normal_proj[c * 4 + rr] = view[c * 4 + rr];
    out->normal_proj[15] = 1.0f;
    out->camera_pos[0] = eye.x; out->camera_pos[1] = eye.y; out->camera_pos[2] = eye.z; out->camera_pos[3] = 0.0f;
    float m3[9];
    euler_xyz_ex_a0(rs_to_radians(90.0f - sun_phi_deg), rs_to_radians(sun_theta_deg), m3);
    out->sun_direction[0] = m3[6]; out->sun_direction[1] = m3[7]; out->sun_direction[2] = m3[8];   // * Vec3::Z
    out->view_mode = view_mode;
}

// The cameras of a 360-degree strip of n_sectors perspective sectors: sector k looks at yaw0 - k * 360/n
// degrees with the vertical field of view that makes every sector 360/n degrees wide.
void panorama_uniforms(const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h, float sun_theta_deg, float sun_phi_deg,
                       int32_t view_mode, uint32_t n_sectors, topo_uniforms* out) {
    const double kPi = 3.14159265358979323846;
    const double fov = 2.0 * atan(tan(kPi / (double)n_sectors) * (double)sector_h / (double)sector_w);
    for (uint32_t k = 0; k < n_sectors; ++k)
        camera_uniforms(eye, (float)((double)yaw0 - (double)k * (2.0 * kPi / (double)n_sectors)), pitch, (float)fov, (float)sector_w,
                        (float)sector_h, sun_theta_deg, sun_phi_deg, view_mode, out + k);
}

// UiController::get_locations_range (control/ui_controller.rs:61-83), f32 as in the reference.
uint32_t locations_range(float latitude, float longitude, float range_dist, int32_t* out, uint32_t cap) {
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    (void)clampi;
    // center.0 = (floor(lat) as i32).min(-90).max(89): always 89, as written
    int c_lat = (int)floorf(latitude);
    c_lat = c_lat < -90 ? c_lat : -90;
    c_lat = c_lat > 89 ? c_lat : 89;
    const int c_lon = ((int)(floorf(longitude) + 540.0f)) % 360 - 180;
    const float lat_cos = cosf(rs_to_radians(latitude));
    const float arc_factor = 0.5f * range_dist / kR0;
    const float arc_factor_sin = sinf(arc_factor);
    const float afs_sq = arc_factor_sin * arc_factor_sin;
    const float R2D = 57.29577951308232f;                       // f32::to_degrees: self * (180 / PI)
    const float dlon = acosf(1.0f - afs_sq / lat_cos / lat_cos) * R2D;
    const float dlat = acosf(1.0f - afs_sq) * R2D;
    int lat_start = (int)floorf(latitude - dlat);
    lat_start = lat_start > -90 ? lat_start : -90;
    int lat_end = (int)floorf(latitude + dlat);
    lat_end = lat_end < 89 ? lat_end : 89;
    const int lon_start = (int)floorf(longitude - dlon), lon_end = (int)floorf(longitude + dlon);
    struct Item { int lat, lon, k0, k1; };
    std::vector<Item> v;
    for (int la = lat_start; la <= lat_end; ++la)
        for (int lo = lon_start; lo <= lon_end; ++lo) v.push_back({la, lo, std::abs(la - c_lat), std::abs(lo - c_lon)});
    std::stable_sort(v.begin(), v.end(), [](const Item& a, const Item& b) { return a.k0 != b.k0 ? a.k0 < b.k0 : a.k1 < b.k1; });
    uint32_t n = 0;
    for (const Item& it : v) {
        if (n < cap && out) { out[2 * n] = it.lat; out[2 * n + 1] = (it.lon + 540) % 360 - 180; }
        ++n;
    }
    return n;
}

// UiController::change_location (control/ui_controller.rs:23-59) as a plan: the tiles of get_locations_range(location,
// range) that are not loaded yet are to be requested, the loaded ones outside it are to be unloaded.  The reference walks
// HashSets (order unspecified); here both lists come out in a defined order: `request` in get_locations_range's sorted
// order, `unload` in the order of `loaded`.
void change_location_plan(float latitude, float longitude, float range_dist, const int32_t* loaded, uint32_t n_loaded,
                          std::vector<std::pair<int32_t, int32_t>>& unload, std::vector<std::pair<int32_t, int32_t>>& request) {
    const uint32_t n = locations_range(latitude, longitude, range_dist, nullptr, 0);
    std::vector<int32_t> want(2 * (size_t)n);
    locations_range(latitude, longitude, range_dist, want.data(), n);
    std::vector<bool> have(n, false);
    unload.clear();
    request.clear();
    for (uint32_t i = 0; i < n_loaded; ++i) {
        bool in_new = false;
        for (uint32_t k = 0; k < n; ++k)
            if (want[2 * k] == loaded[2 * i] && want[2 * k + 1] == loaded[2 * i + 1]) { in_new = true; have[k] = true; }
        if (!in_new) unload.emplace_back(loaded[2 * i], loaded[2 * i + 1]);
    }
    for (uint32_t k = 0; k < n; ++k) {
        bool dup = false;       // (the range wraps at 180 degrees: a location can appear twice; a HashSet holds it once)
        for (uint32_t j = 0; j < k && !dup; ++j) dup = want[2 * j] == want[2 * k] && want[2 * j + 1] == want[2 * k + 1];
        if (!have[k] && !dup) request.emplace_back(want[2 * k], want[2 * k + 1]);
    }
}

int TerrainRenderer::change_location(float latitude, float longitude, float range_dist, std::vector<std::pair<int32_t, int32_t>>& request,
                                     uint32_t* n_unloaded) {
    std::vector<int32_t> loaded;
    for (const auto& kv : tiles_) { loaded.push_back(kv.second.lat); loaded.push_back(kv.second.lon); }
    std::vector<std::pair<int32_t, int32_t>> unload;
    change_location_plan(latitude, longitude, range_dist, loaded.data(), (uint32_t)(loaded.size() / 2), unload, request);
    for (const auto& u : unload)
        if (int rc = unload_terrain(u.first, u.second)) return rc;
    if (n_unloaded) *n_unloaded = (uint32_t)unload.size();
    return TOPO_OK;
}

// Synthetic COP90-shaped heights: 5-octave value-noise fBm over global texel coordinates with an integer
// hash (same definition as topo-renderer_amd/synth.py; f32 ops in the same order).
namespace {
inline float synth_hash(int64_t ix, int64_t iy, uint32_t seed) {
    uint32_t h = ((uint32_t)ix * 0x9E3779B1u) ^ ((uint32_t)iy * 0x85EBCA77u) ^ (seed * 0xC2B2AE3Du);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
}  // namespace

void synth_tile(int32_t lat, int32_t lon, uint32_t w, uint32_t h, uint32_t seed, float* out) {
    static const int wl[5] = {512, 256, 128, 64, 32};
    static const float amp[5] = {1.0f, 0.5f, 0.25f, 0.125f, 0.0625f};
    const float norm = (float)(3000.0 / 1.9375);
    for (uint32_t y = 0; y < h; ++y) {
        const int64_t gy = (int64_t)(89 - lat) * h + y;
        for (uint32_t x = 0; x < w; ++x) {
            const int64_t gx = ((int64_t)lon + 180) * w + x;
            float acc = 0.0f;
            for (int o = 0; o < 5; ++o) {
                const int64_t cx = gx / wl[o], cy = gy / wl[o];
                const float fx = (float)(gx % wl[o]) / (float)wl[o], fy = (float)(gy % wl[o]) / (float)wl[o];
                const float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy);
                const uint32_t s = seed + (uint32_t)o;
                const float v00 = synth_hash(cx, cy, s), v10 = synth_hash(cx + 1, cy, s);
                const float v01 = synth_hash(cx, cy + 1, s), v11 = synth_hash(cx + 1, cy + 1, s);
                const float a = v00 + ux * (v10 - v00), b = v01 + ux * (v11 - v01);
                const float v = a + uy * (b - a);
                acc = acc + amp[o] * v;
            }
            out[(size_t)y * w + x] = acc * norm;
        }
    }
}

}  // namespace topo
