// terrain_renderer.hpp -- host side of the MI355X terrain path: a C++ mirror of the reference's
// `TerrainRenderer` (topo-renderer/src/render/terrain_renderer.rs) with the same five methods
// (new / update / add_terrain / unload_terrain / render).  The reference's host code is Rust; no Rust
// toolchain exists in this image, so the host side above the C ABI is C++ (include/topo_hip.h wraps this
// class one-to-one; INTEGRATION.md has the Rust shim).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/topo_hip.h"
#include "topo_kernels.h"

namespace topo {

constexpr uint32_t kPanoramaSectors = 8;      // fixed, independent of the GPU count, so the strip is the same for every N
struct Comm;                                  // panorama.cpp: an RCCL communicator + this process's rank

// BTreeMap<GeoLocation, _> key: Ord over {latitude{degree,direction S<N}, longitude{degree,direction W<E}}
// (topo-common/src/lib.rs:7-38); GeoLocation::from_coord maps sign > 0 to N/E, else S/W (:102-121).
using GeoKey = std::tuple<int, int, int, int>;
inline GeoKey geo_key(int lat, int lon) {
    return GeoKey(lat < 0 ? -lat : lat, lat > 0 ? 1 : 0, lon < 0 ? -lon : lon, lon > 0 ? 1 : 0);
}

struct Tile {          // RenderBuffer (render_buffer.rs:23-31) minus the wgpu plumbing
    int lat = 0, lon = 0;
    uint64_t seq = 0;  // insertion order (for topo_recompute_normals)
    float* d_heights = nullptr;
    uint32_t* d_normals = nullptr;
    float* d_minmax = nullptr;
    void* d_pool = nullptr;         // the one allocation the three pointers above point into
    // a sphere around the centres of the tile's block spheres (the cull's load-time table, read back once per tile): lets a
    // submission prove on the host that none of its blocks can be an occlusion-test candidate.  radius < 0: unknown.
    double centres[4] = {0.0, 0.0, 0.0, -1.0};
    TileDev dev{};
};

class TerrainRenderer {
   public:
    static int create(TerrainRenderer** out, int device, uint32_t w, uint32_t h, uint32_t format, std::string* err);
    ~TerrainRenderer();

    int add_terrain(int32_t lat, int32_t lon, const float* heights, bool heights_on_device, uint32_t w, uint32_t h,
                    const float rp[2], const float mp[2], const float ps[2]);
    int unload_terrain(int32_t lat, int32_t lon);
    bool last_far_phase() const { return last_far_phase_; }
    int update(uint32_t w, uint32_t h, const topo_uniforms* u, const topo_post_uniforms* pu);
    int render(uint8_t* rgba, size_t rgba_pitch, float* depth, size_t depth_pitch);
    int render_views_device(uint32_t n, const topo_uniforms* views, uint32_t w, uint32_t h, const OutputParams& out);
    int render_device(uint8_t* rgba_dev, size_t rgba_pitch, float* depth_dev, size_t depth_pitch);
    int render_panorama(Comm* comm, const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h, float sun_theta_deg,
                        float sun_phi_deg, int32_t view_mode, uint8_t* strip_dev, float* depth_dev);
    int render_batch(uint32_t n_viewpoints, const float* eyes, const float* yaw0s, const float* sun_theta_phi_deg, float pitch, uint32_t sector_w,
                     uint32_t sector_h, int32_t view_mode, uint8_t* rgba_dev, float* depth_dev);
    int recompute_normals();
    int overlay_lines_device(const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, float line_width, uint8_t* rgba_dev,
                             size_t rgba_pitch);
    int overlay_glyphs(const void* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t atlas_w, uint32_t atlas_h, uint8_t* rgba, size_t rgba_pitch);
    int overlay_glyphs_device(const void* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t atlas_w, uint32_t atlas_h, uint8_t* rgba_dev,
                              size_t rgba_pitch);
    int overlay_lines(const void* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, float line_width, uint8_t* rgba,
                      size_t rgba_pitch);
    int change_location(float latitude, float longitude, float range_dist, std::vector<std::pair<int32_t, int32_t>>& request, uint32_t* n_unloaded);

    int set_stream(hipStream_t s);
    int synchronize();
    int set_pipeline_depth(int depth);
    int join();                 // waits for the frames in flight on the contexts' own streams
    int join_frames();          // topo_join: join + the frames' status (TOPO_ERR_CAPACITY for an incomplete frame)
    int set_normals_lds_rows(int rows);
    int set_queue_caps(uint32_t big_cap, uint32_t rare_cap);
    int pin_host_buffer(void* p, size_t bytes);
    int unpin_host_buffer(void* p);
    int get_timings(float out[TOPO_TIMING_SLOTS]);
    int get_timing_history(uint32_t n_frames, float* out_ms, uint32_t* n_out);
    int get_counters(uint32_t out[6]);
    int frame_status(uint32_t out[4]);
    int set_occlusion_split(float metres);
    int set_timing_slots(uint32_t mask);
    int read_normals(int32_t lat, int32_t lon, uint8_t* out);
    int read_tile_tables(int32_t lat, int32_t lon, float* minmax_out, float* trig_out, double* bounds_out, uint32_t* n_blocks_out);      // test hook
    int geotiff_to_device(const uint8_t* bytes, size_t n, float** d_heights, uint32_t* w, uint32_t* h, float rp[2], float mp[2], float ps[2]);
    int geotiff_decode(const uint8_t* bytes, size_t n, float* heights_out, size_t capacity);
    int add_terrain_geotiff(int32_t lat, int32_t lon, const uint8_t* bytes, size_t n);
    int probe_sincos(const float* x, float* s, float* c, size_t n);
    int probe_div(int32_t kind, const float* x, const float* y, float* out, size_t n);
    int visible_peaks(uint32_t n, const float* peaks, uint8_t* visible, uint32_t* xy);
    int visible_peaks_device(const topo_uniforms* view, uint32_t w, uint32_t h, const float* depth_dev, size_t depth_pitch,
                             uint32_t n, const float* peaks_dev, uint8_t* visible_dev, uint32_t* xy_dev);

    const char* last_error() const { return err_.c_str(); }

   private:
    TerrainRenderer() = default;
    int fail(int code, const std::string& msg);
    int hip_fail(hipError_t e, const char* what);
    int bind_device();
    int ensure(void** p, size_t* cap, size_t need);
    void collect_jobs(const Tile& nt, const std::map<GeoKey, uint32_t>& rank, std::vector<EdgeJob>& edges,
                      std::vector<CornerJob>& corners);   // the seam/corner orchestration of add_terrain
    int run_seam_jobs(const std::vector<EdgeJob>& edges, const std::vector<CornerJob>& corners);
    int upload_seam_jobs(const std::vector<EdgeJob>& edges, const std::vector<CornerJob>& corners);
    void launch_seam_jobs(size_t n_edges, size_t n_corners);
    void launch_load_kernels(uint32_t first, uint32_t count, hipEvent_t mid);      // tables + interior normals of a run of tiles
    std::map<GeoKey, uint32_t> ranks() const;
    Tile* find(int lat, int lon);
    int upload_tile_table();

    int device_ = 0;
    uint32_t W_ = 0, H_ = 0;
    uint32_t format_ = TOPO_FORMAT_RGBA8_UNORM_SRGB;
    uint32_t tile_w_ = 0, tile_h_ = 0;
    topo_uniforms uniforms_{};
    topo_post_uniforms post_{{0.0f, 0.0f}, 100.0f, 0.0f};      // (pixelize_n 100: the branch off, as the reference always has it)
    bool have_uniforms_ = false;
    std::map<GeoKey, Tile> tiles_;
    uint64_t next_seq_ = 1;
    bool table_dirty_ = true;
    int lds_rows_ = 0;        // interior normals: 0 = without an LDS tile (k_normals_rolling: the sweep's optimum, profiles/r03_normals_lds_sweep.json;
                              // needs a tile width that is a multiple of four, else the LDS form with 32 rows); 4..64 = LDS tile height of k_normals_interior
    uint32_t big_cap_cfg_ = 0, rare_cap_cfg_ = 0;
    uint64_t rare_cap_auto_ = 0;           // rare-queue capacity topo_render grew to after an overflow (0 = default)
    uint32_t timing_slots_ = 0x3Fu;        // topo_set_timing_slots: which per-kernel durations to measure
    bool timing_total_ = true;        // ev[0] and ev[8] (TOPO_TIMING_NO_TOTAL clears it)
    float occlusion_split_m_ = 90000.0f;   // flat optimum 60..120 km at c4 (profiles/README.md)

    hipStream_t own_stream_ = nullptr, stream_ = nullptr;
    static constexpr int kNumEvents = 9;
    static constexpr int kEvRing = 32;
    uint64_t frame_seq_ = 0;          // frames submitted by this renderer
    hipEvent_t load_ev_[3] = {};      // recompute_normals: start, end, between the tables and the normals
    bool load_timed_ = false;

    // Everything one frame in flight owns.  With pipeline depth 1 (default) there is one context and it runs on
    // stream_; with depth d > 1 topo_render_views_device rotates through d contexts, each on a stream of its own, so
    // that the memory-latency-bound cull/raster phases of one frame run under the ALU-bound resolve of the previous one.
    struct FrameCtx {
        hipStream_t stream = nullptr;        // own stream (depth > 1)
        // timing events of the last kEvRing frames of this context (a frame's durations stay readable while later frames are
        // submitted: topo_get_timing_history reads a whole timed region's frames after it, without a wait inside it)
        hipEvent_t evr[kEvRing][kNumEvents] = {};
        uint32_t evr_recorded[kEvRing] = {}, evr_slots[kEvRing] = {};
        uint64_t evr_frame[kEvRing] = {};     // the renderer-wide number of the frame that used the set
        uint64_t frames = 0;                  // frames submitted on this context
        hipEvent_t done = nullptr;
        bool timed = false, pending = false;
        // pinned ring of the last kStatusRing frames' 16 counter words, each stored by its frame's k_resolve (the bounds-checking build: copied out behind it);
        // frames [checked, submitted) have not been looked at by check_frames yet
        uint32_t* h_status = nullptr;
        uint64_t submitted = 0, checked = 0;
        void* d_vis = nullptr;      size_t cap_vis = 0;
        void* d_dirty = nullptr;    size_t cap_dirty = 0;   // one mark per 64 visibility keys (topo_kernels.hip: struct Vis)
        void* d_work = nullptr;     size_t cap_work = 0;
        void* d_work2 = nullptr;    size_t cap_work2 = 0;
        void* d_far = nullptr;      size_t cap_far = 0;
        void* d_big = nullptr;      size_t cap_big = 0;
        void* d_rare = nullptr;     size_t cap_rare = 0;
        void* d_counters = nullptr; size_t cap_counters = 0;
    };
    static constexpr int kMaxPipeline = 4;
    static constexpr uint64_t kStatusRing = 64;
    FrameCtx ctx_[kMaxPipeline];
    int pipeline_depth_ = 1, next_ctx_ = 0, last_ctx_ = 0;
    int init_ctx(FrameCtx& c, bool own_stream);
    int check_frames();                        // after a wait: turn a finished frame's overflow status into TOPO_ERR_CAPACITY
    uint32_t last_status_[4] = {};             // status word + bounds record of the last frame looked at
    bool fold_frames(FrameCtx& c);             // folds the finished, unchecked frames of c into last_status_; true if one overflowed
    bool overflow_pending_ = false;
    int ensure_on(hipStream_t s, void** p, size_t* cap, size_t need);
    // slots: the frame resolved in several launches (k_resolve over block ranges), after_slot(i, stream) called behind each --
    // null / 0: one launch over the whole frame
    struct ResolveSlot { uint32_t block_first, block_count; };
    int render_frame(FrameCtx& c, hipStream_t s, uint32_t n, const topo_uniforms* views, uint32_t w, uint32_t h, const OutputParams& out,
                     const ResolveSlot* slots = nullptr, uint32_t n_slots = 0, const std::function<int(uint32_t, hipStream_t)>* after_slot = nullptr);
    int frame_durations(FrameCtx& c, int ring, float out[7]);

    // grow-only device buffers
    void* d_tiles_ = nullptr;    size_t cap_tiles_ = 0;
    void* d_views_ = nullptr;    size_t cap_views_ = 0;
    void* d_edge_jobs_ = nullptr;   size_t cap_edge_jobs_ = 0;
    void* d_corner_jobs_ = nullptr; size_t cap_corner_jobs_ = 0;
    void* d_out_rgba_ = nullptr; size_t cap_out_rgba_ = 0;
    void* d_out_depth_ = nullptr; size_t cap_out_depth_ = 0;
    void* d_pre_rgba_ = nullptr; size_t cap_pre_rgba_ = 0;      // the pixelise branch: the render-target image k_post_pixelize samples,
    void* d_pre_depth_ = nullptr; size_t cap_pre_depth_ = 0;    // and a depth image when the caller wants none
    // topo_render's way out to host memory: a pinned staging image and the events of its slices; the buffers the caller pinned
    uint8_t* h_stage_ = nullptr; size_t cap_stage_ = 0;
    hipEvent_t stage_ev_[8] = {};
    std::vector<std::pair<uint8_t*, size_t>> pinned_;
    int download(uint8_t* dst, size_t dst_pitch, const uint8_t* src_dev, size_t row);
    static constexpr int kViewSlots = 16;
    static constexpr uint32_t kMaxViewsPerSlot = 64;
    ViewDev* h_views_ = nullptr;          // pinned staging ring
    hipEvent_t view_ev_[kViewSlots] = {};
    bool view_used_[kViewSlots] = {};
    void* d_peaks_ = nullptr;    size_t cap_peaks_ = 0;      // xyz in, then visible + xy out
    void* d_proj_ = nullptr;     size_t cap_proj_ = 0;
    void* d_overlay_geo_ = nullptr;  size_t cap_overlay_geo_ = 0;     // overlay vertices + indices
    void* d_overlay_keys_ = nullptr; size_t cap_overlay_keys_ = 0;    // W*H overlay keys (depth | ~triangle), kept at the post quad's depth between calls
    uint32_t overlay_w_ = 0, overlay_h_ = 0;
    bool have_depth_ = false;
    uint32_t depth_w_ = 0, depth_h_ = 0;
    uint32_t last_blocks_tested_ = 0;
    bool last_far_phase_ = true;           // whether the last submission launched the far phase (test hook)

    std::string err_;
};

// host-side restatements of the reference's CPU math (glam 0.31)
void camera_uniforms(const float eye[3], float yaw, float pitch, float fov_y, float width, float height,
                     float sun_theta_deg, float sun_phi_deg, int32_t view_mode, topo_uniforms* out);
void panorama_uniforms(const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h, float sun_theta_deg, float sun_phi_deg,
                       int32_t view_mode, uint32_t n_sectors, topo_uniforms* out);
int comm_unique_id(uint8_t out[128], std::string* err);
int comm_init(Comm** out, int device, const uint8_t id128[128], int rank, int world, std::string* err);
int comm_from_nccl(Comm** out, void* nccl_comm, int rank, int world, std::string* err);
void comm_destroy(Comm* c);
void panorama_sector_range(int rank, int world, uint32_t* first, uint32_t* count);
uint32_t panorama_slots(int world, uint32_t sector_w, uint32_t sector_h, topo_panorama_slot* out, uint32_t cap);
void geometry_transform(float h, float lon_deg, float lat_deg, float out[3]);
void terrain_rotation(float model_lon_deg, float model_lat_deg, float rot3x3_colmajor[9]);
uint32_t locations_range(float latitude, float longitude, float range_dist, int32_t* out_lat_lon, uint32_t cap);
void change_location_plan(float latitude, float longitude, float range_dist, const int32_t* loaded, uint32_t n_loaded,
                          std::vector<std::pair<int32_t, int32_t>>& unload, std::vector<std::pair<int32_t, int32_t>>& request);
void synth_tile(int32_t lat, int32_t lon, uint32_t w, uint32_t h, uint32_t seed, float* out);

}  // namespace topo
