// topo_kernels.h -- launch interface of the gfx950 kernels (topo_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "topo_pipeline.h"

namespace topo {

// Raster work decomposition: a tile's (w-1) x (h-1) cell grid is cut into blocks of kBCX x kBCY cells
// ((kBCX+1) x (kBCY+1) vertices staged in LDS by one 256-thread workgroup).  1199 = 19*60+59 = 79*15+14,
// so a 1200x1200 COP90 tile gives 20 x 80 blocks with no sliver blocks.
constexpr uint32_t kBCX = 60, kBCY = 15;
constexpr uint32_t kVX = kBCX + 1, kVY = kBCY + 1;
#ifndef TOPO_RESOLVE_RPW
#define TOPO_RESOLVE_RPW 8      // pixel rows per wave of k_resolve (4 or 8)
#endif
constexpr uint32_t kResolveBlockW = 64, kResolveBlockH = 4 * TOPO_RESOLVE_RPW;   // k_resolve's pixel blocks: four waves, a 64 x RPW strip each

struct WorkItem {          // one (view, tile, block) that survived the frustum cull
    uint32_t view_rank;    // view << 16 | tile rank (draw order)
    uint32_t block;        // by * bx_count + bx
};

struct FarItem {           // a frustum-surviving block beyond the occlusion split, with its conservative screen footprint
    uint32_t view_rank;
    uint32_t block;
    uint16_t x0, x1, y0, y1;   // inclusive pixel box of everything the block can draw (already clamped to the target)
    uint32_t zmin_bits;        // f32 bits of a lower bound of every depth the block can produce
};

struct BigItem {           // a triangle too large for the in-lane loop, restricted to one 64x64 px region
    uint32_t view;
    uint32_t id;           // draw << 1 | fan   (kNoTri = empty slot)
    uint32_t region;       // ry << 16 | rx  (64 px units)
    int32_t X[3], Y[3];    // the snapped vertices: the consumer re-runs the exact integer setup on them
    float z[3];
};

struct RareItem {          // a triangle the lean raster kernel does not handle itself: >= 64 px across, or near-clipped
    uint32_t view;
    uint32_t draw;         // tile rank * tris_per_tile + triangle
};

struct FrameParams {
    const TileDev* tiles;      // n_tiles entries in draw order (BTreeMap order, terrain_renderer.rs:407)
    const ViewDev* views;      // n_views
    uint64_t* vis;             // n_views * W * H visibility keys: depth bits << 32 | id
    uint8_t* dirty;            // one byte per 64 consecutive keys: 1 = some key of the segment was written this frame
    WorkItem* work;
    uint32_t* counters;        // [0] near work count, [1] big count, [2] status bits, [3] rare count, [4] far candidates,
                               // [5] far survivors, [6] big start, [7] rare start (of the current phase), [8..10] first
                               // bounds violation (check build); 16 words; two sets per frame context, alternating: a frame's set was zeroed by the clear of the frame before
    uint32_t* status_out;      // pinned host memory, or null: k_resolve's first workgroup stores the frame's 16 counter words there (nothing
                               // changes them once the raster kernels are done), instead of a copy operation behind the frame
    BigItem* big;
    RareItem* rare;
    FarItem* far;              // far candidates (k_cull -> k_occlusion)
    WorkItem* work2;           // far survivors (k_occlusion -> second k_raster)
    uint32_t work_cap, big_cap, rare_cap;   // work_cap: entries of work2 (one per block and view)
    uint32_t far_sub_cap;                   // entries of each of the kFarLists sub-lists of `far` (enough for every candidate its workgroups can produce)
    uint32_t near_cap;                      // entries of work / work2 (ceil(15 / near_strip) strips per block)
    uint32_t near_strip;                    // cell rows per strip of a near block / far survivor (1 .. 15)
    float split_m;             // view depth (m) beyond which a block is an occlusion-test candidate; 0 = feature off
    uint32_t n_views, n_tiles;
    int32_t W, H;
    uint32_t tile_w, tile_h;
    uint32_t bx_count, by_count;
    uint32_t tris_per_tile;
    FastDiv div_tris, div_hm1;   // exact division by tris_per_tile / (tile_h - 1)
    uint32_t sky_c8;           // the cleared render-target texel (clear colour encoded for the target format), R G B A from the low byte
    uint32_t linear_target;    // 1: plain *Unorm targets (no sRGB encode/decode); 0: *UnormSrgb
    uint32_t bgra;             // 1: output texels are B G R A in memory
    uint32_t rblocks_x, rblocks_view;             // k_resolve's 64 x (4 RPW) px blocks: per row of a view, per view
    uint32_t post_off;         // 1: k_resolve stores the RENDER TARGET texel (no post pass), always R G B A -- the pixelise branch
                               // samples that image in a pass of its own (k_post_pixelize)
    uint32_t rblock_first, rblock_count;          // the blocks THIS launch of k_resolve shades (a frame can be resolved in several
                                                  // launches -- by view and band of rows -- so that an exchange of the finished part
                                                  // runs under the rest: topo_render_panorama)
    FastDiv div_rblocks_x, div_rblocks_view;
};

struct OutputParams {
    uint8_t* rgba;             // device pointer
    size_t rgba_view_stride, rgba_pitch;   // bytes
    float* depth;              // device pointer or null
    size_t depth_view_stride, depth_pitch; // bytes
};

constexpr uint32_t kStatusBigOverflow = 1u;    // big-triangle queue full: handled in-lane (slower, still exact)
constexpr uint32_t kStatusRareOverflow = 2u;   // rare-triangle queue full: triangles were DROPPED -> the frame is invalid
constexpr uint32_t kFarLists = 64;             // the far-candidate list is kept as 64 sub-lists, each with a counter on a cache line of its own
constexpr uint32_t kCounterWords = 16 + 16 * kFarLists;      // queue counters and status words of one frame, then the sub-list counters (word 16 + 16 q)
constexpr uint32_t kStatusBounds = 4u;         // TOPO_BOUNDS_CHECK build only: an out-of-range index was formed (and not used);
                                               // counters[8] = site tag, counters[9..10] = the offending value

// load phase (add_terrain).  Every pass of the reference's add_terrain writes a disjoint set of texels (interior
// / one seam per adjacent pair / one corner per 2x2 block), so any number of them can run in one launch each;
// jobs name tiles by their index in the device tile table.
struct EdgeJob {           // compute_normals_left_right / _top_bottom (compute_normals_edge_shader.wgsl)
    uint32_t lt, rb;       // left-or-top tile, right-or-bottom tile
    uint32_t uni;          // the tile whose uniforms the pass reads: the one that was inserted last
    uint32_t top_bottom;
};
struct CornerJob {         // compute_normals_corner (compute_normals_corner_shader.wgsl)
    uint32_t lt, rt, lb, rb;
    uint32_t uni;
};

void launch_block_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s);   // block min/max, cull bounds, sin/cos tables of `count` tiles
// The load path that reads the DEM once, taken when normals_tables_fused(): launch_trig_tables -> launch_normals_tables (interior
// normals AND the block minima / maxima) -> launch_block_bounds (the f64 cull bounds from them) [-> launch_normals_border].
bool normals_tables_fused(uint32_t w, uint32_t h, int lds_rows);
void launch_trig_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s);
void launch_normals_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s);
void launch_block_bounds(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s);
// interior normals of tiles[first .. first+count); also zeroes their border ring (fresh Rgba8Unorm texture)
void launch_normals_interior(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, int lds_rows,
                             hipStream_t s);
void launch_normals_border(const TileDev* tiles, const EdgeJob* edges, uint32_t n_edges, const CornerJob* corners, uint32_t n_corners, uint32_t w,
                           uint32_t h, hipStream_t s);      // the seam and corner passes of any number of jobs, one launch

// frame phase (render)
// `zero`: the counter set of the NEXT frame (the two sets of a frame context alternate), zeroed by the clear
// A submission's view constants travel as a kernel argument (up to kPackViews views = 704 bytes): the launch copies them, so the
// caller's array is free at once and no staging slot, copy-engine operation or event stands in front of the frame.
constexpr uint32_t kPackViews = 8;
struct ViewPack { ViewDev v[kPackViews]; };
void launch_put_views(const ViewPack& pack, uint32_t n, ViewDev* dst, hipStream_t s);
// (`start` / `stop`, where a launcher has them: events that take the kernel's own start / end time -- hipExtLaunchKernel: the
// dispatch's completion signal carries both, no marker packet stands between two kernels)
void launch_clear(const FrameParams& p, uint32_t* zero, hipStream_t s, hipEvent_t start = nullptr);        // re-initialises the marked segments
void launch_clear_cull(const FrameParams& p, uint32_t* zero, hipStream_t s, hipEvent_t start = nullptr, const ViewPack* pack = nullptr, uint32_t n_pack_views = 0);   // the clear and the cull in one launch, side by side
void launch_cull(const FrameParams& p, hipStream_t s);
void launch_raster(const FrameParams& p, int phase, hipStream_t s);   // phase 0: near list, 1: far survivors
void launch_occlusion(const FrameParams& p, hipStream_t s);
void launch_raster_rare(const FrameParams& p, hipStream_t s);
void launch_raster_big(const FrameParams& p, hipStream_t s);
void launch_resolve(const FrameParams& p, const OutputParams& o, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

// overlay pass (line_shader.wgsl over the post pass's image): keys = W*H overlay keys, (re-)initialised when keys_fresh
void launch_overlay(const OverlayVertex* verts, const uint32_t* idx, uint32_t n_tris, uint32_t n_verts, float width, int32_t W, int32_t H, uint64_t* keys,
                    bool keys_fresh, uint8_t* rgba, size_t pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s);

// depth consumer (RenderEngine::get_visible_labels, render_engine.rs:338-396)
void launch_overlay_glyphs(const GlyphInstance* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t aw, uint32_t ah, int32_t W, int32_t H,
                           uint64_t* keys, bool keys_fresh, uint8_t* rgba, size_t pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s);
void launch_post_pixelize(uint32_t n_views, int32_t W, int32_t H, float vw, float vh, float pixelize_n, const uint8_t* pre_rgba, const OutputParams& out,
                          const float* depth, size_t depth_view_stride, size_t depth_pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s);
void launch_visible_peaks(const float* proj16_dev, uint32_t w, uint32_t h, const float* depth, size_t depth_pitch, uint32_t n,
                          const float* peaks_xyz, uint8_t* visible, uint32_t* xy, hipStream_t s);

// unit-test probes
// GeoTIFF decode, device half: one workgroup per stored row of a strip/tile (geotiff.hpp).
struct TiffSegDev {
    uint64_t byte_off;       // of the segment's decompressed bytes in the staging buffer
    uint32_t row0;           // index of its first row in the global row list
    uint32_t x0, y0, w, h;   // position in the image, stored size
    uint32_t pad_;
};
void launch_tiff_rows(uint8_t* bytes, const TiffSegDev* segs, const uint32_t* row_seg, uint32_t n_rows, float* out, uint32_t W, uint32_t H,
                      uint32_t predictor, bool big_endian, hipStream_t s);
void launch_probe_sincos(const float* x, float* s, float* c, size_t n, hipStream_t st);
void launch_probe_div(int kind, const float* x, const float* y, float* out, size_t n, hipStream_t st);

}  // namespace topo
