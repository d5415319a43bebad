/*
 * topo_hip.h -- C ABI of libtopo_hip.so: the MI355X-native terrain render path of krzyz/topo-renderer.
 *
 * The reference has no FFI or plugin interface; the seam this library plugs into is the inherent-method
 * surface of `TerrainRenderer` (topo-renderer/src/render/terrain_renderer.rs), called only from
 * `RenderEngine` (render_engine.rs:143,160-166,183-189,212-214,276-284).  Each entry point below names the
 * reference method it replaces; INTEGRATION.md shows the Rust `extern "C"` shim that binds them.
 *
 * Conventions: every function returns TOPO_OK (0) or a negative topo_status; topo_last_error() gives the
 * text.  No exception crosses the boundary.  A context is thread-affine like the reference (all calls from
 * the one event-loop thread; the reference keeps its shared mesh buffers in thread_local!,
 * render_buffer.rs:12-15).  The caller keeps ownership of every input pointer; outputs are caller-allocated.
 * There is no CPU fallback: with no HIP device every compute call fails with TOPO_ERR_HIP.
 */
#ifndef TOPO_HIP_H
#define TOPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct topo_ctx topo_ctx;

typedef enum topo_status {
    TOPO_OK = 0,
    TOPO_ERR_INVALID = -1,     /* bad argument (null pointer, zero size, mixed tile sizes, ...) */
    TOPO_ERR_UNSUPPORTED = -2, /* valid in the reference but outside this path (pixelize_n < 99.99999, format) */
    TOPO_ERR_HIP = -3,         /* a HIP runtime call failed (no device, out of memory, launch failure) */
    TOPO_ERR_NOT_FOUND = -4,   /* no such tile */
    TOPO_ERR_CAPACITY = -5     /* draw-order id space exhausted (too many / too large tiles), or a frame overflowed its
                                * rare-triangle queue and is incomplete (see topo_join) */
} topo_status;

/* The surface format the reference picks: `surface_caps.formats[0]`, with the sRGB suffix when the surface also offers
 * that variant (render_engine.rs:77-84); render target and final target share it (terrain_renderer.rs:88-93).  The *Srgb
 * formats encode linear -> sRGB8 on store and decode on sample (so the pipeline encodes twice and decodes once in
 * between); the plain formats store round(clamp(v) * 255) and sample c / 255.  Bgra differs from Rgba
 * only in the byte order of the output texels (B G R A in memory). */
#define TOPO_FORMAT_RGBA8_UNORM_SRGB 1u
#define TOPO_FORMAT_BGRA8_UNORM_SRGB 2u
#define TOPO_FORMAT_RGBA8_UNORM 3u
#define TOPO_FORMAT_BGRA8_UNORM 4u

/* `Uniforms`, #[repr(C)], 160 bytes: topo-renderer/src/render/data.rs:33-41 (WGSL mirror render_shader.wgsl:3-9).
 * Matrices are glam column-major. */
typedef struct topo_uniforms {
    float camera_proj[16]; /* projection * view (camera.rs:122-128) */
    float normal_proj[16]; /* uploaded by the reference, read by no shader */
    float camera_pos[4];   /* eye, w = 0; only .xy reaches the output (dither seed) */
    float sun_direction[3];
    int32_t view_mode;     /* 0 lit + dither, 1 lit, 2 normals (render_shader.wgsl:108-114) */
} topo_uniforms;

/* `PostprocessingUniforms`, 16 bytes: render/data.rs:74-80 */
typedef struct topo_post_uniforms {
    float viewport[2];
    float pixelize_n; /* the reference always passes 100.0 (application_data.rs:31); < 99.99999 is rejected */
    float _padding;
} topo_post_uniforms;

/* ---- the TerrainRenderer surface ---------------------------------------------------------------------- */

/* TerrainRenderer::new(device, format, target_size)                        terrain_renderer.rs:37-69 */
int topo_create(topo_ctx** out, int hip_device, uint32_t width, uint32_t height, uint32_t color_format);
void topo_destroy(topo_ctx* ctx);

/* TerrainRenderer::add_terrain(.., location, height_map_data, coordinate_transform, size, ..)
 *                                                                           terrain_renderer.rs:173-350
 * heights: w*h little-endian f32, row-major, row 0 = north (RenderBuffer::new, render_buffer.rs:76-90), host
 * memory, borrowed for the call.  raster_point/model_point/pixel_scale: CoordinateTransform
 * (common/coordinate_transform.rs:16-20).  Uploads, computes the interior normals and the seam/corner normals
 * against the already loaded neighbours exactly as the reference orchestrates them.  Returns when the copy of
 * `heights` is complete (work stays queued on the context's stream). */
int topo_add_terrain(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg, const float* heights, uint32_t w, uint32_t h,
                     const float raster_point[2], const float model_point[2], const float pixel_scale[2]);

/* TerrainRenderer::unload_terrain(&location)                                terrain_renderer.rs:361-363 */
int topo_unload_terrain(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg);

/* TerrainRenderer::update(device, queue, target_size, &uniforms, &postprocessing_uniforms)
 *                                                                           terrain_renderer.rs:151-171 */
int topo_update(topo_ctx* ctx, uint32_t width, uint32_t height, const topo_uniforms* uniforms,
                const topo_post_uniforms* post);

/* TerrainRenderer::render(target, encoder, viewport) + the depth copy of RenderEngine::render
 *                                               terrain_renderer.rs:365-452, render_engine.rs:219-249
 * rgba_out: height rows of width RGBA8 (sRGB-encoded) texels, rgba_pitch bytes apart.  depth_out (nullable):
 * Depth32Float rows depth_pitch bytes apart; the reference uses pad_256(4*width) (data/mod.rs:9-11).
 * Host pointers; synchronous. */
int topo_render(topo_ctx* ctx, uint8_t* rgba_out, size_t rgba_pitch, float* depth_out, size_t depth_pitch);
/* The way out to host memory.  The reference presents its frame without a read-back and maps the depth buffer only when the
 * camera moved (render_engine.rs:219-252); a host that asks topo_render for host images gets them at the link's rate if it
 * pins the buffers it reuses frame after frame (hipHostRegister under the hood; unpin before freeing them):
 * topo_render then copies straight into them.  Any other buffer is filled through a pinned staging image of the context's
 * own, slice by slice (the device copies slice k + 1 while host threads move slice k to the caller's rows) -- several times
 * the rate of a plain copy into pageable memory, below the pinned one.  depth_out == NULL skips the depth copy. */
int topo_pin_host_buffer(topo_ctx* ctx, void* buffer, size_t bytes);
int topo_unpin_host_buffer(topo_ctx* ctx, void* buffer);

const char* topo_last_error(topo_ctx* ctx);

/* ---- additive entry points (no reference counterpart) ------------------------------------------------- */

/* topo_render with device outputs: the frame of the last topo_update into rgba_dev / depth_dev (depth nullable), in stream
 * order on the context's stream; no host copy.  Asynchronous: the frame's status is reported by the call that waits
 * for it (topo_join / topo_synchronize, or topo_frame_status). */
int topo_render_device(topo_ctx* ctx, uint8_t* rgba_dev, size_t rgba_pitch, float* depth_dev, size_t depth_pitch);
/* As topo_add_terrain with `heights` already in device memory (device-to-device copy). */
int topo_add_terrain_device(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg, const float* heights_dev, uint32_t w,
                            uint32_t h, const float raster_point[2], const float model_point[2],
                            const float pixel_scale[2]);

/* Re-runs the whole load phase (interior, seam and corner normals) over the resident heights, replaying the
 * original insertion order.  Used to time the load phase without PCIe. */
int topo_recompute_normals(topo_ctx* ctx);

/* n_views complete reference frames (one per `views[i]`) of width x height over the loaded tiles, in one
 * submission.  View i's image starts at rgba_dev + i*rgba_view_stride (bytes), rows rgba_pitch apart; same for
 * depth_dev (nullable).  Device pointers; asynchronous on the context's stream.  A 360-degree panorama is
 * 8 views of 45 degrees; with rgba_view_stride = 4*width and rgba_pitch = 4*n_views*width the
 * views land side by side in one row-major strip. */
int topo_render_views_device(topo_ctx* ctx, uint32_t n_views, const topo_uniforms* views, uint32_t width,
                             uint32_t height, uint8_t* rgba_dev, size_t rgba_view_stride, size_t rgba_pitch,
                             float* depth_dev, size_t depth_view_stride, size_t depth_pitch);

/* ---- multi-GPU: the 360-degree strip sharded by azimuth sector, and the viewpoint batch ---------------------------
 * (no reference counterpart: the reference renders one perspective view on one GPU.)
 * One process per GPU, each with its own topo_ctx over the same tiles (the DEM is replicated: 1.15 GB of 288 GB).  The
 * strip is TOPO_PANORAMA_SECTORS = 8 fixed sectors of sector_w x sector_h (the cameras of topo_panorama_uniforms), stored
 * SECTOR-MAJOR -- uint8 strip[8][sector_h][sector_w][4], float depth[8][sector_h][sector_w] -- so that the share of rank g,
 * sectors [8g/N, 8(g+1)/N), is one contiguous block.  topo_render_panorama renders this rank's sectors into their place in
 * strip_dev / depth_dev (device pointers, caller-owned, the same size on every rank) -- ONE cull / raster submission for all of
 * them -- and exchanges them over RCCL (xGMI) so that after topo_synchronize every rank holds the whole strip, bit-identical
 * for every N.  The exchange is overlapped: the frame is resolved in slots (topo_panorama_slots: a sector of the rank's range
 * and a band of its rows, ~8 MiB of RGBA each, the same plan on every rank), and behind each slot's resolve kernel the slot
 * is shipped on a second stream -- every rank sends its part to every other rank and receives theirs straight into place
 * (grouped ncclSend / ncclRecv: all seven xGMI links of a fully connected node carry a slot at once) -- while the next slot
 * is being resolved; the context's stream waits for the last exchange.  comm == NULL (or a world of 1): all 8 sectors, no
 * exchange.  RCCL is bound at run time (dlopen of librccl.so); nothing else in this library needs it.
 *
 * Call sequence for N ranks: rank 0 calls topo_comm_unique_id and hands the 128 bytes to the other ranks through the
 * host's own channel (a file, a socket, MPI, ...); every rank calls topo_comm_init(&comm, its_device, id, rank, N)
 * (collective: returns when all N have joined), loads the same tiles, then calls topo_render_panorama with the same
 * arguments; topo_comm_destroy at the end.  A host that already owns an ncclComm_t (e.g. one created by PyTorch) wraps it
 * with topo_comm_from_nccl instead; it is borrowed, not destroyed. */
#define TOPO_PANORAMA_SECTORS 8u
#define TOPO_COMM_ID_BYTES 128u
typedef struct topo_comm topo_comm;
int topo_comm_unique_id(uint8_t out_id[TOPO_COMM_ID_BYTES]);
int topo_comm_init(topo_comm** out, int hip_device, const uint8_t id[TOPO_COMM_ID_BYTES], int rank, int world);
int topo_comm_from_nccl(topo_comm** out, void* nccl_comm, int rank, int world);
void topo_comm_destroy(topo_comm* comm);
/* The sectors rank `rank` of `world` renders: [*first, *first + *count). */
void topo_panorama_sector_range(int rank, int world, uint32_t* first, uint32_t* count);
typedef struct topo_panorama_slot {
    uint32_t sector;       /* index within the rank's sector range: sector (first + sector) of the strip */
    uint32_t row0, rows;   /* the band of pixel rows [row0, row0 + rows) of that sector */
} topo_panorama_slot;
/* The resolve / exchange slots of one panorama for a world of `world` ranks, in the order every rank goes through them;
 * writes up to `cap` of them to `out` (nullable) and returns their number.  Slot i of rank g is bytes
 * [((g * 8 / world + sector) * sector_h + row0) * sector_w * 4, + rows * sector_w * 4) of the strip. */
uint32_t topo_panorama_slots(int world, uint32_t sector_w, uint32_t sector_h, topo_panorama_slot* out, uint32_t cap);
int topo_render_panorama(topo_ctx* ctx, topo_comm* comm, const float eye[3], float yaw0, float pitch, uint32_t sector_w,
                         uint32_t sector_h, float sun_theta_deg, float sun_phi_deg, int32_t view_mode, uint8_t* strip_dev,
                         float* depth_dev /* nullable */);
/* BASELINE config 5 (throughput mode): n_viewpoints independent panoramas -- viewpoint v has eye eyes_xyz[3v..], first
 * sector yaw yaw0[v], sun (sun_theta_phi_deg[2v], [2v+1]) -- into rgba_dev[v][8][sector_h][sector_w][4] (and depth_dev
 * likewise, nullable).  Eight viewpoints (64 views) go into one submission and, with topo_set_pipeline_depth(d), d
 * submissions are in flight; asynchronous (topo_join / topo_synchronize).  Viewpoints are independent: a multi-GPU host
 * gives each rank its own share of the batch, there is no collective. */
int topo_render_batch(topo_ctx* ctx, uint32_t n_viewpoints, const float* eyes_xyz, const float* yaw0,
                      const float* sun_theta_phi_deg, float pitch, uint32_t sector_w, uint32_t sector_h, int32_t view_mode,
                      uint8_t* rgba_dev, float* depth_dev /* nullable */);

/* RenderEngine::get_visible_labels(peaks, projection, size, depth_state, depth_buffer_view)   render_engine.rs:338-396
 * (the immediate consumer of the depth output, kept on the device so the pad_256 read-back
 * disappears).  For each peak (ECEF f32 xyz, PeakInstance.position): project_point3 through `camera_proj`; inside
 * the open NDC cube (|x| < 1, |y| < 1, z < 1) it maps to pixel (x_pos, y_pos) = ((0.5*(x+1)*w) as u32,
 * (-0.5*(y-1)*h) as u32) and is visible iff dist_from_depth(z) - 10 < dist_from_depth(depth[y_pos][x_pos]).
 * visible_out[i] = 0/1; xy_out[2i], xy_out[2i+1] = pixel (0 when not visible).  Host pointers; uses the uniforms of
 * the last topo_update and the depth of the last topo_render (which must have been called with depth_out). */
int topo_visible_peaks(topo_ctx* ctx, uint32_t n_peaks, const float* peaks_xyz, uint8_t* visible_out, uint32_t* xy_out);
/* Same over caller-owned device memory (any view of a topo_render_views_device submission); asynchronous. */
int topo_visible_peaks_device(topo_ctx* ctx, const topo_uniforms* view, uint32_t width, uint32_t height,
                              const float* depth_dev, size_t depth_pitch, uint32_t n_peaks, const float* peaks_xyz_dev,
                              uint8_t* visible_dev, uint32_t* xy_dev);

/* ---- overlay pass -------------------------------------------------------------------------
 * LineRenderer::render (line_renderer.rs:200-212, resources/shaders/line_shader.wgsl): the label leader lines and label
 * backgrounds, tessellated on the CPU (lyon, on the host's side) into a triangle list of `GpuVertex` (line_renderer.rs:18-25),
 * drawn INTO the post pass: opaque, counter-clockwise front faces, back faces culled, depth test Greater with write against
 * the post pass's depth attachment, which the post quad has filled with 1/4096 -- so z_index 2 (strokes) lies under 3
 * (fills) under 100 (the text renderer's layer, text_renderer.rs:291), an equal z keeps the earlier triangle, and z_index
 * <= 1 is never visible.  vs_main: z = z_index / 4096, p = (position + normal * line_width) * (1, -1),
 * clip = (2 p.x / width - 1, 2 p.y / height + 1, z, 1); positions are pixels, y down.  The image is the context's
 * width x height in its colour format (what topo_render / topo_render_device produced). */
typedef struct topo_overlay_vertex {
    float position[2];
    float normal[2];
    float color[3];      /* linear RGB; alpha is 1 */
    int32_t z_index;
} topo_overlay_vertex;   /* 32 bytes, = GpuVertex */
int topo_overlay_lines(topo_ctx* ctx, const topo_overlay_vertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                       uint32_t n_indices, float line_width /* Primitive.width: 0.5 in the reference */, uint8_t* rgba /* host, in/out */,
                       size_t rgba_pitch);
int topo_overlay_lines_device(topo_ctx* ctx, const topo_overlay_vertex* vertices, uint32_t n_vertices, const uint32_t* indices,
                              uint32_t n_indices, float line_width, uint8_t* rgba_dev /* device, in/out */, size_t rgba_pitch);

/* TextRenderer::render (text_renderer.rs:198-204, :259-291): the label text, drawn into the same pass after the lines by
 * glyphon 0.10.0 (third-party; Cargo.lock).  The host keeps what is the host's -- shaping and glyph rasterisation (cosmic-text,
 * swash) into an R8 mask atlas -- and hands over what glyphon's own vertex buffer holds: one `GlyphToRender` per glyph (28
 * bytes: quad pos .. pos + dim in pixels showing the atlas texels uv .. uv + dim 1 : 1, color = a << 24 | r << 16 | g << 8 | b,
 * content_type_with_srgb = {1 = mask atlas, 1 = decode r, g, b from sRGB (ColorMode::Accurate on an *Srgb surface)}).  The
 * fragment is (color.rgb, color.a * mask) under BlendState::ALPHA_BLENDING -- in linear light on an *Srgb surface -- with the
 * pass's depth state: Greater with write, every glyph at `depth` (the reference: 100/4096, above the lines), so where quads
 * overlap the one drawn first keeps its pixels.  Colour glyphs (content type 0) are rejected with TOPO_ERR_UNSUPPORTED. */
typedef struct topo_glyph {
    int32_t pos[2];
    uint16_t dim[2];
    uint16_t uv[2];
    uint32_t color;
    uint16_t content_type_with_srgb[2];
    float depth;         /* not read: `depth` of the call applies to every glyph */
} topo_glyph;            /* 28 bytes, = glyphon's GlyphToRender */
int topo_overlay_glyphs(topo_ctx* ctx, const topo_glyph* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas_mask /* host, R8 */,
                        uint32_t atlas_w, uint32_t atlas_h, uint8_t* rgba /* host, in/out */, size_t rgba_pitch);
int topo_overlay_glyphs_device(topo_ctx* ctx, const topo_glyph* glyphs /* host */, uint32_t n_glyphs, float depth, const uint8_t* atlas_mask /* host */,
                               uint32_t atlas_w, uint32_t atlas_h, uint8_t* rgba_dev /* device, in/out */, size_t rgba_pitch);

/* Run the context's work on an existing hipStream_t (e.g. PyTorch's current stream); NULL restores the
 * context's own stream. */
int topo_set_stream(topo_ctx* ctx, void* hip_stream);
int topo_synchronize(topo_ctx* ctx);
/* Throughput mode for topo_render_views_device: with depth d > 1 (max 4) consecutive submissions rotate through d frame
 * contexts, each with its own buffers and HIP stream, so that the memory-latency-bound cull/raster phases of one frame run
 * under the ALU-bound resolve pass of the previous one (c4: 1.27 -> 1.08 ms per panorama at depth 2).  Each submission is
 * ordered after the work queued so far on the context's stream (topo_set_stream), but NOT the other way round: outputs are
 * complete after topo_join (waits for the frames in flight) or topo_synchronize.  Depth 1 (default): everything runs
 * in order on the context's stream.  topo_render (host outputs) always waits for its frame. */
int topo_set_pipeline_depth(topo_ctx* ctx, int32_t depth);
/* Waits for every frame in flight.  A frame's status word is per frame; if one of the frames waited for overflowed its
 * rare-triangle queue (triangles were dropped: its outputs are incomplete) topo_join -- and likewise topo_synchronize --
 * returns TOPO_ERR_CAPACITY, once per such frame; the frames after it are unaffected.  (topo_render, the synchronous
 * entry point, never hands out such a frame: it grows the queue and renders the frame again.) */
int topo_join(topo_ctx* ctx);
/* Status bits of the frames that have completed since the previous call (waits for the frames in flight first; the bits
 * of all of them OR-ed together, so a burst of frames cannot hide an earlier frame's overflow behind a clean last frame; the
 * call clears them): out[0] = status bits (bit 0 big-triangle queue overflowed: handled exactly, slower; bit 1 rare-triangle queue overflowed:
 * frame incomplete; bit 2 bounds violation, only ever set by the TOPO_BOUNDS_CHECK build libtopo_hip_check.so),
 * out[1] = site tag and out[2], out[3] = low/high word of the offending value of the first bounds violation. */
int topo_frame_status(topo_ctx* ctx, uint32_t out[4]);

/* Form of the interior-normals kernel: 4, 8, 16, 32 or 64 = staged through an LDS tile of that many output rows per
 * workgroup (BASELINE config 3's sweep knob); 0 (the default: the sweep's optimum on MI355X) = without an LDS tile -- four
 * texels per lane, the rows above and below kept in registers, neighbours by wave shifts -- for tile widths that are a
 * multiple of four (other widths take the LDS form with 32 rows).  Same bytes either way. */
int topo_set_normals_lds_rows(topo_ctx* ctx, int rows);

/* Two-phase occlusion filter: blocks whose nearest possible view depth exceeds `metres` are rastered only if some
 * pixel of their conservative footprint is not already covered by nearer terrain (exact: a dropped block cannot
 * win a pixel).  Default 90 000 m; 0 turns the filter off (one phase, every frustum-surviving block rastered). */
int topo_set_occlusion_split(topo_ctx* ctx, float metres);

/* Per-kernel durations (ms, HIP events on the context's stream) of the last topo_render* call:
 * [0] clear  [1] cull  [2] raster (both phases)  [3] occlusion test  [4] raster_rare + raster_big (both phases)
 * [5] resolve+post  [6] total; of the last topo_recompute_normals: [7] the load phase -- every load-time kernel of the
 * resident tiles: the per-tile tables of the frame phase (block min/max, cull bounds, sin/cos tables) and the normals K1-K3 --
 * and [8] its tables part alone.  Synchronises. */
#define TOPO_TIMING_SLOTS 9
int topo_get_timings(topo_ctx* ctx, float out_ms[TOPO_TIMING_SLOTS]);
/* The same durations of the last `n_frames` frames (each context keeps the events of its last 32), oldest first:
 * out_ms[7 * i + k] = slot [k] (k = 0..6) of frame i, *n_out = frames written.  Waits for the frames in flight, so it is
 * called AFTER a timed region: nothing inside the region has to wait for a frame just to learn its kernels' durations. */
int topo_get_timing_history(topo_ctx* ctx, uint32_t n_frames, float* out_ms, uint32_t* n_out);
/* Which of the slots [0]..[5] to measure (bit i = slot i; default all).  Every timing event between two kernels leaves the
 * GPU idle for ~6 us while the marker completes -- 4 % of a c4 frame, a third of a c1 frame with all nine events -- so a
 * caller that only wants one kernel's duration (bench.py: the dominant one) or none selects just that; unselected slots
 * read 0.  The total [6] (an event in front of the frame's first kernel and one behind its last) is measured unless
 * TOPO_TIMING_NO_TOTAL is or'ed into the mask (then it reads 0 too; slot_mask = TOPO_TIMING_NO_TOTAL: no events at all). */
#define TOPO_TIMING_NO_TOTAL 0x80u
int topo_set_timing_slots(topo_ctx* ctx, uint32_t slot_mask);

/* Counters of the last topo_render* call: [0] near blocks rastered, [1] big-triangle items, [2] that frame's status bits
 * (bit 0: big-triangle queue overflowed -- handled in-lane, slower, still exact; bit 1: rare-triangle queue
 * overflowed -- triangles dropped, frame incomplete: see topo_join), [3] rare triangles
 * (>= 64 px across or near-clipped), [4] far blocks occlusion-tested, [5] far blocks that survived the test. */
int topo_get_counters(topo_ctx* ctx, uint32_t out[6]);

/* ---- host-side helpers mirroring the reference's CPU code ---------------------------------------------- */

/* Uniforms::new(&camera, bounds) with Camera{eye, yaw, pitch, fov_y, NEAR, FAR, view_mode, sun_angle{theta,phi}}:
 * render/data.rs:44-58, data/camera.rs:44-53,97-128 (glam 0.31 arithmetic restated in f32). */
void topo_camera_uniforms(const float eye[3], float yaw, float pitch, float fov_y, float width, float height,
                          float sun_theta_deg, float sun_phi_deg, int32_t view_mode, topo_uniforms* out);
/* The cameras of a 360-degree strip of n_sectors perspective sectors (no reference counterpart: the
 * reference renders one perspective view).  Sector k looks at yaw0 - k * 360/n degrees -- the reference's yaw grows
 * counter-clockwise seen from above (Camera::direction, camera.rs:101-109), so the strip reads left to right -- with
 * the vertical field of view that makes every sector 360/n degrees wide: 2 atan(tan(180/n deg) * sector_h / sector_w)
 * (topo_sector_fov_y, radians).  Writes n_sectors Uniforms blocks, ready for topo_render_views_device. */
float topo_sector_fov_y(uint32_t sector_w, uint32_t sector_h, uint32_t n_sectors);
void topo_panorama_uniforms(const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h,
                            float sun_theta_deg, float sun_phi_deg, int32_t view_mode, uint32_t n_sectors, topo_uniforms* out);
/* TerrainUniforms::new(coordinate_transform, (width, height)) -> 96 bytes (raster_point, model_point, pixel_scale,
 * size, normal_to_world_rot column-major mat4)                              render/data.rs:113-151 */
void topo_terrain_uniforms(const float raster_point[2], const float model_point[2], const float pixel_scale[2],
                           uint32_t w, uint32_t h, float out24[24]);
/* geometry::transform(h, longitude_deg, latitude_deg)                       render/geometry.rs:12-20 */
void topo_geometry_transform(float h, float longitude_deg, float latitude_deg, float out[3]);
/* dist_from_depth                                                           data/camera.rs:12-14 */
float topo_dist_from_depth(float depth);
/* pad_256                                                                   data/mod.rs:9-11 */
uint32_t topo_pad_256(uint32_t size);

/* CoordinateTransform::from_geo_tag_data(pixel_scale, tie_points, model_transformation)   common/coordinate_transform.rs:23-57
 * GeoTIFF ModelPixelScaleTag (3 doubles) + ModelTiepointTag (6 doubles) -> the six f32 the render path consumes.  A null
 * pointer stands for an absent tag (None).  Returns TOPO_ERR_UNSUPPORTED for the reference's IncorrectGeoTags (a
 * ModelTransformationTag is present, or one of the two required tags is absent) and TOPO_ERR_INVALID for
 * IncorrectGeoTagData (wrong value counts). */
int topo_coordinate_transform(const double* pixel_scale, uint32_t n_pixel_scale, const double* tie_points, uint32_t n_tie_points,
                              const double* model_transformation, float raster_point[2], float model_point[2],
                              float pixel_scale_out[2]);
/* CoordinateTransform::to_model / to_raster                                  common/coordinate_transform.rs:59-71 */
void topo_to_model(const float raster_point[2], const float model_point[2], const float pixel_scale[2], float x, float y,
                   float out[2]);
void topo_to_raster(const float raster_point[2], const float model_point[2], const float pixel_scale[2], float lon, float lat,
                    float out[2]);
/* get_height_value_at (F32 tiles)                                            common/coordinate_transform.rs:73-90
 * The terrain height under (longitude, latitude), as the reference looks it up to place the camera
 * (render_engine.rs:322-328): index = (raster.y as usize) * w + (raster.x as usize) with Rust's saturating float
 * casts, no bounds check other than the slice's.  Returns TOPO_ERR_NOT_FOUND where the reference yields None. */
int topo_height_value_at(const float* heights, uint32_t w, uint32_t h, const float raster_point[2], const float model_point[2],
                         const float pixel_scale[2], double longitude, double latitude, float* out);

/* ---- GeoTIFF decode: fetch_terrain's decode step, background_runner.rs:113-136 ------------------
 * The reference hands the downloaded bytes to the `tiff` crate (0.11.2): find_tag(ModelPixelScale / ModelTiepoint /
 * ModelTransformation) -> CoordinateTransform::from_geo_tag_data, read_image_to_buffer -> DecodingResult::F32, dimensions().
 * Supported here: classic TIFF of either byte order, first IFD, strips or tiles, one 32-bit IEEE float sample per pixel,
 * compression none / Deflate / LZW / PackBits, predictor none / horizontal / floating point.  The container is parsed
 * and the byte streams are decompressed on the host; predictor, byte order and tile placement run on the GPU.
 * Errors: TOPO_ERR_INVALID (malformed file, IncorrectGeoTagData), TOPO_ERR_UNSUPPORTED (BigTIFF, other sample layouts or
 * compressions, IncorrectGeoTags).  Unlike the reference, which ignores a failed read_image_to_buffer and carries on with
 * an empty raster, a decode error is reported. */
/* Size and CoordinateTransform of the first image; no GPU involved. */
int topo_geotiff_info(const uint8_t* bytes, size_t n_bytes, uint32_t* width, uint32_t* height, float raster_point[2],
                      float model_point[2], float pixel_scale[2]);
/* The raster as width*height host floats (row-major, row 0 first) -- what the reference keeps for get_height_value_at. */
int topo_geotiff_decode(topo_ctx* ctx, const uint8_t* bytes, size_t n_bytes, float* heights_out, size_t capacity_floats);
/* Decode + add_terrain in one step; the raster goes from the decoder to the tile without leaving the device. */
int topo_add_terrain_geotiff(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg, const uint8_t* bytes, size_t n_bytes);

/* UiController::get_locations_range(location, range_dist)                    control/ui_controller.rs:61-83
 * (the tile working set around a viewpoint; the reference calls it with 100 000 m).  Writes up
 * to `cap` (lat_deg, lon_deg) pairs in the reference's order -- sorted by (|lat - c_lat|, |lon - c_lon|), stable over
 * the row-major cartesian product, where c_lat is always 89 because of the reference's `.min(-90).max(89)` -- and
 * returns the number of tiles in the range (which may exceed cap).  In the reference the subsequent load order is
 * that of a HashSet (unspecified); callers here get the sorted order. */
uint32_t topo_locations_range(float latitude, float longitude, float range_dist, int32_t* out_lat_lon, uint32_t cap);

/* UiController::change_location(location, data, engine)                        control/ui_controller.rs:23-59
 * The working-set maintenance around get_locations_range: tiles of the new range that are not loaded are to be requested
 * (the reference sends DataRequested to its background runner), loaded tiles outside it are unloaded
 * (terrain.unload_terrain).  The reference walks HashSets, so its orders are unspecified; here `request` comes in
 * get_locations_range's order and `unload` in the order of `loaded`.  Counts may exceed the caps (nothing is written past
 * them).  topo_change_location_plan is the pure set arithmetic over a caller-held list of (lat, lon) pairs;
 * topo_change_location applies it to the context: it unloads what falls outside (as topo_unload_terrain) and reports
 * what the host still has to fetch and hand to topo_add_terrain / topo_add_terrain_geotiff. */
void topo_change_location_plan(float latitude, float longitude, float range_dist, const int32_t* loaded_lat_lon, uint32_t n_loaded,
                               int32_t* unload_out, uint32_t unload_cap, uint32_t* n_unload, int32_t* request_out,
                               uint32_t request_cap, uint32_t* n_request);
int topo_change_location(topo_ctx* ctx, float latitude, float longitude, float range_dist, int32_t* request_out,
                         uint32_t request_cap, uint32_t* n_request, uint32_t* n_unloaded);

#ifdef __cplusplus
}
#endif
#endif /* TOPO_HIP_H */
