// topo_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the terrain path.
//
//   load phase   k_tiff_rows (GeoTIFF predictor / layout), k_block_tables (block min/max, cull bounds, per-tile sin/cos tables),
//                k_normals_interior<ROWS>, k_normals_border (seams + corners) (compute_normals*.wgsl; once per add_terrain)
//   frame phase  k_clear -> k_cull -> [near] k_raster -> k_raster_rare -> k_raster_big -> k_occlusion ->
//                [far survivors] k_raster -> k_raster_rare -> k_raster_big -> k_resolve
//                (render_shader.wgsl vs_main + fixed-function raster/depth, fs_main, postprocessing_shader.wgsl)
//
// The frame is a visibility-buffer renderer: every surviving fragment does a 64-bit atomic min of
// (depth bits << 32 | draw-order id) -- the minimum reproduces CompareFunction::Less *and* the API-order
// tie-break of the reference's in-order draws -- and one resolve pass shades the winner of each pixel and
// applies the contour post pass.  Integer/float work without a contraction: no MFMA.
//
// Compiled with -ffp-contract=off: results must match the arithmetic spec bit for bit.
#include "topo_kernels.h"

#include <atomic>
#include <type_traits>

namespace topo {

namespace {

// A fragment meets the visibility buffer through one 64-bit atomic min, issued blind: the atomic returns nothing,
// so the wave never waits for it, whereas reading the current key first (to skip fragments that cannot win) puts a
// full memory round trip into every loop that emits fragments.  Measured on c4: k_raster 0.187 -> 0.158 ms,
// k_raster_big 0.45 -> 0.37 ms without the pre-test (profiles/README.md).
// The buffer is tracked in segments of 64 consecutive keys: whoever writes a key marks its segment (a plain byte
// store of 1: racing writers agree), k_clear re-initialises only marked segments, and k_resolve does not even read
// the keys of a block whose segments are all unmarked.  About half of a panorama is sky that no fragment touches.
struct Vis {
    uint64_t* p;             // this view's keys
    const uint64_t* base;    // the whole buffer (segment numbers are global)
    uint8_t* dirty;
#ifdef TOPO_BOUNDS_CHECK
    uint32_t* counters;
    size_t view_keys;        // W * H
#endif
};

// TOPO_BOUNDS_CHECK build (libtopo_hip_check.so, `make check`): every index this file forms into the visibility buffer,
// the segment marks, the queues, the tile rasters and the outputs is tested first; a violation sets kStatusBounds,
// records (site tag, offending value) of the first one in counters[8..10] and the access is skipped instead of made.
// It is the address sanitizer this pool does not offer for the GPU (tests/test_gpu_parity.py runs the suite's scenes
// through it once).  In the product build TOPO_CHK is `true` and costs nothing.
#ifdef TOPO_BOUNDS_CHECK
__device__ __noinline__ void bounds_violation(uint32_t* counters, uint32_t tag, uint64_t value) {
    if ((atomicOr(&counters[2], kStatusBounds) & kStatusBounds) == 0) {
        counters[8] = tag;
        counters[9] = (uint32_t)value;
        counters[10] = (uint32_t)(value >> 32);
    }
}
#define TOPO_CHK(counters, ok, tag, value) ((ok) ? true : (bounds_violation((counters), (tag), (uint64_t)(value)), false))
#else
#define TOPO_CHK(counters, ok, tag, value) true
#endif

__device__ __forceinline__ Vis view_vis(const FrameParams& P, uint32_t view) {
#ifdef TOPO_BOUNDS_CHECK
    (void)TOPO_CHK(P.counters, view < P.n_views, 1u, view);
    return Vis{P.vis + (size_t)view * P.W * P.H, P.vis, P.dirty, P.counters, (size_t)P.W * P.H};
#else
    return Vis{P.vis + (size_t)view * P.W * P.H, P.vis, P.dirty};
#endif
}
// the atomic alone, for callers that mark the segments themselves (k_raster_big: once per item and pixel row)
__device__ __forceinline__ void vis_min_unmarked(const Vis& v, size_t pix, uint64_t key) {
#ifdef TOPO_BOUNDS_CHECK
    if (!TOPO_CHK(v.counters, pix < v.view_keys, 2u, pix)) return;
#endif
    atomicMin(reinterpret_cast<unsigned long long*>(v.p + pix), (unsigned long long)key);
}
__device__ __forceinline__ void vis_min(const Vis& v, size_t pix, uint64_t key) {
#ifdef TOPO_BOUNDS_CHECK
    if (!TOPO_CHK(v.counters, pix < v.view_keys, 2u, pix)) return;
#endif
    uint64_t* q = v.p + pix;
    atomicMin(reinterpret_cast<unsigned long long*>(q), (unsigned long long)key);
    v.dirty[(size_t)(q - v.base) >> 6] = 1;
}

// ======================================================================================================
// load phase
// ======================================================================================================

struct SinCos64 { double s, c; };
__device__ __forceinline__ SinCos64 sincos64(double a) { SinCos64 r; r.s = sin(a); r.c = cos(a); return r; }

// The view-independent half of the cull for one raster block, in f64: the bounding sphere of the block's patch, the unit
// directions of its four corners and the sagitta of the patch over their flat hull.  lo / la: sin/cos of the block's first and
// last longitude / latitude, loc / lac: of its centre.
__device__ __forceinline__ void block_bounds_store(double* bounds, uint32_t blocks_per_tile, uint32_t blk, float bmn, float bmx, const SinCos64 lo[2],
                                                   const SinCos64 la[2], const SinCos64& loc, const SinCos64& lac) {
    const double hmin = (double)bmn, hmax = (double)bmx, hmid = 0.5 * (hmin + hmax);
    double* bs = bounds + (size_t)blk * 4;                                              // sphere
    double* bb = bounds + (size_t)blocks_per_tile * 4 + (size_t)blk * 12;               // corner directions
    double u[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const SinCos64 &o = lo[k & 1], &a = la[k >> 1];
        u[k][0] = a.c * o.c; u[k][1] = a.c * o.s; u[k][2] = a.s;
        bb[3 * k] = u[k][0]; bb[3 * k + 1] = u[k][1]; bb[3 * k + 2] = u[k][2];
    }
    const double Rm = (double)kR0 + hmid;
    const double c[3] = {Rm * lac.c * loc.c, Rm * lac.c * loc.s, Rm * lac.s};
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double dx = Rm * u[k][0] - c[0], dy = Rm * u[k][1] - c[1], dz = Rm * u[k][2] - c[2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        r2 = d2 > r2 ? d2 : r2;
    }
    // every direction of the patch lies within the angular distance of the farthest corner from the centre direction,
    // so the corners' chord distance bounds the sphere; + half the height range + margin
    bs[0] = c[0]; bs[1] = c[1]; bs[2] = c[2];
    bs[3] = sqrt(r2) + 0.5 * (hmax - hmin) + 8.0 + 64.0;
    // How far the curved patch can stick out of the flat-faced hull of its eight slab corners (radially over the top
    // face, sideways over the face along its equator-side parallel): at most the sagitta of the farthest corner's
    // arc, R (1 - cos theta_max).  0.3 .. 0.7 m for a 60 x 15 cell block of a 1200-px tile, hundreds of metres for the
    // blocks of a coarse tile: the occlusion filter pads its slab by this and only takes blocks where it is <= 1 m.
    double dmin = 1.0;
    const double uc[3] = {lac.c * loc.c, lac.c * loc.s, lac.s};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double d = u[k][0] * uc[0] + u[k][1] * uc[1] + u[k][2] * uc[2];
        dmin = d < dmin ? d : dmin;
    }
    bounds[(size_t)blocks_per_tile * 16 + blk] = ((double)kR0 + (hmax > 0.0 ? hmax : 0.0) + 2.0) * (1.0 - dmin);
}
// The angles whose f64 sin/cos the bounds take: the latitude of vertex row vy / the longitude of vertex column vx (halves allowed).
__device__ __forceinline__ double block_lat64(const TileDev& t, double vy) { return ((vy - (double)t.raster_y) * -(double)t.scale_y + (double)t.model_y) * 0.017453292519943295; }
__device__ __forceinline__ double block_lon64(const TileDev& t, double vx) { return ((vx - (double)t.raster_x) * (double)t.scale_x + (double)t.model_x) * 0.017453292519943295; }
// entries [start, start + stride, ...) of a tile's sin/cos tables (TileDev::trig_lon, trig_lat)
__device__ __forceinline__ void trig_tables_fill(const TileDev& t, uint32_t w, uint32_t h, uint32_t start, uint32_t stride) {
    for (uint32_t e = start; e < w + h; e += stride) {
        float sn, cs;
        if (e < w) {
            sincos_f(vertex_lon(t, e), sn, cs);
            const_cast<float*>(t.trig_lon)[2 * e] = sn;
            const_cast<float*>(t.trig_lon)[2 * e + 1] = cs;
        } else {
            sincos_f(vertex_lat(t, e - w), sn, cs);
            const_cast<float*>(t.trig_lat)[2 * (e - w)] = sn;
            const_cast<float*>(t.trig_lat)[2 * (e - w) + 1] = cs;
        }
    }
}

// Per-tile tables of the frame phase, for a batch of tiles (blockIdx.y) in ONE launch: min/max height of the (kVX x kVY)
// vertices of every raster block, the view-independent half of the cull (f64: the block's bounding sphere, the unit directions
// of its four corners, the sagitta of its patch), and the tile's sin/cos tables (TileDev::trig_lon / trig_lat).
// One WAVE per run of four horizontally adjacent raster blocks (241 vertex columns x 16 vertex rows): lane i keeps the column
// minima / maxima of columns i, i + 64, i + 128, i + 192 while the rows stream by as coalesced 256-byte reads (the DEM is
// read once, at HBM speed; round 2 launched one 64-thread workgroup per block and tile after tile: 19 us per tile, 0.3 TB/s),
// the 61-column ranges of the four blocks are reduced through a wave-private LDS strip, the fifteen f64 sin/cos pairs the four
// blocks need (three latitudes, twelve longitudes) are evaluated by fifteen lanes at once instead of six per block one after
// the other on lane 0, and lanes 0..3 finish one block each.  Same expressions, same results as the one-block-per-wave form.
constexpr uint32_t kTblBlocks = 4;                                  // raster blocks per wave
constexpr uint32_t kTblCols = kTblBlocks * kBCX + 1;                // 241 vertex columns
static_assert(kTblCols <= 256, "four column slots per lane");
__global__ __launch_bounds__(256) void k_block_tables(const TileDev* __restrict__ tiles, uint32_t first, uint32_t w, uint32_t h, uint32_t bx_count,
                                                      uint32_t by_count) {
    __shared__ float s_mn[4][256], s_mx[4][256];
    __shared__ double s_sc[4][15][2];
    const TileDev& t = tiles[first + blockIdx.y];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t runs_per_row = (bx_count + kTblBlocks - 1) / kTblBlocks, n_runs = runs_per_row * by_count;
    const uint32_t blocks_per_tile = bx_count * by_count;
    const auto heights = TOPO_GLOBAL_F32(t.heights);
    float* const minmax = const_cast<float*>(t.block_minmax);
    double* const bounds = const_cast<double*>(t.block_bounds);
    for (uint32_t run = blockIdx.x * 4 + wave; run < n_runs; run += gridDim.x * 4) {
        const uint32_t by = run / runs_per_row, bx0 = (run - by * runs_per_row) * kTblBlocks;
        const uint32_t nb = min(kTblBlocks, bx_count - bx0);       // blocks of this run
        const uint32_t c0 = bx0 * kBCX, y0 = by * kBCY;
        const uint32_t ncols = min(nb * kBCX + 1, w - c0), nrows = min(kVY, h - y0);
        // ---- column minima / maxima
        float mn[4], mx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
        if ((w & 3u) == 0u) {
            // rows are 16-byte aligned and c0 = 240 (run) is a multiple of four: lane i reads columns 4 i .. 4 i + 3 in ONE load
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const uint32_t cl = 4 * lane < ncols ? 4 * lane : (ncols - 1) & ~3u;      // (surplus lanes re-read the last vector: it exists, w % 4 == 0)
            for (uint32_t r = 0; r < nrows; ++r) {
                const f32x4_t v = *(const __attribute__((address_space(1))) f32x4_t*)(heights + (size_t)(y0 + r) * w + c0 + cl);
                mn[0] = fminf(mn[0], v.x); mx[0] = fmaxf(mx[0], v.x);
                mn[1] = fminf(mn[1], v.y); mx[1] = fmaxf(mx[1], v.y);
                mn[2] = fminf(mn[2], v.z); mx[2] = fmaxf(mx[2], v.z);
                mn[3] = fminf(mn[3], v.w); mx[3] = fmaxf(mx[3], v.w);
            }
            if (lane < 64) {      // (columns beyond ncols hold copies of real columns of this run or, in its last vector, of the tile's last columns: never read below)
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) { s_mn[wave][(4 * lane + k) & 255u] = mn[k]; s_mx[wave][(4 * lane + k) & 255u] = mx[k]; }
            }
        } else {
            for (uint32_t r = 0; r < nrows; ++r) {
                const auto row = heights + (size_t)(y0 + r) * w + c0;
                float v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t c = lane + 64 * k;
                    v[k] = row[c < ncols ? c : ncols - 1];      // (unconditional loads; the surplus lanes re-read the last column)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { mn[k] = fminf(mn[k], v[k]); mx[k] = fmaxf(mx[k], v[k]); }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) { s_mn[wave][lane + 64 * k] = mn[k]; s_mx[wave][lane + 64 * k] = mx[k]; }
        }
        // ---- the f64 sin/cos pairs: lanes 0..2 latitudes (y0, y1, centre), lanes 3 + 3 b .. 5 + 3 b longitudes (x0, x1, centre) of block b
        const double yy0 = (double)(by * kBCY);
        double yy1 = yy0 + (double)kBCY;
        if (yy1 > (double)(h - 1)) yy1 = (double)(h - 1);
        if (lane < 3u + 3u * nb) {
            double a;
            if (lane < 3u) {
                const double vy = lane == 0 ? yy0 : (lane == 1 ? yy1 : 0.5 * (yy0 + yy1));
                a = block_lat64(t, vy);
            } else {
                const uint32_t b = (lane - 3u) / 3u, which = (lane - 3u) - 3u * b;
                const double xx0 = (double)((bx0 + b) * kBCX);
                double xx1 = xx0 + (double)kBCX;
                if (xx1 > (double)(w - 1)) xx1 = (double)(w - 1);
                const double vx = which == 0 ? xx0 : (which == 1 ? xx1 : 0.5 * (xx0 + xx1));
                a = block_lon64(t, vx);
            }
            const SinCos64 sc = sincos64(a);
            s_sc[wave][lane][0] = sc.s;
            s_sc[wave][lane][1] = sc.c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (wave-private LDS: orders the compiler, emits nothing)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- the blocks' own 61-column ranges
        float bmn = INFINITY, bmx = -INFINITY;     // lane b ends up with block b's
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t c = b * kBCX + lane;
            float lo = lane < kVX && c < ncols ? s_mn[wave][c] : INFINITY, hi = lane < kVX && c < ncols ? s_mx[wave][c] : -INFINITY;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                lo = fminf(lo, __shfl_xor(lo, off));
                hi = fmaxf(hi, __shfl_xor(hi, off));
            }
            if (lane == b) { bmn = lo; bmx = hi; }
        }
        if (lane < nb) {
            const uint32_t blk = by * bx_count + bx0 + lane;
            minmax[2 * blk] = bmn;
            minmax[2 * blk + 1] = bmx;
            const double(*sc)[2] = s_sc[wave];
            const SinCos64 lo[2] = {{sc[3 + 3 * lane][0], sc[3 + 3 * lane][1]}, {sc[4 + 3 * lane][0], sc[4 + 3 * lane][1]}};
            const SinCos64 la[2] = {{sc[0][0], sc[0][1]}, {sc[1][0], sc[1][1]}};
            const SinCos64 loc = {sc[5 + 3 * lane][0], sc[5 + 3 * lane][1]}, lac = {sc[2][0], sc[2][1]};
            block_bounds_store(bounds, blocks_per_tile, blk, bmn, bmx, lo, la, loc, lac);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the next run rewrites the strips)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    trig_tables_fill(t, w, h, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// The two halves of k_block_tables that do not read the DEM, for the load path whose normals pass collects the block minima /
// maxima itself (k_normals_rolling<.., true>): the sin/cos tables BEFORE that pass (it reads cos(latitude) from them), the f64
// bounds AFTER it (one lane per raster block, from the block's min/max).  Same expressions as k_block_tables, same results.
__global__ __launch_bounds__(256) void k_trig_tables(const TileDev* __restrict__ tiles, uint32_t first, uint32_t w, uint32_t h) {
    trig_tables_fill(tiles[first + blockIdx.y], w, h, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
// kLanes = 8: eight lanes per raster block -- lanes 0..5 of a group evaluate one f64 sin/cos pair each (first / last / centre
// latitude, first / last / centre longitude), lane 0 collects them and finishes the block: a block's six sin/cos calls one after the
// other on one lane are the whole latency of this kernel when a single tile is added (add_terrain: 0.43 -> 0.36 ms per tile).
// kLanes = 1: one lane per block, for a batch of tiles, where the lanes are what counts (100 tiles: 0.239 against 0.251 ms for the
// whole load phase).
__device__ __forceinline__ double shfl_f64(double v, int src) {
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}
template <int kLanes>
__global__ __launch_bounds__(256) void k_block_bounds(const TileDev* __restrict__ tiles, uint32_t first, uint32_t w, uint32_t h, uint32_t bx_count,
                                                      uint32_t by_count) {
    static_assert(kLanes == 1 || kLanes == 8, "");
    const TileDev& t = tiles[first + blockIdx.y];
    const uint32_t sub = kLanes == 8 ? threadIdx.x & 7u : 0u, blocks_per_tile = bx_count * by_count;
    const uint32_t blk_raw = kLanes == 8 ? blockIdx.x * 32 + (threadIdx.x >> 3) : blockIdx.x * 256 + threadIdx.x;
    const uint32_t blk = blk_raw < blocks_per_tile ? blk_raw : blocks_per_tile - 1;      // (surplus lanes redo the last block and store nothing)
    const uint32_t by = blk / bx_count, bx = blk - by * bx_count;
    const double yy0 = (double)(by * kBCY), xx0 = (double)(bx * kBCX);
    double yy1 = yy0 + (double)kBCY, xx1 = xx0 + (double)kBCX;
    if (yy1 > (double)(h - 1)) yy1 = (double)(h - 1);
    if (xx1 > (double)(w - 1)) xx1 = (double)(w - 1);
    SinCos64 g[6];      // latitudes of yy0, yy1, the centre; longitudes of xx0, xx1, the centre
    if (kLanes == 8) {
        // sub 0 1 2: the latitudes; sub 3 4 5: the longitudes (6, 7: idle copies of 5)
        const double vy = sub == 0 ? yy0 : (sub == 1 ? yy1 : 0.5 * (yy0 + yy1)), vx = sub == 3 ? xx0 : (sub == 4 ? xx1 : 0.5 * (xx0 + xx1));
        const SinCos64 mine = sincos64(sub < 3 ? block_lat64(t, vy) : block_lon64(t, vx));
        const int base = (int)((threadIdx.x & 63u) & ~7u);
#pragma unroll
        for (int k = 0; k < 6; ++k) { g[k].s = shfl_f64(mine.s, base + k); g[k].c = shfl_f64(mine.c, base + k); }
    } else {
        g[0] = sincos64(block_lat64(t, yy0)); g[1] = sincos64(block_lat64(t, yy1)); g[2] = sincos64(block_lat64(t, 0.5 * (yy0 + yy1)));
        g[3] = sincos64(block_lon64(t, xx0)); g[4] = sincos64(block_lon64(t, xx1)); g[5] = sincos64(block_lon64(t, 0.5 * (xx0 + xx1)));
    }
    if (sub == 0 && blk_raw < blocks_per_tile) {
        const SinCos64 la[2] = {g[0], g[1]}, lo[2] = {g[3], g[4]};
        block_bounds_store(const_cast<double*>(t.block_bounds), blocks_per_tile, blk, t.block_minmax[2 * blk], t.block_minmax[2 * blk + 1], lo, la, g[5], g[2]);
    }
}

// Workgroups are dealt round-robin over the chip's eight XCDs (each with an L2 of its own), so the workgroups that share an L2
// are L, L + 8, L + 16, ... of the launch order -- and neighbouring pieces of a tile, which re-read each other's halo rows and
// columns, never meet in one.  (Measured on the c4 load phase: FETCH_SIZE 1.54x the DEM for the LDS form, 1.25x for the
// LDS-less one -- exactly their halo ratios: every halo line came over the fabric again.)  This hands each XCD a CONTIGUOUS
// eighth of a launch's pieces instead: piece = (L % 8) * ceil(n / 8) + L / 8.  A speed matter only (nothing depends on
// which XCD runs what); returns false for the slack pieces at the end.
__device__ __forceinline__ bool xcd_contiguous_piece(uint32_t n_pieces, uint32_t& piece) {
    const uint32_t L = blockIdx.x, per_xcd = (n_pieces + 7u) / 8u;
    piece = (L & 7u) * per_xcd + (L >> 3);
    return (L >> 3) < per_xcd && piece < n_pieces;
}

// Interior normals (compute_normals_shader.wgsl:22-51) of a batch of tiles (blockIdx.z).  128 x ROWS output texels per
// 256-thread workgroup, TWO horizontally adjacent texels per lane: the kernel issues as many instructions as it moves
// bytes (one texel per lane: ~90 instructions per 64 texels, 0.21 ms of issue slots beside 0.20 ms of HBM time at c4), and
// everything that is not the stencil's own arithmetic -- addresses, edge tests, LDS traffic, loop control, the staging
// loads -- is paid per lane, not per texel.  The (ROWS+2) x 130 height tile is staged in LDS row by row -- wave w takes
// rows w, w + 4, ...: one coalesced 512-byte read per row (a pair of columns per lane) plus a two-lane read for the halo
// columns -- each texel's four taps then come from LDS; cos(latitude) is evaluated once per row.  The border ring, which
// the shader leaves untouched (:30-33) and which is zero in a freshly created texture, is written as zero here so no
// separate clear is needed; seam/corner passes run afterwards.  ROWS is the LDS tile-size knob
// (topo_set_normals_lds_rows).
// Arithmetic: normal_texel_fast() -- a reciprocal-square-root estimate and a guard band around the 8-bit code boundaries
// -- settles 998 texels in 1000; a wave in which some lane's texel falls inside the guard band (or is not finite)
// evaluates the full chain (correctly rounded sqrt, three IEEE divisions) for those lanes.  Same bytes either way.
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));         // a pair of floats at any 4-byte boundary
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int ROWS>
__global__ __launch_bounds__(256) void k_normals_interior(const TileDev* __restrict__ tiles, uint32_t first, uint32_t n_tiles, int W, int H) {
    // column c of the tile (c = -1 .. 128) lives at index c + 2: a lane's pair (2 tx, 2 tx + 1) at the even index 2 tx + 2
    __shared__ __attribute__((aligned(16))) float tile[ROWS + 2][132];
    __shared__ float s_ys[ROWS];
    // pieces = (tile, row band, column block), column block fastest; handed out XCD by XCD (xcd_contiguous_piece)
    const uint32_t gx = ((uint32_t)W + 127u) / 128u, gy = ((uint32_t)H + ROWS - 1u) / ROWS;
    uint32_t piece;
    if (!xcd_contiguous_piece(gx * gy * n_tiles, piece)) return;      // (workgroup-uniform: before any barrier)
    const uint32_t bz = piece / (gx * gy), by_ = (piece - bz * gx * gy) / gx, bx_ = piece - bz * gx * gy - by_ * gx;
    const TileDev& t = tiles[first + bz];
    const auto heights = TOPO_GLOBAL_F32(t.heights);          // global, not flat, memory operations
    const auto normals = TOPO_GLOBAL_U32_RW(t.normals);
    const int x0 = (int)bx_ * 128, y0 = (int)by_ * ROWS;
    const int tx = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int gx0 = x0 + 2 * tx;                              // the lane's first column (the second: gx0 + 1)
    {
        // Every load is unconditional (clamped address, value discarded where it does not apply) and all of a wave's loads
        // are issued before the first LDS write: a branch around a load makes the compiler wait for it before going on, one
        // trip to memory per row.  The pair is read from columns (px, px + 1) with px clamped to W - 2, so that both exist.
        constexpr int kIter = (ROWS + 2 + 3) / 4;
        const int px = gx0 > W - 2 ? W - 2 : gx0;
        const int hx = x0 - 1 + 129 * (tx & 1);               // lanes 0 / 1: columns -1 / 128
        const int chx = hx < 0 ? 0 : (hx > W - 1 ? W - 1 : hx);
        f32x2_a4 a[kIter];
        float b[kIter];
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int gy = y0 + wy + 4 * k - 1;
            const int cy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
            a[k] = *(const __attribute__((address_space(1))) f32x2_a4*)(heights + ((size_t)cy * W + px));
            b[k] = heights[(size_t)cy * W + chx];
        }
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const int ly = wy + 4 * k, gy = y0 + ly - 1;
            const bool row_in = gy >= 0 && gy < H;
            if (ly < ROWS + 2) {
                // (gx0 == W - 1: the tile's last column is the second element of the clamped pair)
                const float v0 = !row_in || gx0 > W - 1 ? 0.0f : (gx0 == W - 1 ? a[k].y : a[k].x);
                const float v1 = row_in && gx0 + 1 <= W - 1 ? a[k].y : 0.0f;
                *reinterpret_cast<float2*>(&tile[ly][2 * tx + 2]) = make_float2(v0, v1);
                if (tx < 2) tile[ly][1 + 129 * tx] = row_in && hx >= 0 && hx < W ? b[k] : 0.0f;
            }
        }
    }
    if (threadIdx.x < ROWS) {
        const float latitude = ((float)(y0 + (int)threadIdx.x) - t.raster_y) * -t.scale_y + t.model_y;
        s_ys[threadIdx.x] = deg2rad(t.scale_y) * kR0 * cos_f(deg2rad(latitude));
    }
    __syncthreads();
    const float xs = deg2rad(t.scale_x) * kR0;
    const bool col_in0 = gx0 >= 1 && gx0 < W - 1, col_in1 = gx0 + 1 < W - 1;      // (gx0 + 1 >= 1 always)
    auto out = normals + ((size_t)(y0 + wy) * W + (gx0 < W ? gx0 : 0));
    const size_t out_step = (size_t)4 * W;
#pragma unroll
    for (int r = wy; r < ROWS; r += 4, out += out_step) {
        const int gy = y0 + r;
        if (gy >= H) break;      // (wave-uniform)
        const bool row_in = gy >= 1 && gy < H - 1;
        const float2 top = *reinterpret_cast<const float2*>(&tile[r][2 * tx + 2]), bot = *reinterpret_cast<const float2*>(&tile[r + 2][2 * tx + 2]);
        const float2 mid = *reinterpret_cast<const float2*>(&tile[r + 1][2 * tx + 2]);      // the pair's own heights: each is the other's neighbour
        const float hl = tile[r + 1][2 * tx + 1], hr = tile[r + 1][2 * tx + 4], ys = s_ys[r];
        uint32_t t0 = 0, t1 = 0;
        const bool in0 = col_in0 && row_in, in1 = col_in1 && row_in;
        const bool settled0 = normal_texel_fast(xs, ys, top.x, hl, mid.y, bot.x, t0) || !in0;
        const bool settled1 = normal_texel_fast(xs, ys, top.y, mid.x, hr, bot.y, t1) || !in1;
        if (!settled0) t0 = normal_texel(xs, ys, top.x, hl, mid.y, bot.x);      // the guard band and non-finite heights: the full chain
        if (!settled1) t1 = normal_texel(xs, ys, top.y, mid.x, hr, bot.y);
        t0 = in0 ? t0 : 0u;
        t1 = in1 ? t1 : 0u;
        if (gx0 + 1 < W) {
            u32x2_a4 v;
            v.x = t0; v.y = t1;
            // (non-temporal: the texture is written once here and read much later -- 0.246 -> 0.235 ms at c4)
            __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u32x2_a4*)(out));
        } else if (gx0 < W) {
            *out = t0;
        }
    }
}

// The same pass WITHOUT an LDS tile (topo_set_normals_lds_rows(0); needs a tile width that is a multiple of four): a wave owns
// a strip of 256 columns -- FOUR adjacent texels per lane, one 16-byte load and one 16-byte store per lane and row -- and
// walks kRollRows rows of it top to bottom with the rows above and below the current one kept in registers (each height is
// loaded once per strip and chunk; the chunk's first and last rows twice), the next four rows always in flight.  The texel
// left of a lane's first and right of its last come from the neighbouring lanes by DPP wave shifts; the two columns beside
// the strip by one extra two-address load per row.  No barrier, no LDS traffic, 1 KiB per wave and memory instruction.
// cos(latitude) of a row is the tile's trig_lat table entry (k_block_tables: the same function of the same input).
__device__ __forceinline__ float unif2(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }
__device__ __forceinline__ float unif_first(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wave_from_left(float v, float first) {      // lane i: lane i - 1's v; lane 0: `first`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_from_right(float v, float last) {      // lane i: lane i + 1's v; lane 63: `last`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(last), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
}
// kTables: the pass also collects the min / max height of every raster block (TileDev::block_minmax) -- the one thing
// k_block_tables reads the DEM for -- so that the load phase reads the DEM ONCE.  A strip is then 240 columns (four raster
// blocks of kBCX = 60 cells; lanes 60..63 only feed lane 59's right neighbour) and a workgroup's waves share one block row
// (kBCY = 15 rows: 4 + 4 + 4 + 3): a lane folds its four columns and its right neighbour's first one into one running minimum
// and maximum per row (lanes 15 b .. 15 b + 14 then hold exactly the 61 vertex columns of block b), a wave adds the row below
// its last one (the block's 16th vertex row for the last wave, a row of the same block for the others), the fifteen lanes of a
// block are folded by four shuffles (1, 2, 4, 7: the windows overlap, which a minimum does not mind), the waves' partial
// results meet in LDS.  Needs W % 240 == 0 (COP90: 1200, COP30: 3600); k_trig_tables runs before, k_block_bounds after.
template <int kRollRows, int kWaves, bool kTables, int kBatch = 4>      // kBatch: rows loaded per round
__global__ __launch_bounds__(64 * kWaves) void k_normals_rolling(const TileDev* __restrict__ tiles, uint32_t first, uint32_t n_tiles, int W, int H,
                                                                 uint32_t bx_count, uint32_t by_count) {
    constexpr int kCols = kTables ? 4 * (int)kBCX : 256;                      // columns of a strip
    constexpr int kChunkRows = kTables ? (int)kBCY : kRollRows * kWaves;      // rows of a workgroup
    static_assert(!kTables || (kRollRows * kWaves >= (int)kBCY && kRollRows * (kWaves - 1) < (int)kBCY), "the waves of a workgroup cover one block row");
    __shared__ float s_part[kTables ? kWaves : 1][4][2];
    const uint32_t gx = ((uint32_t)W + kCols - 1u) / kCols, gy = ((uint32_t)H + kChunkRows - 1u) / kChunkRows;
    uint32_t piece;
    if (!xcd_contiguous_piece(gx * gy * n_tiles, piece)) return;      // (workgroup-uniform: before any barrier)
    const uint32_t bz = piece / (gx * gy), by_ = (piece - bz * gx * gy) / gx, bx_ = piece - bz * gx * gy - by_ * gx;
    const TileDev& t = tiles[first + bz];
    const auto heights = TOPO_GLOBAL_F32(t.heights);
    const auto normals = TOPO_GLOBAL_U32_RW(t.normals);
    // (the table was written by an earlier launch and a row's entry is wave-uniform: read through the constant address space it
    // comes over the scalar data path.  As a vector load it was the youngest memory operation of its row, and waiting for it --
    // s_waitcnt vmcnt(0) -- waited for every row in flight and for the previous row's store as well.)
    const auto trig_lat = (const __attribute__((address_space(4))) float*)(const void*)t.trig_lat;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = (int)bx_ * kCols, c0 = x0 + 4 * lane;
    const int y0 = (int)by_ * kChunkRows + wave * kRollRows;
    const int y_end = ((int)by_ + 1) * kChunkRows < H ? ((int)by_ + 1) * kChunkRows : H;
    if (!kTables && y0 >= H) return;
    const int y1 = y0 + kRollRows < y_end ? y0 + kRollRows : y_end;      // rows [y0, y1)   (kTables: possibly none)
    const bool col_active = c0 < W && c0 < x0 + kCols;            // (W % 4 == 0: a lane's four columns are all inside or all outside)
    const int cc = c0 < W ? c0 : W - 4;
    // the two columns beside the strip, one load for both: lanes 0..31 the left one, lanes 32..63 the right one (clamped)
    const int ce = lane < 32 ? (x0 > 0 ? x0 - 1 : 0) : (x0 + 256 < W ? x0 + 256 : W - 1);
    auto row_ptr = [&](int y) { return heights + (size_t)(y < 0 ? 0 : (y > H - 1 ? H - 1 : y)) * W; };
    auto load4 = [&](int y) { return *(const __attribute__((address_space(1))) f32x4_t*)(row_ptr(y) + cc); };
    auto load_edge = [&](int y) { return row_ptr(y)[ce]; };
    const float xs = deg2rad(t.scale_x) * kR0, ys0 = deg2rad(t.scale_y) * kR0;
    float mn = INFINITY, mx = -INFINITY;      // kTables: the lane's columns 4 lane .. 4 lane + 4 over the wave's rows
    if (!kTables || y0 < y1) {
        // rows y - 1 and y of the first output row, then four new rows per round
        f32x4_t above = load4(y0 - 1), mid = load4(y0);
        float mid_edge = load_edge(y0);
        f32x4_t nx[kBatch];
        float ne[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) { nx[k] = load4(y0 + 1 + k); ne[k] = load_edge(y0 + 1 + k); }
        auto out = normals + ((size_t)y0 * W + cc);
        for (int y = y0; y < y1; y += kBatch) {
            f32x4_t cur[kBatch];
            float ce4[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) { cur[k] = nx[k]; ce4[k] = ne[k]; }
            if (y + kBatch < y1) {      // (wave-uniform) the next round's rows: in flight under this round's arithmetic
#pragma unroll
                for (int k = 0; k < kBatch; ++k) { nx[k] = load4(y + kBatch + 1 + k); ne[k] = load_edge(y + kBatch + 1 + k); }
            }
            float cos_lat[kBatch];      // (all of a round's scalar loads up front)
#pragma unroll
            for (int k = 0; k < kBatch; ++k) cos_lat[k] = trig_lat[2 * (y + k < H ? y + k : H - 1) + 1];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int gy = y + k;
                if (gy >= y1) break;      // (wave-uniform)
                const f32x4_t below = cur[k];
                const float ys = ys0 * cos_lat[k];
                const float left_edge = unif2(mid_edge, 0), right_edge = unif2(mid_edge, 63);
                const float hl = wave_from_left(mid.w, left_edge), hr = wave_from_right(mid.x, right_edge);
                if (kTables) {
                    mn = fminf(fminf(fminf(mn, mid.x), fminf(mid.y, mid.z)), fminf(mid.w, hr));
                    mx = fmaxf(fmaxf(fmaxf(mx, mid.x), fmaxf(mid.y, mid.z)), fmaxf(mid.w, hr));
                }
                const bool row_in = gy >= 1 && gy < H - 1;
                const float hL[4] = {hl, mid.x, mid.y, mid.z}, hR[4] = {mid.y, mid.z, mid.w, hr};
                const float hT[4] = {above.x, above.y, above.z, above.w}, hB[4] = {below.x, below.y, below.z, below.w};
                uint32_t tex[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gx = c0 + q;
                    const bool in = row_in && gx >= 1 && gx < W - 1;
                    uint32_t v = 0;
                    const bool settled = normal_texel_fast(xs, ys, hT[q], hL[q], hR[q], hB[q], v) || !in;
                    if (!settled) v = normal_texel(xs, ys, hT[q], hL[q], hR[q], hB[q]);      // the guard band and non-finite heights: the full chain
                    tex[q] = in ? v : 0u;
                }
                if (col_active) {
                    u32x4_t o;
                    o.x = tex[0]; o.y = tex[1]; o.z = tex[2]; o.w = tex[3];
                    __builtin_nontemporal_store(o, (__attribute__((address_space(1))) u32x4_t*)(out));
                }
                out += W;
                above = mid;
                mid = below;
                mid_edge = ce4[k];
            }
        }
        if (kTables) {      // the row below the wave's last one (row H - 1 again at the tile's end: the loads clamp)
            const float hr = wave_from_right(mid.x, unif2(mid_edge, 63));
            mn = fminf(fminf(fminf(mn, mid.x), fminf(mid.y, mid.z)), fminf(mid.w, hr));
            mx = fmaxf(fmaxf(fmaxf(mx, mid.x), fmaxf(mid.y, mid.z)), fmaxf(mid.w, hr));
        }
    }
    if (kTables) {
        // lane 15 b: the minimum / maximum over lanes 15 b .. 15 b + 14 (the last window ends at lane 59)
#pragma unroll
        for (int sh = 1; sh <= 4; sh <<= 1) { mn = fminf(mn, __shfl_down(mn, sh)); mx = fmaxf(mx, __shfl_down(mx, sh)); }
        mn = fminf(mn, __shfl_down(mn, 7));
        mx = fmaxf(mx, __shfl_down(mx, 7));
        if (lane < 60 && lane % 15 == 0) { s_part[wave][lane / 15][0] = mn; s_part[wave][lane / 15][1] = mx; }
        __syncthreads();
        if (threadIdx.x < 4u && by_ < by_count && 4u * bx_ + threadIdx.x < bx_count) {
            float lo = s_part[0][threadIdx.x][0], hi = s_part[0][threadIdx.x][1];
#pragma unroll
            for (int w2 = 1; w2 < kWaves; ++w2) { lo = fminf(lo, s_part[w2][threadIdx.x][0]); hi = fmaxf(hi, s_part[w2][threadIdx.x][1]); }
            float* const minmax = const_cast<float*>(t.block_minmax);
            const uint32_t blk = by_ * bx_count + 4u * bx_ + threadIdx.x;
            minmax[2 * blk] = lo;
            minmax[2 * blk + 1] = hi;
        }
    }
}

// Seam normals (compute_normals_edge_shader.wgsl:25-105): 64 texels (piece `block_x`) of seam job `job_index`.
__device__ __forceinline__ void normals_edge_body(const TileDev* __restrict__ tiles, const EdgeJob* __restrict__ jobs, int W, int H, uint32_t block_x,
                                                  uint32_t job_index) {
    const EdgeJob job = jobs[job_index];
    const TileDev &lt = tiles[job.lt], &rb = tiles[job.rb], &u = tiles[job.uni];
    const auto h_lt = TOPO_GLOBAL_F32(lt.heights);
    const auto h_rb = TOPO_GLOBAL_F32(rb.heights);
    const auto n_lt = TOPO_GLOBAL_U32_RW(lt.normals);
    const auto n_rb = TOPO_GLOBAL_U32_RW(rb.normals);
    const float raster_y = u.raster_y, model_y = u.model_y, scale_x = u.scale_x, scale_y = u.scale_y;
    const int id = (int)block_x * 64 + (int)threadIdx.x;
    if (id < 1 || id >= W - 1) return;
    const float xs = deg2rad(fabsf(scale_x)) * kR0;
    const float ys0 = deg2rad(fabsf(scale_y)) * kR0;
    if (!job.top_bottom) {
        if (id >= H - 1) return;   // the guard uses dimensions.x although id runs along y; see DESIGN.md
        const float latitude = ((float)id - raster_y) * -scale_y + model_y;
        const float ys = ys0 * cos_f(deg2rad(latitude));
        const int lx = W - 1, ly = id, rx = 0, ry = id;
        const float hT = h_lt[(size_t)(ly - 1) * W + lx], hL = h_lt[(size_t)ly * W + lx - 1];
        const float hR = h_rb[(size_t)ry * W + rx + 1], hB = h_lt[(size_t)(ly + 1) * W + lx];
        const uint32_t texel = normal_texel(xs, ys, hT, hL, hR, hB);
        n_lt[(size_t)ly * W + lx] = texel;
        n_rb[(size_t)ry * W + rx] = texel;
    } else {
        const float latitude = ((float)(H - 1) - raster_y) * -scale_y + model_y;
        const float ys = ys0 * cos_f(deg2rad(latitude));
        const int tx = id, ty = H - 1, bx = id, by = 0;
        const float hT = h_lt[(size_t)(ty - 1) * W + tx], hL = h_lt[(size_t)ty * W + tx - 1];
        const float hR = h_lt[(size_t)ty * W + tx + 1], hB = h_rb[(size_t)(by + 1) * W + bx];
        const uint32_t texel = normal_texel(xs, ys, hT, hL, hR, hB);
        n_lt[(size_t)ty * W + tx] = texel;
        n_rb[(size_t)by * W + bx] = texel;
    }
}

// Shared corner of a 2x2 block (compute_normals_corner_shader.wgsl:29-63), one job per lane; `top` comes from
// the bottom-right tile at (0, H-2) exactly as the shader reads it (:49).
__device__ __forceinline__ void normals_corner_body(const TileDev* __restrict__ tiles, const CornerJob* __restrict__ jobs, uint32_t n_jobs, int W, int H,
                                                    uint32_t block) {
    const uint32_t j = block * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    const CornerJob job = jobs[j];
    const TileDev &lt = tiles[job.lt], &rt = tiles[job.rt], &lb = tiles[job.lb], &rb = tiles[job.rb], &u = tiles[job.uni];
    const float latitude = ((float)(H - 1) - u.raster_y) * -u.scale_y + u.model_y;
    const float xs = deg2rad(fabsf(u.scale_x)) * kR0;
    const float ys = deg2rad(fabsf(u.scale_y)) * kR0 * cos_f(deg2rad(latitude));
    const float hT = TOPO_GLOBAL_F32(rb.heights)[(size_t)(H - 2) * W + 0];
    const float hL = TOPO_GLOBAL_F32(lt.heights)[(size_t)(H - 1) * W + (W - 2)];
    const float hR = TOPO_GLOBAL_F32(rt.heights)[(size_t)(H - 1) * W + 1];
    const float hB = TOPO_GLOBAL_F32(lb.heights)[(size_t)1 * W + (W - 1)];
    const uint32_t texel = normal_texel(xs, ys, hT, hL, hR, hB);
    TOPO_GLOBAL_U32_RW(lt.normals)[(size_t)(H - 1) * W + (W - 1)] = texel;
    TOPO_GLOBAL_U32_RW(rt.normals)[(size_t)(H - 1) * W + 0] = texel;
    TOPO_GLOBAL_U32_RW(lb.normals)[(size_t)0 * W + (W - 1)] = texel;
    TOPO_GLOBAL_U32_RW(rb.normals)[0] = texel;
}

// Both border passes in one launch (they write disjoint texels): workgroups [0, chunks * n_edges) take the seam jobs (chunks =
// 64-texel pieces of a seam), the rest the corner jobs, 64 per workgroup.
__global__ __launch_bounds__(64) void k_normals_border(const TileDev* __restrict__ tiles, const EdgeJob* __restrict__ edges, uint32_t n_edges,
                                                       uint32_t chunks, const CornerJob* __restrict__ corners, uint32_t n_corners, int W, int H) {
    const uint32_t n_edge_blocks = chunks * n_edges;
    if (blockIdx.x < n_edge_blocks) normals_edge_body(tiles, edges, W, H, blockIdx.x % chunks, blockIdx.x / chunks);
    else normals_corner_body(tiles, corners, n_corners, W, H, blockIdx.x - n_edge_blocks);
}

// ======================================================================================================
// frame phase
// ======================================================================================================

// Re-initialise the visibility buffer for a new frame: only the segments marked dirty are rewritten (and unmarked).
// A wave takes 64 segments at a time: one coalesced read of their marks, then one 512-byte store per marked segment.
// The queue counters come in two sets that alternate from frame to frame: this pass zeroes the set of the NEXT frame
// (`zero`: queue counters, status word, the far sub-lists' counters), so nothing that runs beside it -- the cull, which appends
// through this frame's set -- depends on it.  `counters`: this frame's set (the check build's status record).
__device__ __forceinline__ void clear_body(uint64_t* __restrict__ vis, uint8_t* __restrict__ dirty, size_t n, uint32_t* __restrict__ counters,
                                           uint32_t* __restrict__ zero, uint32_t block, uint32_t n_blocks) {
    if (block == 0)
        for (uint32_t i = threadIdx.x; i < kCounterWords; i += 256) zero[i] = 0;
    const uint32_t lane = threadIdx.x & 63;
    const size_t nseg = (n + 63) >> 6, wave = (size_t)block * 4 + (threadIdx.x >> 6), nwave = (size_t)n_blocks * 4;
    for (size_t g = wave * 64; g < nseg; g += nwave * 64) {
        const bool mine = g + lane < nseg && dirty[g + lane] != 0;
        uint64_t todo = __ballot(mine);
        if (mine) dirty[g + lane] = 0;
        while (todo) {
            const size_t seg = g + (size_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            if (TOPO_CHK(counters, seg * 64 + lane < ((n + 63) & ~(size_t)63), 3u, seg * 64 + lane))
                vis[seg * 64 + lane] = kVisClear;      // the buffer is allocated in whole segments
        }
    }
}
__global__ __launch_bounds__(256) void k_clear(uint64_t* __restrict__ vis, uint8_t* __restrict__ dirty, size_t n, uint32_t* __restrict__ counters,
                                               uint32_t* __restrict__ zero) {
    clear_body(vis, dirty, n, counters, zero, blockIdx.x, gridDim.x);
}

// Conservative frustum test of one raster block against one view, in f64.  A block is kept unless its
// bounding sphere (inflated by 64 m for the f32 noise of the real vertex path) lies wholly outside one of
// the six clip planes of camera_proj.  Culling is result-neutral: culled blocks cannot produce fragments.

// Clip plane `pl` of a column-major view-projection matrix as (a, b, c, d, |(a, b, c)|): 0..3 = w +- x, w +- y,
// 4 = near (z_clip >= 0), 5 = w - z.
__device__ __forceinline__ void clip_plane(const float* m, int pl, double out[5]) {
    double a, b, cc, d;
    const int row = pl >> 1;            // 0: x, 1: y, 2: z
    const double sgn = (pl & 1) ? -1.0 : 1.0;
    if (pl == 4) {                      // near: z_clip >= 0
        a = m[2]; b = m[6]; cc = m[10]; d = m[14];
    } else {                            // w +- row
        a = (double)m[3] + sgn * (double)m[row];
        b = (double)m[7] + sgn * (double)m[4 + row];
        cc = (double)m[11] + sgn * (double)m[8 + row];
        d = (double)m[15] + sgn * (double)m[12 + row];
    }
    out[0] = a; out[1] = b; out[2] = cc; out[3] = d;
    out[4] = sqrt(a * a + b * b + cc * cc);
}

// One lane per (view, tile, block).  f64 throughout; everything here is a conservative, result-neutral filter:
//  * frustum: the block's bounding sphere (inflated by 72 m for the f32 noise of the real vertex path) against the
//    six clip planes of camera_proj -- culled blocks cannot produce fragments;
//  * near/far split: blocks whose nearest possible view depth exceeds P.split_m become occlusion-test candidates
//    (FarItem) instead of work items; for them the lane also projects the eight corners of the block's bounding
//    slab -- the lat/lon rectangle of its vertices x [hmin - 1 m, hmax + 2 m + sagitta]: the flat-faced hull of those
//    eight points contains every triangle of the block (k_block_minmax measures the sagitta; blocks where it exceeds
//    1 m -- coarse tiles -- are never candidates) -- and records the pixel box (+-2 px; a sideways bulge of <= 1 m is
//    < 0.001 px beyond the split distance) and a lower bound of the depths
//    (z_ndc at the smallest corner w, minus 8/w: the f32 clip-space cancellation noise is ~1 clip unit).
// Emit a block the raster must visit.  With the occlusion filter on, such blocks are few and heavy (large triangles),
// so each is cut into strips of P.near_strip cell rows (the host picks 1, 2 or 4 by the size of the submission) to spread them over the resident waves:
// block = id | first cell row << 24 | rows << 28 (rows 0 = the whole block).
__device__ __forceinline__ void emit_near(const FrameParams& P, uint32_t view, uint32_t rank, uint32_t blk) {
    if (P.split_m > 0.0f) {
        const uint32_t by = blk / P.bx_count;
        const uint32_t cell_rows = min(kBCY, P.tile_h - 1 - by * kBCY);
        const uint32_t strip = P.near_strip, n = (cell_rows + strip - 1) / strip;
        const uint32_t base = atomicAdd(&P.counters[0], n);
        for (uint32_t k = 0; k < n; ++k)
            if (base + k < P.near_cap && TOPO_CHK(P.counters, blk < (1u << 24) && strip * k < 16u, 4u, blk))
                P.work[base + k] = WorkItem{(view << 16) | rank, blk | ((strip * k) << 24) | (min(strip, cell_rows - strip * k) << 28)};
        return;
    }
    const uint32_t slot = atomicAdd(&P.counters[0], 1u);
    if (slot < P.near_cap) P.work[slot] = WorkItem{(view << 16) | rank, blk};
}

__device__ __forceinline__ void cull_body(const FrameParams& P, uint32_t block) {
    const uint32_t blocks_per_tile = P.bx_count * P.by_count;
    const size_t total = (size_t)P.n_views * P.n_tiles * blocks_per_tile;
    const size_t gid0 = (size_t)block * blockDim.x, gid = gid0 + threadIdx.x;
    // the six clip planes (and their norms) of the first two views this workgroup can meet, once per workgroup
    __shared__ double s_plane[2][6][5];
    __shared__ uint32_t s_far[256], s_nfar;      // lanes whose block is an occlusion-test candidate
    const uint32_t view0 = (uint32_t)(gid0 / ((size_t)blocks_per_tile * P.n_tiles));
    if (threadIdx.x < 12 && view0 + threadIdx.x / 6 < P.n_views) clip_plane(P.views[view0 + threadIdx.x / 6].proj, threadIdx.x % 6, s_plane[threadIdx.x / 6][threadIdx.x % 6]);
    if (threadIdx.x == 0) s_nfar = 0;
    __syncthreads();
    // ---- phase A, one lane per (view, tile, block): frustum test, then near / far classification
    if (gid < total) {
        const uint32_t blk = (uint32_t)(gid % blocks_per_tile);
        const uint32_t rank = (uint32_t)((gid / blocks_per_tile) % P.n_tiles);
        const uint32_t view = (uint32_t)(gid / ((size_t)blocks_per_tile * P.n_tiles));
        const TileDev& t = P.tiles[rank];
        const double hmin = (double)t.block_minmax[2 * blk], hmax = (double)t.block_minmax[2 * blk + 1];
        const double* bs = t.block_bounds + (size_t)blk * 4;      // bounding sphere from the load phase
        const double c[3] = {bs[0], bs[1], bs[2]}, radius = bs[3];
        const float* m = P.views[view].proj;
        bool keep = true;
        for (int pl = 0; pl < 6 && keep; ++pl) {
            double own[5];
            const double* q = s_plane[view - view0 < 2 ? view - view0 : 0][pl];
            if (view - view0 >= 2) {            // tiny mosaics: more than two views per workgroup
                clip_plane(m, pl, own);
                q = own;
            }
            const double dist = q[0] * c[0] + q[1] * c[1] + q[2] * c[2] + q[3];
            if (dist < -radius * q[4]) keep = false;
        }
        const bool sane = hmin <= hmax;         // NaN heights: no filtering at all, the raster path deals with it
        if (!sane) keep = true;
        if (keep) {
            // view depth of the nearest point the block can contain
            const double wn = sqrt((double)m[3] * m[3] + (double)m[7] * m[7] + (double)m[11] * m[11]);
            const double w_near = ((double)m[3] * c[0] + (double)m[7] * c[1] + (double)m[11] * c[2] + (double)m[15]) - radius * wn;
            // (coarse tiles: a block whose curvature exceeds the slab's 1 m allowance is always rastered, never filtered)
            const double sagitta = t.block_bounds[(size_t)blocks_per_tile * 16 + blk];
            if (sane && P.split_m > 0.0f && w_near > (double)P.split_m && sagitta <= 1.0) {
                const uint32_t fslot = atomicAdd(&s_nfar, 1u);
                if (TOPO_CHK(P.counters, fslot < 256u, 5u, fslot)) s_far[fslot] = threadIdx.x;
            }
            else emit_near(P, view, rank, blk);
        }
    }
    __syncthreads();
    // ---- phase B, one lane per candidate (they are ~10 % of the lanes, scattered: handled in place, every wave would
    // pay for the f64 projection of the eight slab corners)
    for (uint32_t i = threadIdx.x; i < s_nfar; i += blockDim.x) {
        const size_t g = gid0 + s_far[i];
        const uint32_t blk = (uint32_t)(g % blocks_per_tile);
        const uint32_t rank = (uint32_t)((g / blocks_per_tile) % P.n_tiles);
        const uint32_t view = (uint32_t)(g / ((size_t)blocks_per_tile * P.n_tiles));
        const TileDev& t = P.tiles[rank];
        const double hmin = (double)t.block_minmax[2 * blk], hmax = (double)t.block_minmax[2 * blk + 1];
        const double* bb = t.block_bounds + (size_t)blocks_per_tile * 4 + (size_t)blk * 12;   // corner directions
        const float* m = P.views[view].proj;
        double bxlo = 1e30, bxhi = -1e30, bylo = 1e30, byhi = -1e30, wmin = 1e30, zclip_at_wmin = 0.0;
        const double hs[2] = {hmin - 1.0, hmax + 2.0 + t.block_bounds[(size_t)blocks_per_tile * 16 + blk]};      // + the patch's sagitta (<= 1 m here)
#pragma unroll 1      // (eight corners unrolled kept 124 registers live; the loop form needs half, and the clear running beside this kernel gets the waves)
        for (int k = 0; k < 8; ++k) {
            const double R = (double)kR0 + (k < 4 ? hs[0] : hs[1]);
            const double px = R * bb[3 * (k & 3)], py = R * bb[3 * (k & 3) + 1], pz = R * bb[3 * (k & 3) + 2];
            const double cx = (double)m[0] * px + (double)m[4] * py + (double)m[8] * pz + (double)m[12];
            const double cy = (double)m[1] * px + (double)m[5] * py + (double)m[9] * pz + (double)m[13];
            const double cz = (double)m[2] * px + (double)m[6] * py + (double)m[10] * pz + (double)m[14];
            const double cw = (double)m[3] * px + (double)m[7] * py + (double)m[11] * pz + (double)m[15];
            const double icw = 1.0 / cw;      // one f64 division per corner (the +-2 px margin dwarfs the extra rounding)
            const double sx = (cx * icw * 0.5 + 0.5) * (double)P.W, sy = (0.5 - cy * icw * 0.5) * (double)P.H;
            bxlo = sx < bxlo ? sx : bxlo; bxhi = sx > bxhi ? sx : bxhi;
            bylo = sy < bylo ? sy : bylo; byhi = sy > byhi ? sy : byhi;
            if (cw < wmin) { wmin = cw; zclip_at_wmin = cz; }
        }
        // z_ndc = a + b / w (b < 0) is a function of w alone and w is linear in position, so over the slab's convex
        // hull its minimum sits at the corner with the smallest w.  The real pipeline computes z_clip and w as f32
        // fma chains over ~6.4e6-sized terms: each carries up to ~1 (metre-sized clip units) of cancellation noise,
        // i.e. z_ndc is only good to ~2 / w.  Shave 8 / w.
        const double zmin = (zclip_at_wmin - 8.0) / wmin;
        if (!(wmin > 1000.0 && zmin > 0.0 && zmin < 1.0)) {
            emit_near(P, view, rank, blk);          // no usable bound: rasterise it with the near blocks
            continue;
        }
        const int32_t ix0 = max((int32_t)floor(bxlo) - 2, 0), ix1 = min((int32_t)ceil(bxhi) + 2, P.W - 1);
        const int32_t iy0 = max((int32_t)floor(bylo) - 2, 0), iy1 = min((int32_t)ceil(byhi) + 2, P.H - 1);
        if (ix0 > ix1 || iy0 > iy1) continue;       // wholly outside the target even with the margin
        float zf = (float)zmin;
        if ((double)zf > zmin) zf = bits_f(f_bits(zf) - 1u);      // round down
        // 10^5 candidates appended through ONE counter cost this kernel 12 of its 43 us (atomics on one address are served one
        // at a time, ~12 ns each, however the waves aggregate them): the list is kept as kFarLists sub-lists, workgroup b
        // appending to sub-list b % kFarLists
        const uint32_t q = block % kFarLists, slot = atomicAdd(&P.counters[16 + 16 * q], 1u);
        if (TOPO_CHK(P.counters, slot < P.far_sub_cap, 5u, slot)) {
            FarItem fi;
            fi.view_rank = (view << 16) | rank; fi.block = blk;
            fi.x0 = (uint16_t)ix0; fi.x1 = (uint16_t)ix1; fi.y0 = (uint16_t)iy0; fi.y1 = (uint16_t)iy1;
            fi.zmin_bits = f_bits(zf);
            P.far[(size_t)q * P.far_sub_cap + slot] = fi;
        }
    }
}

__global__ __launch_bounds__(256) void k_cull(FrameParams P) { cull_body(P, blockIdx.x); }
// The clear and the cull of a frame in ONE launch: they share nothing (the clear rewrites visibility keys and zeroes the NEXT
// frame's counters, the cull reads the load-time tables and appends through this frame's counters), one is bound by its
// stores, the other by f64 arithmetic and gathers -- side by side they take what the clear takes alone (c4: 0.046 + 0.031 ->
// 0.05 ms).
// `pack_words` != 0: the submission's view constants ride in this launch's own argument segment (`pack`: up to kPackViews views, 22
// words each).  The cull reads them there -- the segment is ordinary device-visible memory behind a constant-address-space pointer --
// and the launch's first workgroup copies them into the device slot the frame's later kernels read (P.views): no upload in front of
// the frame, not even a kernel's.
struct ClearCullArgs {          // the kernel's parameter list as the argument segment lays it out (natural alignment, in order)
    FrameParams P;
    uint32_t n_cull_blocks, n_clear_blocks;
    uint32_t* zero;
    ViewPack pack;
    uint32_t pack_words;
};
__global__ __launch_bounds__(256) void k_clear_cull(FrameParams P, uint32_t n_cull_blocks, uint32_t n_clear_blocks, uint32_t* __restrict__ zero, ViewPack pack,
                                                    uint32_t pack_words) {
    const auto seg = (const __attribute__((address_space(4))) uint8_t*)__builtin_amdgcn_kernarg_segment_ptr();
    if (pack_words) {
        const auto src = (const __attribute__((address_space(4))) uint32_t*)(seg + offsetof(ClearCullArgs, pack));
        if (blockIdx.x == 0 && threadIdx.x < pack_words) const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(P.views))[threadIdx.x] = src[threadIdx.x];
        P.views = (const ViewDev*)(const void*)(seg + offsetof(ClearCullArgs, pack));
    }
    // the two kinds of workgroup interleaved evenly along the launch order (all of one kind first would run them one after the other:
    // a launch's workgroups start in order)
    const uint32_t total = n_cull_blocks + n_clear_blocks;
    const uint32_t before = (uint32_t)((uint64_t)blockIdx.x * n_clear_blocks / total), upto = (uint32_t)((uint64_t)(blockIdx.x + 1u) * n_clear_blocks / total);
    if (upto > before) clear_body(P.vis, P.dirty, (size_t)P.n_views * P.W * P.H, P.counters, zero, before, n_clear_blocks);
    else cull_body(P, blockIdx.x - before);
}

// One wave per far candidate: the block is dropped iff EVERY pixel of its footprint already holds a depth below
// the block's lower bound -- then none of its fragments could pass `Less`.  Pixels in the gaps between tiles, or
// anywhere nothing nearer has been drawn, keep the block alive, so the filter is exact by construction.
#ifndef TOPO_OCC_ROWS
#define TOPO_OCC_ROWS 4
#endif
constexpr uint32_t kOccRows = TOPO_OCC_ROWS;      // rows of a footprint whose depths are in flight before the wave votes
__global__ __launch_bounds__(256) void k_occlusion(FrameParams P) {
    // between the two raster phases: the rare/big queues keep growing, the second phase starts where the first ended
    // (nothing enqueues while this kernel runs, and the consumers of the marks are launched after it)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.counters[6] = P.counters[1];
        P.counters[7] = P.counters[3];
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_global = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave_count = gridDim.x * 4;
    // The sub-lists are walked as one list: lane k of every wave holds the number of entries in sub-lists 0 .. k (an inclusive
    // scan of the 64 counts), entry g of the whole lies in the sub-list q with incl[q - 1] <= g < incl[q].
    static_assert(kFarLists == 64, "one sub-list per lane");
    uint32_t incl = min(P.counters[16 + 16 * lane], P.far_sub_cap);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o) incl += up;
    }
    const uint32_t count = (uint32_t)__shfl((int)incl, 63);
    if (blockIdx.x == 0 && threadIdx.x == 0) P.counters[4] = count;      // the candidate count, for the frame's statistics
    auto entry = [&](uint32_t g) -> const FarItem& {
        const uint32_t q = (uint32_t)__popcll(__ballot(incl <= g));      // sub-lists that end at or before g
        const uint32_t start = q ? (uint32_t)__shfl((int)incl, (int)q - 1) : 0u;
        return P.far[(size_t)q * P.far_sub_cap + (g - start)];
    };
    if (count == 0) return;
    FarItem fi_next = entry(wave_global < count ? wave_global : 0u);      // (the next candidate's record is fetched under the current one's scan)
    for (uint32_t item = wave_global; item < count; item += wave_count) {
        const FarItem fi = fi_next;
        fi_next = entry(item + wave_count < count ? item + wave_count : item);
        const uint64_t* vis = P.vis + (size_t)(fi.view_rank >> 16) * P.W * P.H;
        // footprints are wide and flat: lanes run along x, kOccRows rows per round so that as many loads are in flight
        // before the first wave-wide vote
        bool visible = false;
        for (uint32_t x = fi.x0; x <= fi.x1 && !visible; x += 64) {
            const uint32_t px = min(x + lane, (uint32_t)fi.x1);       // surplus lanes re-test the last column
            for (uint32_t y = fi.y0; y <= fi.y1; y += kOccRows) {
                uint32_t d[kOccRows];
#pragma unroll
                for (uint32_t k = 0; k < kOccRows; ++k) {
                    const size_t at = (size_t)min(y + k, (uint32_t)fi.y1) * P.W + px;
                    d[k] = TOPO_CHK(P.counters, at < (size_t)P.W * P.H && (fi.view_rank >> 16) < P.n_views, 6u, at) ? (uint32_t)(vis[at] >> 32) : 0u;
                }
                bool open = false;
#pragma unroll
                for (uint32_t k = 0; k < kOccRows; ++k) open |= d[k] >= fi.zmin_bits;
                if (__any(open)) { visible = true; break; }
            }
        }
        if (visible && lane == 0) {
            // survivors are few: cut them into strips like the near blocks, or the second raster phase runs on a
            // fraction of the resident waves
            const uint32_t blk = fi.block, by = blk / P.bx_count;
            const uint32_t cell_rows = min(kBCY, P.tile_h - 1 - by * kBCY);
            const uint32_t strip = P.near_strip, n = (cell_rows + strip - 1) / strip;
            const uint32_t base = atomicAdd(&P.counters[5], n);
            for (uint32_t k = 0; k < n; ++k)
                if (base + k < P.near_cap)
                    P.work2[base + k] = WorkItem{fi.view_rank, blk | ((strip * k) << 24) | (min(strip, cell_rows - strip * k) << 28)};
        }
    }
}


// ---- raster ------------------------------------------------------------------------------------------

// Pixel loop of the generic (int64) path.
__device__ __forceinline__ void raster_box(const TriSetup& ts, const Vis& vis, int32_t W, uint32_t id,
                                           int32_t px0, int32_t px1, int32_t py0, int32_t py1) {
    for (int32_t py = py0; py <= py1; ++py)
        for (int32_t px = px0; px <= px1; ++px) {
            float z, b[3];
            if (triangle_pixel(ts, px, py, z, b)) vis_min(vis, (size_t)py * W + px, vis_key(z, id));
        }
}

// Hand a triangle whose pixel box is larger than 4x4 to k_raster_big: one BigItem per overlapped 64x64 px
// region.  Returns false when the queue is full (the caller then rasterises in-lane).
__device__ bool enqueue_big(const FrameParams& P, uint32_t view, uint32_t id, const SVert& s0, const SVert& s1, const SVert& s2,
                            int32_t px0, int32_t px1, int32_t py0, int32_t py1) {
    const int32_t rx0 = px0 >> 6, rx1 = px1 >> 6, ry0 = py0 >> 6, ry1 = py1 >> 6;
    const uint32_t n = (uint32_t)((rx1 - rx0 + 1) * (ry1 - ry0 + 1));
    const uint32_t base = atomicAdd(&P.counters[1], n);
    BigItem it;
    it.view = view;
    it.id = id;
    it.X[0] = s0.X; it.X[1] = s1.X; it.X[2] = s2.X;
    it.Y[0] = s0.Y; it.Y[1] = s1.Y; it.Y[2] = s2.Y;
    it.z[0] = s0.z; it.z[1] = s1.z; it.z[2] = s2.z;
    if (base >= P.big_cap || n > P.big_cap - base) {
        atomicOr(&P.counters[2], kStatusBigOverflow);
        // neutralise whatever part of the reservation lies inside the queue
        it.id = kNoTri;
        it.region = 0;
        for (uint32_t k = base; k < P.big_cap && k - base < n; ++k) P.big[k] = it;
        return false;
    }
    uint32_t k = base;
    for (int32_t ry = ry0; ry <= ry1; ++ry)
        for (int32_t rx = rx0; rx <= rx1; ++rx) {
            it.region = ((uint32_t)ry << 16) | (uint32_t)rx;
            if (TOPO_CHK(P.counters, k < P.big_cap && rx >= 0 && ry >= 0 && rx * 64 < P.W && ry * 64 < P.H, 7u, k)) P.big[k] = it;
            ++k;
        }
    return true;
}

// Triangles the lean kernel does not handle go to k_raster_rare.
__device__ __forceinline__ void enqueue_rare(const FrameParams& P, uint32_t view, uint32_t draw) {
    const uint32_t slot = atomicAdd(&P.counters[3], 1u);
    if (slot < P.rare_cap) P.rare[slot] = RareItem{view, draw};
    else atomicOr(&P.counters[2], kStatusRareOverflow);
}

// Fragment staging: lanes of k_raster do not touch the visibility buffer while they walk their triangles (the
// walk is divergent: a few lanes would issue one atomic each per iteration); they append (pixel, key) pairs to a
// per-wave LDS list which the wave then drains densely, one atomic per lane and instruction.
#ifndef TOPO_FRAG_CAP
#define TOPO_FRAG_CAP 128
#endif
constexpr uint32_t kFragCap = TOPO_FRAG_CAP;
struct FragList {
    uint32_t count;
    uint32_t pix[kFragCap];
    uint64_t key[kFragCap];
};

#ifndef TOPO_INLANE_ROWS
#define TOPO_INLANE_ROWS 5
#endif
#ifndef TOPO_INLANE_COLS
#define TOPO_INLANE_COLS 24
#endif
constexpr int32_t kInlaneRows = TOPO_INLANE_ROWS, kInlaneCols = TOPO_INLANE_COLS;

__device__ __forceinline__ void frag_push(FragList& fl, const Vis& vis, uint32_t pix, uint64_t key) {
    const uint32_t slot = atomicAdd(&fl.count, 1u);
    if (slot < kFragCap) {
        fl.pix[slot] = pix;
        fl.key[slot] = key;
    } else {
        vis_min(vis, pix, key);   // list full: fall back to the direct path
    }
}

// ---- in-wave triangle compaction -------------------------------------------------------------------------
// Far-field cells are sub-pixel: nine triangles in ten die in the early tests (back face, no pixel centre in the
// box).  Walking the survivors' pixels in the lane that found them would leave 58 of 64 lanes idle through every
// loop, so k_raster works in two stages: stage 1 classifies the two triangles of each lane's cell and appends the
// survivors to a per-wave LDS list (slot = running count + rank among the pushing lanes: no atomics); whenever
// the list holds a wave's worth, stage 2 pops 64 of them, one per lane, and walks their pixel rows.
constexpr uint32_t kTriCap = 128;
struct TriList {                 // structure of arrays: lane-consecutive entries hit consecutive banks
    int32_t X0[kTriCap], Y0[kTriCap], X1[kTriCap], Y1[kTriCap], X2[kTriCap], Y2[kTriCap];
    float z0[kTriCap], z1[kTriCap], z2[kTriCap];
    uint32_t id[kTriCap];
};

// Stage 1.  A triangle whose three snapped vertices span < 64 px fits int32 (|delta| < 2^14, so every product is
// < 2^28; 24-bit multiplies give the exact integers triangle_setup computes in int64).  Returns true when the
// triangle is to be walked in-wave (front-facing, its pixel box holds a centre and is at most kInlaneRows x
// kInlaneCols); larger boxes go to k_raster_big, >= 64 px spans to k_raster_rare.
__device__ __forceinline__ bool classify_small(const FrameParams& P, const SVert& s0, const SVert& s1, const SVert& s2, uint32_t view,
                                               uint32_t id) {
    const int32_t X0 = s0.X, Y0 = s0.Y, X1 = s1.X, Y1 = s1.Y, X2 = s2.X, Y2 = s2.Y;
    const int32_t mnx = min(X0, min(X1, X2)), mxx = max(X0, max(X1, X2));
    const int32_t mny = min(Y0, min(Y1, Y2)), mxy = max(Y0, max(Y1, Y2));
    if ((mxx - mnx) >= (1 << 14) || (mxy - mny) >= (1 << 14)) {
        enqueue_rare(P, view, id >> 1);
        return false;
    }
    const int32_t area2 = __mul24(X1 - X0, Y2 - Y0) - __mul24(Y1 - Y0, X2 - X0);
    if (area2 >= 0) return false;
    const int32_t px0 = max((mnx + 127) >> 8, 0), px1 = min((mxx - 128) >> 8, P.W - 1);
    const int32_t py0 = max((mny + 127) >> 8, 0), py1 = min((mxy - 128) >> 8, P.H - 1);
    if (px0 > px1 || py0 > py1) return false;
    if (py1 - py0 >= kInlaneRows || px1 - px0 >= kInlaneCols) {
        if (!enqueue_big(P, view, id, s0, s1, s2, px0, px1, py0, py1)) enqueue_rare(P, view, id >> 1);
        return false;
    }
    return true;
}

// Stage 2 = raster_rows() (topo_pipeline.h), one listed triangle per lane, fragments into the per-wave LDS list.

// Append this lane's triangle (if `push`) behind the `n` entries already listed; returns the new count.  Runs in
// wave-uniform control flow: the slot is n + the lane's rank among the pushing lanes.
__device__ __forceinline__ uint32_t tri_push(TriList& tl, uint32_t n, bool push, const SVert& s0, const SVert& s1, const SVert& s2,
                                             uint32_t id, uint32_t* vis_counters) {
    const uint64_t mask = __ballot(push);
    if (push) {
        const uint32_t slot = n + __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
        if (!TOPO_CHK(vis_counters, slot < kTriCap, 8u, slot)) return n;
        tl.X0[slot] = s0.X; tl.Y0[slot] = s0.Y; tl.X1[slot] = s1.X; tl.Y1[slot] = s1.Y; tl.X2[slot] = s2.X; tl.Y2[slot] = s2.Y;
        tl.z0[slot] = s0.z; tl.z1[slot] = s1.z; tl.z2[slot] = s2.z;
        tl.id[slot] = id;
    }
    return n + (uint32_t)__popcll(mask);
}

// Pop up to 64 listed triangles (the newest ones), one per lane, walk them, then flush the fragment list if it
// holds a wave's worth (or unconditionally when `flush`).  Returns the remaining count.
__device__ __forceinline__ uint32_t tri_drain(TriList& tl, FragList& fl, const Vis& vis, int32_t W, int32_t H, uint32_t n,
                                              bool flush) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t take = min(n, 64u), base = n - take;
    if (lane < take) {
        const uint32_t e = base + lane;
        raster_rows(W, H, tl.X0[e], tl.Y0[e], tl.X1[e], tl.Y1[e], tl.X2[e], tl.Y2[e], tl.z0[e], tl.z1[e], tl.z2[e], tl.id[e],
                    [&](uint32_t pix, uint64_t key) { frag_push(fl, vis, pix, key); });
    }
    const uint32_t nfrag = min(fl.count, kFragCap);
    if (nfrag >= 64 || (flush && nfrag > 0)) {
        for (uint32_t f = lane; f < nfrag; f += 64) vis_min(vis, fl.pix[f], fl.key[f]);
        if (lane == 0) fl.count = 0;
    }
    return base;
}

__device__ __forceinline__ SVert shfl_down1(const SVert& v) {
    SVert o;
    o.X = __shfl_down(v.X, 1);
    o.Y = __shfl_down(v.Y, 1);
    o.z = __shfl_down(v.z, 1);
    o.flag = __shfl_down(v.flag, 1);
    return o;
}

// One WAVE per surviving (view, tile, block) -- no workgroup barriers, no LDS vertex staging.  Lane i owns
// vertex column x0+i of the block (61 of 64 lanes); the wave walks the block's 16 vertex rows top to bottom,
// each lane transforming one vertex per row (coalesced 244-B row reads of the DEM, the next row's heights
// prefetched while the current row is processed; sin/cos of the longitude once per lane, of the latitudes once
// per row on lanes 0..15 and broadcast).  The previous row stays in registers; lane i then owns grid cell
// (x0+i, row-1): its four corners are its own two vertices and lane i+1's two, fetched with wave shuffles.
// Fragments go to a per-wave LDS list that the wave drains densely (one fragment per lane) whenever a row left
// more than a wave's worth in it.
#ifndef TOPO_RASTER_WAVES
#define TOPO_RASTER_WAVES 5
#endif
__global__ __launch_bounds__(256, TOPO_RASTER_WAVES) void k_raster(FrameParams P, int phase) {
    __shared__ FragList s_fl[4];
    __shared__ TriList s_tl[4];
    const WorkItem* __restrict__ work = phase == 0 ? P.work : P.work2;
    uint32_t count = P.counters[phase == 0 ? 0 : 5];
    const uint32_t cap = P.near_cap;      // both lists hold strips
    if (count > cap) count = cap;
    // the wave index is wave-uniform: say so (readfirstlane), or the compiler treats everything derived from the
    // work item -- the view matrix, the tile descriptor -- as per-lane data and re-loads it with vector loads.
    // Waves stride statically over the work list (pulling chunks from an atomic cursor measured 17 % slower).
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    FragList& fl = s_fl[wave];
    TriList& tl = s_tl[wave];
    uint32_t ntri = 0;           // triangles waiting in tl (wave-uniform)
    const uint32_t wave_global = blockIdx.x * 4 + wave, wave_count = gridDim.x * 4;
    for (uint32_t item = wave_global; item < count; item += wave_count) {
        // The work item, the tile's descriptor and the view matrix are wave-uniform and written by EARLIER kernels: read through
        // the constant address space they come over the scalar data path (s_load), in two round trips -- item, then descriptor
        // and matrix together -- that wait on lgkmcnt.  As vector loads they were four dependent trips behind s_waitcnt vmcnt(0),
        // each of which also waits for every visibility atomic the wave still has in flight from the strip before.
        typedef const __attribute__((address_space(4))) uint32_t* cu32;
        typedef const __attribute__((address_space(4))) float* cf32;
        const cu32 wi_c = (cu32)(const void*)(work + item);
        const WorkItem wi = {wi_c[0], wi_c[1]};
        const uint32_t view_idx = wi.view_rank >> 16, rank = wi.view_rank & 0xFFFFu;
        TileDev t;      // the fields this kernel reads (the rest stay unset)
        {
            const auto tc = (const __attribute__((address_space(4))) TileDev*)(const void*)(P.tiles + rank);
            t.heights = tc->heights;
            t.raster_x = tc->raster_x; t.raster_y = tc->raster_y;
            t.model_x = tc->model_x; t.model_y = tc->model_y;
            t.scale_x = tc->scale_x; t.scale_y = tc->scale_y;
        }
        const cf32 view_proj = (cf32)(const void*)P.views[view_idx].proj;
        const uint32_t blk_id = wi.block & 0xFFFFFFu, strip_first = (wi.block >> 24) & 15u, strip_rows = wi.block >> 28;
        const uint32_t bx = blk_id % P.bx_count, by = blk_id / P.bx_count;
        const uint32_t x0 = bx * kBCX, y0 = by * kBCY + strip_first;   // strip_rows == 0: the whole block
        const uint32_t nrows = min(strip_rows ? strip_rows + 1 : kVY, P.tile_h - y0);   // vertex rows of this item
        const uint32_t ncx = min(kBCX, P.tile_w - 1 - x0);       // cells per row
        const uint32_t vx = x0 + lane;
        const bool vcol = lane < kVX && vx < P.tile_w;
        const Vis vis = view_vis(P, view_idx);
        // the view matrix, read once per block into scalar registers: left to the compiler it is re-loaded from
        // memory every row (it cannot prove the visibility-buffer atomics do not alias it) behind an
        // s_waitcnt vmcnt(0) that also drains the height prefetch
        float proj[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) proj[q] = view_proj[q];
        if (lane == 0) fl.count = 0;
        float slo, clo, lat_s = 0.0f, lat_c = 0.0f;
        sincos_f(vertex_lon(t, vcol ? vx : x0), slo, clo);
        if (lane < nrows) sincos_f(vertex_lat(t, y0 + lane), lat_s, lat_c);
        const auto hcol = TOPO_GLOBAL_F32(t.heights) + (size_t)y0 * P.tile_w + (vcol ? vx : x0);   // global, not flat, loads
        // heights are prefetched four rows ahead (a rotating register window): one 244-B row read per wave is
        // too little to have in flight at a time
        (void)TOPO_CHK(P.counters, rank < P.n_tiles && view_idx < P.n_views && y0 + nrows <= P.tile_h && (vcol ? vx : x0) < P.tile_w && nrows >= 1u, 9u,
                       ((uint64_t)y0 << 32) | x0);
        float h0 = hcol[0];
        float h1 = nrows > 1 ? hcol[(size_t)1 * P.tile_w] : 0.0f;
        float h2 = nrows > 2 ? hcol[(size_t)2 * P.tile_w] : 0.0f;
        float h3 = nrows > 3 ? hcol[(size_t)3 * P.tile_w] : 0.0f;
        SVert prev;
        prev.X = 0; prev.Y = 0; prev.z = 0.0f; prev.flag = kVtxNear;
        for (uint32_t r = 0; r < nrows; ++r) {
            const float h = h0;
            h0 = h1; h1 = h2; h2 = h3;
            if (r + 4 < nrows) h3 = hcol[(size_t)(r + 4) * P.tile_w];
            const float sla = __shfl(lat_s, (int)r), cla = __shfl(lat_c, (int)r);
            SVert cur;
            cur.X = 0; cur.Y = 0; cur.z = 0.0f; cur.flag = kVtxNear;
            if (vcol) {
                const f3 p = world_from_sincos(h, sla, cla, slo, clo);
                float clip[4];
                mat4_point(proj, p.x, p.y, p.z, clip);
                clip_to_screen(clip, (float)P.W, (float)P.H, cur);
            }
            if (r > 0) {
                // cell (i, j) = (x0 + lane, y0 + r - 1): a = (i,j) b = (i,j+1) c = (i+1,j) d = (i+1,j+1)
                const SVert cc = shfl_down1(prev), d = shfl_down1(cur);
                // Quad-level reject (result-neutral): if all four corners are plain vertices and their common pixel
                // box holds no pixel centre, neither triangle can produce a fragment.
                bool live = lane < ncx;
                if (live && (prev.flag | cur.flag | cc.flag | d.flag) == kVtxOk) {
                    const int32_t qx0 = min(min(prev.X, cur.X), min(cc.X, d.X)), qx1 = max(max(prev.X, cur.X), max(cc.X, d.X));
                    const int32_t qy0 = min(min(prev.Y, cur.Y), min(cc.Y, d.Y)), qy1 = max(max(prev.Y, cur.Y), max(cc.Y, d.Y));
                    const int32_t bx0 = max((qx0 + 127) >> 8, 0), bx1 = min((qx1 - 128) >> 8, P.W - 1);
                    const int32_t by0 = max((qy0 + 127) >> 8, 0), by1 = min((qy1 - 128) >> 8, P.H - 1);
                    live = bx0 <= bx1 && by0 <= by1;
                }
                const SVert &a = prev, &b = cur;
                const uint32_t i = x0 + lane, j = y0 + r - 1;
                const bool even = ((i + j) & 1u) == 0;
                const uint32_t tri0 = (i * (P.tile_h - 1) + j) * 2;
#pragma unroll
                for (uint32_t k = 0; k < 2; ++k) {
                    const SVert& s0 = k == 0 ? a : d;
                    const SVert& s1 = k == 0 ? b : cc;
                    const SVert& s2 = k == 0 ? (even ? d : cc) : (even ? a : b);
                    const uint32_t draw = rank * P.tris_per_tile + tri0 + k;
                    bool push = false;
                    if (live) {
                        const int fg = s0.flag | s1.flag | s2.flag;
                        if (fg == kVtxOk) {
                            push = classify_small(P, s0, s1, s2, view_idx, draw << 1);
                        } else if (fg & kVtxNear) {
                            const int nnear = (s0.flag == kVtxNear) + (s1.flag == kVtxNear) + (s2.flag == kVtxNear);
                            if (nnear != 3) enqueue_rare(P, view_idx, draw);
                        }   // else: guard band -> primitive discarded
                    }
                    ntri = tri_push(tl, ntri, push, s0, s1, s2, draw << 1, P.counters);
                    if (ntri >= 64) ntri = tri_drain(tl, fl, vis, P.W, P.H, ntri, false);
                }
            }
            prev = cur;
        }
        // block end: the list refers to this block's view, so it is emptied before the next item
        while (ntri > 0) ntri = tri_drain(tl, fl, vis, P.W, P.H, ntri, true);
        const uint32_t nfrag = min(fl.count, kFragCap);
        for (uint32_t f = lane; f < nfrag; f += 64) vis_min(vis, fl.pix[f], fl.key[f]);
        if (lane == 0) fl.count = 0;
    }
}

// One lane per RareItem: the generic exact path (near clipping, int64 setup).  Boxes up to 4x4 px are
// rasterised in-lane, larger ones go to the big queue (or, if that is full, are rasterised here as well).
// A triangle with at least kCoopRegions regions has its BigItems written by the WHOLE wave, 64 regions at a time: a lane's own loop
// over the regions of a triangle that covers a good part of the target (the near field's largest, cut by the near plane: a thousand
// regions and more) was this kernel's duration -- ~20 instructions per region on ONE lane, while the other 15 000 triangles had long
// been done.
constexpr uint32_t kCoopRegions = 24;
__global__ __launch_bounds__(256) void k_raster_rare(FrameParams P) {
    uint32_t count = P.counters[3];
    if (count > P.rare_cap) count = P.rare_cap;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t item = P.counters[7] + blockIdx.x * blockDim.x + threadIdx.x; item < count; item += gridDim.x * blockDim.x) {
        const RareItem ri = P.rare[item];
        const uint32_t rank = fastdiv(ri.draw, P.div_tris), tri = ri.draw - rank * P.tris_per_tile;
        if (!TOPO_CHK(P.counters, rank < P.n_tiles && ri.view < P.n_views, 10u, ri.draw)) continue;
        const Vis vis = view_vis(P, ri.view);
        for (uint32_t fan = 0; fan < 2; ++fan) {
            ResolvedTri r;
            const bool has = resolve_triangle(P.tiles[rank], P.tile_w, P.div_hm1, P.tile_h - 1, P.views[ri.view], P.W, P.H, tri, fan, r);
            const TriSetup& ts = r.ts;
            const uint32_t id = (ri.draw << 1) | fan;
            const int32_t nx = has ? ts.px1 - ts.px0 + 1 : 0, ny = has ? ts.py1 - ts.py0 + 1 : 0;
            const bool small = nx <= 4 && ny <= 4;
            const int32_t rx0 = has ? ts.px0 >> 6 : 0, rx1 = has ? ts.px1 >> 6 : 0, ry0 = has ? ts.py0 >> 6 : 0, ry1 = has ? ts.py1 >> 6 : 0;
            const uint32_t rw = (uint32_t)(rx1 - rx0 + 1), n_regions = rw * (uint32_t)(ry1 - ry0 + 1);
            const bool coop = has && !small && n_regions >= kCoopRegions && rw <= 256u;
            bool in_lane = has && small;      // rasterised by this lane itself: boxes up to 4 x 4 px, and whatever the queue has no room for
            if (has && !small && !coop) in_lane = !enqueue_big(P, ri.view, id, r.s[0], r.s[1], r.s[2], ts.px0, ts.px1, ts.py0, ts.py1);
            // ---- the wave's large jobs, one after the other, every lane that is still in this loop taking part
            uint64_t jobs = __ballot(coop);
            if (jobs) {
                const uint64_t act = __ballot(true);
                const uint32_t n_act = (uint32_t)__popcll(act), mine = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
                while (jobs) {
                    const int L = __builtin_ctzll(jobs);
                    jobs &= jobs - 1ull;
                    auto from = [&](int32_t v) { return __builtin_amdgcn_readlane(v, L); };
                    BigItem it;
                    it.view = (uint32_t)from((int32_t)ri.view);
                    it.id = (uint32_t)from((int32_t)id);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        it.X[k] = from(r.s[k].X);
                        it.Y[k] = from(r.s[k].Y);
                        it.z[k] = __int_as_float(from(__float_as_int(r.s[k].z)));
                    }
                    const int32_t jx0 = from(rx0), jy0 = from(ry0);
                    const uint32_t jw = (uint32_t)from((int32_t)rw), n = (uint32_t)from((int32_t)n_regions);
                    uint32_t base = 0;
                    if ((int)lane == L) base = atomicAdd(&P.counters[1], n);
                    base = (uint32_t)from((int32_t)base);
                    if (base >= P.big_cap || n > P.big_cap - base) {      // no room: neutralise the part of the reservation inside the queue; the owner rasterises
                        if ((int)lane == L) { atomicOr(&P.counters[2], kStatusBigOverflow); in_lane = true; }
                        it.id = kNoTri;
                        it.region = 0;
                        for (uint32_t k = mine; k < n && base + k < P.big_cap; k += n_act) P.big[base + k] = it;
                        continue;
                    }
                    const uint32_t magic = (1u << 24) / jw + 1u;      // k / jw = k * magic >> 24, exact while k * jw < 2^24 (jw <= 256, k < 2^16)
                    for (uint32_t k = mine; k < n; k += n_act) {
                        const uint32_t q = n < 65536u ? (uint32_t)(((uint64_t)k * magic) >> 24) : k / jw;
                        const int32_t ry = jy0 + (int32_t)q, rx = jx0 + (int32_t)(k - q * jw);
                        it.region = ((uint32_t)ry << 16) | (uint32_t)rx;
                        if (TOPO_CHK(P.counters, base + k < P.big_cap && rx >= 0 && ry >= 0 && rx * 64 < P.W && ry * 64 < P.H, 7u, base + k)) P.big[base + k] = it;
                    }
                }
            }
            if (in_lane) raster_box(ts, vis, P.W, id, ts.px0, ts.px1, ts.py0, ts.py1);
        }
    }
}

__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v); }
__device__ __forceinline__ float unif(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int32_t)__float_as_uint(v))); }

// One wave per BigItem: the item carries the snapped vertices, so every lane re-runs the exact integer setup
// (wave-uniform: the item's fields are forced into scalar registers) and the wave sweeps the part of the triangle's
// pixel box inside the item's 64x64 px region.  Triangles spanning < 64 px (all that k_raster enqueues) take the int32
// form of the same integers (big_medium_lane), the giants that come through k_raster_rare the int64 form
// (big_giant_lane); both are in topo_pipeline.h and run lane by lane on the CPU in the tests.  Fragments are issued
// blind (no depth pre-test, see vis_min): only entries that carry a fragment (key != kVisClear) are dereferenced.
__global__ __launch_bounds__(256) void k_raster_big(FrameParams P) {
    // Segment marks: an item stays inside one 64 x 64 px region, i.e. inside 64 pixel rows of one or two 64-key segments
    // each.  Instead of one mark store beside every atomic instruction (half of this kernel's memory instructions), the
    // lanes note the rows they hit in LDS and lane r marks row r's segment(s) once per item.
    __shared__ uint8_t s_rows[4][64];
    uint32_t count = P.counters[1];
    if (count > P.big_cap) count = P.big_cap;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s_rows[wave][lane] = 0;
    const uint32_t wave_global = blockIdx.x * 4 + wave, wave_count = gridDim.x * 4;
    for (uint32_t item = P.counters[6] + wave_global; item < count; item += wave_count) {
        // The item is the same for the whole wave, but the compiler cannot use scalar loads for it (the queue is
        // written by other kernels through the same pointer type): say so field by field, and the integer setup
        // runs on the scalar unit instead of 64 times over on the vector one.
        BigItem bi;
        {
            const BigItem& g = P.big[item];
            bi.view = uni(g.view); bi.id = uni(g.id); bi.region = uni(g.region);
#pragma unroll
            for (int k = 0; k < 3; ++k) { bi.X[k] = uni(g.X[k]); bi.Y[k] = uni(g.Y[k]); bi.z[k] = unif(g.z[k]); }
        }
        if (bi.id == kNoTri) continue;
        const Vis vis = view_vis(P, bi.view);
        const int32_t rx = (int32_t)(bi.region & 0xFFFFu), ry = (int32_t)(bi.region >> 16);
        if (!TOPO_CHK(P.counters, bi.view < P.n_views && rx * 64 < P.W && ry * 64 < P.H, 11u, bi.region)) continue;
        uint8_t* const rows = s_rows[wave];
        if (spans_fit_int32(bi.X[0], bi.Y[0], bi.X[1], bi.Y[1], bi.X[2], bi.Y[2])) {
            big_medium_lane(bi.X, bi.Y, bi.z, bi.id, P.W, P.H, rx, ry, lane, [&](const uint32_t pix[4], const uint64_t key[4], const int32_t py[4]) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (key[k] != kVisClear) {
                        vis_min_unmarked(vis, pix[k], key[k]);
                        rows[py[k] & 63] = 1;
                    }
            });
        } else {
            big_giant_lane(bi.X, bi.Y, bi.z, bi.id, P.W, P.H, rx, ry, lane, [&](size_t pix, uint64_t key, int32_t py) {
                vis_min_unmarked(vis, pix, key);
                rows[py & 63] = 1;
            });
        }
        // (LDS operations of one wave complete in order: the notes above are visible to the reads below)
        if (rows[lane]) {
            rows[lane] = 0;
            const int32_t y = ry * 64 + (int32_t)lane, x0 = rx * 64, x1 = min(rx * 64 + 63, P.W - 1);
            const size_t first = (size_t)(vis.p - vis.base) + (size_t)y * P.W;
            const size_t s0 = (first + x0) >> 6, s1 = (first + x1) >> 6;      // a region row lies in one segment, two when W or the view origin is not a multiple of 64
            if (TOPO_CHK(P.counters, y < P.H && s1 < (((size_t)P.n_views * P.W * P.H + 63) >> 6), 14u, s1)) {
                vis.dirty[s0] = 1;
                if (s1 != s0) vis.dirty[s1] = 1;
            }
        }
    }
}

// ---- resolve: fs_main for the winner of every pixel, then the post pass --------------------------------
// A workgroup takes 64 x 16 px blocks; wave w shades rows 4w .. 4w+3 of each (a 64 x 4 px strip) on its own: its own halo,
// depth tile and record table, no barrier once the tables are staged (see k_resolve).
//
// The grid is PERSISTENT (four times the resident workgroups) and each workgroup walks its blocks with a static stride,
// software-pipelined: a strip's shading needs two dependent trips to memory before it can start -- the segment marks that
// say whether anything was drawn there (about half of a panorama is sky: such a strip is written out as constants without
// reading a key; with every tap at depth 1 the contour term is exactly 0 and the post pass returns the cleared texel
// unchanged), then the visibility keys -- and at ~1.5 us per trip under load those two waits were three quarters of a
// block's 12 us in a one-block-per-workgroup kernel (measured: with ALL arithmetic removed it still took 0.38 of its
// 0.50 ms).  So the marks of up to 64 strips are read in one go (lane j: strip j), and while strip i is shaded the keys of
// strip i+1 are already on their way into registers (each lane reads the keys of its own four pixels, 512-byte row
// segments, and three depth words of the 140-entry ring around the strip); the sRGB tables are staged into LDS once per
// workgroup.
//
// Winners are shared: the near field consists of triangles tens to thousands of pixels large, and two thirds of a
// pixel's arithmetic (three vs_main, the perspective divides, the doubled area: resolve_setup) depends on the winning
// triangle alone.  Each wave therefore lists the distinct winners of its 256 pixels -- a lane starts a new entry when
// its id differs from its left neighbour's -- and computes their records densely, one triangle per lane, into a
// per-wave LDS table; the pixels then finish from the record (resolve_pixel: the values of the one-step resolve_varyings,
// bit for bit).  The table holds 32 records (at c4 a wave's 256 pixels share 6.7 winners on average); a wave that meets
// more takes its rows in groups that fit, and a single row with more than that (far field: a triangle or less per
// pixel) is shaded in one step per pixel.
//
// Round 3: the kernel is bound by instruction issue (vector AND scalar instructions take the SIMD's one issue slot), so
// the row loop carries no per-row selects any more: what a row needs of the listing pass -- each pixel's record slot, or
// its winner id where the row is shaded in one step -- waits in LDS (s_id, s_slot), the output pointers advance by the pitch,
// the colour format is a template parameter, a record's kind-specific part is affine in the pixel (TriRecord), the
// positions of the ring entries are lane constants, and a row none of whose pixels can have a non-zero contour factor
// (decided by one comparison per pixel that can only err towards the long route) skips the post pass's divisions.
constexpr int kRPW = TOPO_RESOLVE_RPW;             // pixel rows per wave
constexpr int kResolveRows = 4 * kRPW;
static_assert(kResolveRows == (int)kResolveBlockH && kResolveBlockW == 64u, "the host sizes k_resolve's block grid from these");
static_assert(kRPW == 4 || kRPW == 8, "RowN below names the rows of a wave");
#ifndef TOPO_RESOLVE_RECS
#define TOPO_RESOLVE_RECS 20      // (20 records + 5 workgroups per CU beat 32 + 4: the table's 4.6 KB are what the fifth workgroup's LDS needs)
#endif
constexpr uint32_t kRecCap = TOPO_RESOLVE_RECS;    // triangle records per wave
#ifndef TOPO_RESOLVE_WGS
#define TOPO_RESOLVE_WGS 5
#endif
// One value per row of a wave.  Named members, not an array: an array indexed by a loop variable goes to scratch memory.
template <typename T>
struct RowN {
    T a, b, c, d, e, f, g, h;
};
#if TOPO_RESOLVE_RPW == 8
#define TOPO_ROWS(X) X(0, a) X(1, b) X(2, c) X(3, d) X(4, e) X(5, f) X(6, g) X(7, h)
#else
#define TOPO_ROWS(X) X(0, a) X(1, b) X(2, c) X(3, d)
#endif

struct ResolveBlock {          // wave-uniform description of one 64 x (4 kRPW) block
    uint32_t view;
    int32_t bx, by;            // pixel origin
};
__device__ __forceinline__ ResolveBlock resolve_block(const FrameParams& P, uint32_t b) {
    const uint32_t view = P.rblocks_view > 1u ? fastdiv(b, P.div_rblocks_view) : b, in_view = b - view * P.rblocks_view;      // (fastdiv needs a divisor >= 2)
    const uint32_t row = P.rblocks_x > 1u ? fastdiv(in_view, P.div_rblocks_x) : in_view;
    return ResolveBlock{view, (int32_t)(in_view - row * P.rblocks_x) * 64, (int32_t)row * kResolveRows};
}
// Did anything write a key of wave `wave`'s strip (rows kRPW wave .. kRPW wave + kRPW - 1 of the block) or its halo?  Every row
// of strip + halo spans at most three 64-key segments; t < kStripMarks names one (row, segment) mark.
constexpr uint32_t kStripMarks = (kRPW + 2) * 3;
__device__ __forceinline__ bool resolve_strip_marked(const FrameParams& P, const ResolveBlock& B, uint32_t wave, uint32_t t) {
    const int32_t row = (int32_t)t / 3, k = (int32_t)t - row * 3;
    int32_t y = B.by + kRPW * (int32_t)wave + row - 1;
    y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
    const int32_t x0 = B.bx > 0 ? B.bx - 1 : 0, x1 = B.bx + 64 < P.W ? B.bx + 64 : P.W - 1;
    const size_t first = (size_t)B.view * P.W * P.H + (size_t)y * P.W;
    const size_t seg = ((first + x0) >> 6) + k;
    const size_t last = (first + x1) >> 6;
    const size_t at = seg <= last ? seg : last;      // (always a load, of a mark of this row: no branch around it)
    const bool mark = TOPO_CHK(P.counters, at < (((size_t)P.n_views * P.W * P.H + 63) >> 6), 12u, at) ? P.dirty[at] != 0 : false;
    return seg <= last && mark;
}
// What a lane holds of a strip: the keys of its own kRPW pixels and up to three depths of the ring around the strip:
// ring0 = the pixel above the lane's column (row -1), ring1 = the pixel below it (row kRPW), ring2 (lanes 0 .. 2 kRPW + 3) =
// columns -1 and 64 of rows -1 .. kRPW (lane = 2 (row + 1) + side).
struct ResolveKeys {
    RowN<uint32_t> id, raw;
    uint32_t ring0, ring1, ring2;
};
constexpr uint32_t kRing2Lanes = 2 * (kRPW + 2);
__device__ __forceinline__ void resolve_load_keys(const FrameParams& P, const ResolveBlock& B, uint32_t lane, uint32_t wave, ResolveKeys& K) {
    const uint64_t* vis = P.vis + (size_t)B.view * P.W * P.H;
    const int32_t px = B.bx + (int32_t)lane, sy = B.by + kRPW * (int32_t)wave;
    // (outside the target the positions clamp to the edge -- the depth sampler is clamp-to-edge (texture.rs:113-117) --; lanes /
    // rows beyond the target only feed the contour taps' LDS tile)
    const int32_t cx = px > P.W - 1 ? P.W - 1 : px;
    const int32_t ym = sy > 0 ? sy - 1 : 0;            // the row above the strip
    auto row_of = [&](int32_t y) { return y > P.H - 1 ? P.H - 1 : y; };      // (wave-uniform)
    const uint64_t* col = vis + cx;
#define TOPO_X(r, m)                                                      \
    {                                                                     \
        const uint64_t key = col[(size_t)row_of(sy + r) * P.W];           \
        K.id.m = (uint32_t)key;                                           \
        K.raw.m = (uint32_t)(key >> 32);                                  \
    }
    TOPO_ROWS(TOPO_X)
#undef TOPO_X
    // the ring: depth words only; every lane loads three (clamped positions: no branches around the loads)
    K.ring0 = reinterpret_cast<const uint32_t*>(col + (size_t)ym * P.W)[1];
    K.ring1 = reinterpret_cast<const uint32_t*>(col + (size_t)row_of(sy + kRPW) * P.W)[1];
    {
        const int32_t e = (int32_t)(lane < kRing2Lanes ? lane : kRing2Lanes - 1u);
        int32_t x = (e & 1) ? B.bx + 64 : B.bx - 1, y = sy + (e >> 1) - 1;
        x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
        y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
        K.ring2 = reinterpret_cast<const uint32_t*>(vis + (size_t)y * P.W + x)[1];
    }
}
// the winner id of one pixel again (rows shaded in one step per pixel, later record groups: both rare)
__device__ __forceinline__ uint32_t resolve_reload_id(const FrameParams& P, const ResolveBlock& B, int32_t px, int32_t py) {
    const int32_t cx = px > P.W - 1 ? P.W - 1 : px, cy = py > P.H - 1 ? P.H - 1 : py;
    return (uint32_t)P.vis[(size_t)B.view * P.W * P.H + (size_t)cy * P.W + cx];
}

template <bool kBgra>
__device__ __forceinline__ uint32_t surface_order(uint32_t c) {      // Rgba -> the surface's channel order
    return kBgra ? (c & 0xFF00FF00u) | ((c >> 16) & 0xFFu) | ((c & 0xFFu) << 16) : c;
}
template <bool kBgra>
__device__ __forceinline__ void resolve_fill_sky(const FrameParams& P, const OutputParams& O, const ResolveBlock& B, uint32_t lane, uint32_t wave) {
    const int32_t px = B.bx + (int32_t)lane;
    if (px >= P.W) return;
    const int32_t y0 = B.by + kRPW * (int32_t)wave;
    uint8_t* rgba = O.rgba + (size_t)B.view * O.rgba_view_stride + (size_t)y0 * O.rgba_pitch + (size_t)px * 4;
    uint8_t* depth = O.depth ? reinterpret_cast<uint8_t*>(O.depth) + (size_t)B.view * O.depth_view_stride + (size_t)y0 * O.depth_pitch + (size_t)px * 4 : nullptr;
    const uint32_t sky = surface_order<kBgra>(P.sky_c8);
    for (int32_t r = 0; r < kRPW && y0 + r < P.H; ++r, rgba += O.rgba_pitch) {
        *reinterpret_cast<uint32_t*>(rgba) = sky;
        if (depth) { *reinterpret_cast<float*>(depth) = 1.0f; depth += O.depth_pitch; }
    }
}
__device__ __forceinline__ uint32_t pop_bit(uint64_t& m) {      // wave-uniform mask: scalar instructions
    const uint32_t j = (uint32_t)__builtin_ctzll(m);
    m &= m - 1ull;
    return j;
}
// A wave's LDS tables are written and read by that wave alone, and a wave's LDS operations complete in order; what the
// hardware does not promise is that the COMPILER keeps a lane's read behind another lane's write to a different address.
// This fence (no instruction: it only orders the compiler's memory operations within the wave) stands between every write
// phase and the read phase that follows it.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The four waves of a workgroup share the tables and the list of blocks, and nothing else: wave w takes rows kRPW w .. of every
// block (a 64 x kRPW strip) with its own halo, its own depth tile and its own record table, at its own pace -- no barrier after
// the tables are in place.  (With one depth tile per block, two barriers per block made every wave wait for the block's
// slowest: 29 % of all wave time.)
// kSrgb: the targets are *Srgb formats (encode on store, decode on sample); otherwise plain unorm8.  kBgra: channel order.
template <bool kSrgb, bool kBgra>
__global__ __launch_bounds__(256, TOPO_RESOLVE_WGS) void k_resolve(FrameParams P, OutputParams O) {
    __shared__ float s_thresh[258];    // sRGB code boundaries; [255..257] = NaN: never <= anything (srgb_encode_lut probes up to 256)
    __shared__ float s_decode[256];
    __shared__ float s_ndec[256];      // normal channel decode 2c/255 - 1
    __shared__ uint32_t s_lut[1024];   // 4096 one-byte bins of srgb_encode_lut
    __shared__ float s_lin[4][kRPW + 2][66];                 // per wave: linear depth of the strip + halo
#ifdef TOPO_EXP_REC_WORDMAJOR      // experiment build: the round-2 layout (word-major: 34 + 17 LDS instructions per record written / read)
    __shared__ uint32_t s_rec[4][kTriRecordWords][kRecCap];
#define TOPO_REC_AT(wv, slot, word) s_rec[wv][word][slot]
#else
    // per wave: the records, record-major at a stride of 36 words (16-byte aligned): a record is written and read as nine 16-byte
    // LDS operations instead of 34 / 17 four- and eight-byte ones
    constexpr int kRecStride = (kTriRecordWords + 3) & ~3;
    __shared__ __attribute__((aligned(16))) uint32_t s_rec[4][kRecCap][kRecStride];
#define TOPO_REC_AT(wv, slot, word) s_rec[wv][slot][word]
#endif
    __shared__ uint32_t s_uid[4][kRecCap];                   // per wave: the distinct winner ids of a group of rows
    __shared__ uint8_t s_slot[4][kRPW][64];                  // per wave and pixel: the number of its entry among the strip's table entries (0xFF: none)
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t tx = (int32_t)lane;
    // Blocks are dealt out with a static stride: workgroup g takes blocks g, g + grid, g + 2 grid, ... -- a sample of every
    // part of every view, so each workgroup gets the same mix of sky, far field and near field.  (Handing out runs of
    // consecutive blocks dynamically measured 10 % to 3.5x slower: a run is all sky or all near field, and a block takes
    // ~10 us from first mark to last store, so whoever draws the last near-field run finishes long after everyone else.)
    const uint32_t n_blocks = P.rblock_count, stride = gridDim.x;               // blocks P.rblock_first .. of the submission's rblocks_view * n_views
    const uint32_t per_wg = (n_blocks - blockIdx.x + stride - 1) / stride;      // blocks blockIdx.x + j * stride, j < per_wg (the grid is <= n_blocks)
    // once per workgroup: the tables
    s_thresh[threadIdx.x] = threadIdx.x < 255 ? bits_f(TOPO_SRGB_THRESH_BITS[threadIdx.x]) : NAN;
    if (threadIdx.x < 2) s_thresh[256 + threadIdx.x] = NAN;
    s_decode[threadIdx.x] = bits_f(TOPO_SRGB_DECODE_BITS[threadIdx.x]);
    s_ndec[threadIdx.x] = normal_channel(threadIdx.x);
#pragma unroll
    for (int k = 0; k < 4; ++k) s_lut[threadIdx.x + 256 * k] = TOPO_SRGB_LUT12_WORDS[threadIdx.x + 256 * k];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut);
    // the frame's counters (queue fills, status bits) for whoever waits for the frame: final since the last raster kernel, stored to
    // the host's pinned ring from here (a copy operation behind the frame was a blit kernel of its own: ~10 us of every frame)
    if (P.status_out && blockIdx.x == 0 && threadIdx.x < 16) P.status_out[threadIdx.x] = P.counters[threadIdx.x];
    __syncthreads();                   // the only barrier
#ifdef TOPO_RESOLVE_PROF      // experiment build: where do a wave's cycles go?  counters[8..15], units of 1024 cycles summed over waves
    uint32_t pf_t = (uint32_t)__builtin_amdgcn_s_memtime(), pf_acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t pf_start = pf_t;
#define TOPO_PROF(slot) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); pf_acc[slot] += now_ - pf_t; pf_t = now_; }
#else
#define TOPO_PROF(slot)
#endif
    float (*const lin_tile)[66] = s_lin[wave];
    uint8_t (*const slot_tile)[64] = s_slot[wave];
    const int32_t sy0 = kRPW * (int32_t)wave;      // the strip's first row within its block
    // lane constants: the pixel's column as a double (TriRecord kind 1), the lane's entry of the ring's side columns
    const double lane_d = (double)tx;
    const float two_over_w = div_f(2.0f, (float)P.W), two_over_h = div_f(2.0f, (float)P.H);
    const int32_t ring2_e = (int32_t)(lane < kRing2Lanes ? lane : kRing2Lanes - 1u);
    float* const ring2_at = &lin_tile[ring2_e >> 1][(ring2_e & 1) ? 65 : 0];

    for (uint32_t j0 = 0; j0 < per_wg; j0 += 64) {
        const uint32_t nj = per_wg - j0 < 64u ? per_wg - j0 : 64u;
        // ---- which of these blocks' strips hold anything: lane j looks at block j0 + j, all its marks in one trip to memory
        uint64_t mm, mc;       // strips with / without anything drawn
        {
            const ResolveBlock Bl = resolve_block(P, P.rblock_first + blockIdx.x + (j0 + (lane < nj ? lane : 0u)) * stride);
            bool any = false;
#pragma unroll
            for (uint32_t t = 0; t < kStripMarks; ++t) any |= resolve_strip_marked(P, Bl, wave, t);      // (unconditional loads, none chained to another)
            const bool exists = lane < nj && Bl.by + sy0 < P.H;
            mm = __ballot(exists && any);
            mc = __ballot(exists && !any);
        }
        const uint32_t n_marked = (uint32_t)__popcll(mm), n_clear = (uint32_t)__popcll(mc);
        TOPO_PROF(0)      // marks
        auto block_of = [&](uint32_t j) { return resolve_block(P, P.rblock_first + blockIdx.x + (j0 + j) * stride); };      // (j comes out of a wave-uniform mask)
        // the untouched strips are pure stores: spread over the marked strips' iterations, so that their bandwidth hides
        // under the shading
        const uint32_t fills_per_iter = n_marked ? (n_clear + n_marked - 1) / n_marked : n_clear;
        // The keys of strip i + 1 are requested once strip i's keys have been consumed (depths into the LDS tile and the depth
        // output, ids into entry numbers): they travel under strip i's record pass and shading -- the bulk of a strip's time --
        // in the registers strip i's keys have just left.
        ResolveKeys K;
        uint32_t j_cur = 0;
        bool have = mm != 0ull;
#ifndef TOPO_EXP_KEY_PREFETCH      // every strip's keys are requested when the strip starts, none ahead (TOPO_EXP_KEY_PREFETCH: the round-2 form, see below)
        if (have) j_cur = pop_bit(mm);
        while (have) {
            resolve_load_keys(P, block_of(j_cur), lane, wave, K);
            // (this strip's share of the untouched strips -- pure stores -- goes out under the keys' trip to memory)
            for (uint32_t f = 0; f < fills_per_iter && mc; ++f) resolve_fill_sky<kBgra>(P, O, block_of(pop_bit(mc)), lane, wave);
#else
        if (have) { j_cur = pop_bit(mm); resolve_load_keys(P, block_of(j_cur), lane, wave, K); }
        while (have) {
#endif
#ifdef TOPO_RESOLVE_EARLY_PREFETCH      // experiment build: the next strip's keys requested at the top of the iteration, into registers of their own
            ResolveKeys Kn;
            const bool more = mm != 0ull;
            uint32_t j_next = 0;
            if (more) { j_next = pop_bit(mm); resolve_load_keys(P, block_of(j_next), lane, wave, Kn); }
#endif
            const ResolveBlock B = block_of(j_cur);
            const int32_t px = B.bx + tx, y0 = B.by + sy0;
            const bool in_x = px < P.W;        // lanes beyond the target's right edge stay: they compute triangle records
            const int32_t n_rows = P.H - y0 < kRPW ? P.H - y0 : kRPW;      // rows of the strip inside the target (>= 1)
            bool terrain = K.ring0 != 0x3F800000u || K.ring1 != 0x3F800000u || (lane < kRing2Lanes && K.ring2 != 0x3F800000u);
#define TOPO_X(r, m) terrain |= K.raw.m != 0x3F800000u;
            TOPO_ROWS(TOPO_X)
#undef TOPO_X
            TOPO_PROF(2)  // wait for this strip's keys
            const bool any_terrain = __ballot(terrain) != 0ull;
            RowN<uint32_t> n_row;
            uint32_t n_all = 0;            // table entries of the strip
            if (any_terrain) {
                wave_lds_fence();              // (the previous strip's reads of the tiles are done)
#define TOPO_X(r, m) lin_tile[r + 1][tx + 1] = linear_depth(bits_f(K.raw.m));
                TOPO_ROWS(TOPO_X)
#undef TOPO_X
                lin_tile[0][tx + 1] = linear_depth(bits_f(K.ring0));
                lin_tile[kRPW + 1][tx + 1] = linear_depth(bits_f(K.ring1));
                {
                    const float l2 = linear_depth(bits_f(K.ring2));
                    if (lane < kRing2Lanes) *ring2_at = l2;
                }
                TOPO_PROF(4)  // linear depths
                // the depth output is the key's depth word
                if (in_x && O.depth) {
                    uint8_t* dp = reinterpret_cast<uint8_t*>(O.depth) + (size_t)B.view * O.depth_view_stride + (size_t)y0 * O.depth_pitch + (size_t)px * 4;
#define TOPO_X(r, m) if (r < n_rows) { *reinterpret_cast<uint32_t*>(dp) = K.raw.m; dp += O.depth_pitch; }
                    TOPO_ROWS(TOPO_X)
#undef TOPO_X
                }
                // ---- the distinct winners of this wave's pixels: a lane opens an entry where its id differs from its left
                // neighbour's.  Entries are numbered over the strip's rows that are shaded from the table (rows with at most kRecCap
                // entries, while the numbers fit a byte); n_row = entries of a row.  A pixel's entry number waits in LDS.
#define TOPO_X(r, m)                                                                                                            \
    {                                                                                                                           \
        const bool valid = in_x && r < n_rows && K.id.m != kNoTri;                                                              \
        const uint32_t left = (uint32_t)__shfl_up((int)K.id.m, 1);                                                              \
        const bool leader = valid && (lane == 0 || K.id.m != left);                                                             \
        const uint64_t mask = __ballot(leader);                                                                                 \
        n_row.m = (uint32_t)__popcll(mask);                                                                                     \
        if (n_all + n_row.m > 254u) n_row.m = kRecCap + 1u; /* (entry numbers are bytes: such a row is shaded in one step per pixel) */ \
        const uint32_t slot = n_all + (uint32_t)__popcll(mask & ((2ull << lane) - 1ull)) - 1u; /* valid lanes: the last leader at or before them */ \
        slot_tile[r][tx] = (uint8_t)(valid && n_row.m <= kRecCap ? slot : 0xFFu);                                               \
        if (leader && n_row.m <= kRecCap && slot < kRecCap) s_uid[wave][slot] = K.id.m; /* the first group's ids (later groups: below) */ \
        n_all += n_row.m <= kRecCap ? n_row.m : 0u;                                                                             \
    }
                TOPO_ROWS(TOPO_X)
#undef TOPO_X
                wave_lds_fence();
            }
#ifndef TOPO_RESOLVE_EARLY_PREFETCH
            // ---- this strip's keys are consumed: request the next strip's
            const bool more = mm != 0ull;
            uint32_t j_next = 0;
            // (Rounds 2 and 3 requested the NEXT strip's keys here, to travel under the record pass and the rows.  But the wait counter
            // is in order: the record pass below waits for its own loads -- cache hits -- behind that request's trip to HBM, so what
            // the request hid of the trip at the next strip's start it cost here: with no request ahead at all the kernel took the
            // same 0.371 ms -- with 106 registers instead of 125, which is what lets a fifth workgroup onto the CU (0.362 ms).
            // Requesting BEHIND the record pass would hide the trip under the rows; every form of it tried -- the request inside the
            // group loop, the first record pass peeled in front of the loop, its loads and its arithmetic as two calls with the request
            // between them -- spilled 12 to 35 registers and lost.)
#ifndef TOPO_EXP_KEY_PREFETCH
            if (more) j_next = pop_bit(mm);
#else
            if (more) { j_next = pop_bit(mm); resolve_load_keys(P, block_of(j_next), lane, wave, K); }
#endif
#endif
#ifdef TOPO_EXP_KEY_PREFETCH
            for (uint32_t f = 0; f < fills_per_iter && mc; ++f) resolve_fill_sky<kBgra>(P, O, block_of(pop_bit(mc)), lane, wave);
#endif
            TOPO_PROF(1)  // issue of the next keys + sky fills
            if (!any_terrain) {                // marked, but every key still cleared (a mark covers 64 keys): the cleared texel and depth 1
                resolve_fill_sky<kBgra>(P, O, B, lane, wave);
                have = more;
                j_cur = j_next;
#ifdef TOPO_RESOLVE_EARLY_PREFETCH
                K = Kn;
#endif
                continue;
            }
            uint8_t* rgba_p = O.rgba + (size_t)B.view * O.rgba_view_stride + (size_t)y0 * O.rgba_pitch + (size_t)px * 4;
            const ViewDev& view = P.views[B.view];
            // what fs_main reads of the view, once per strip and wave-uniform: left to the compiler these are re-loaded in every
            // row (it cannot prove the output stores do not alias them) behind an s_waitcnt vmcnt(0) that also waits for the
            // previous row's stores to land
            // (over the scalar data path -- constant address space --: as vector loads they were waited for with s_waitcnt vmcnt(0)
            // right behind the request for the next strip's keys, i.e. every strip began by sitting out that request's trip to HBM)
            const auto view_c = (const __attribute__((address_space(4))) ViewDev*)(const void*)(P.views + B.view);
            const f3 sun = {view_c->sun[0], view_c->sun[1], view_c->sun[2]};
            const float cam_x = view_c->cam_x, cam_y = view_c->cam_y;
            const int32_t view_mode = view_c->view_mode;
#ifdef TOPO_RESOLVE_STATS      // experiment build: how well do winners share?  counters[12] entries, [13] waves with terrain, [14] groups, [15] terrain pixels
            {
                uint32_t npx = 0;
                for (int32_t r = 0; r < n_rows; ++r) npx += (uint32_t)__popcll(__ballot(in_x && slot_tile[r][tx] != 0xFFu));
                if (lane == 0 && n_all) { atomicAdd(&P.counters[12], n_all); atomicAdd(&P.counters[13], 1u); atomicAdd(&P.counters[15], npx); }
            }
#endif
            const float gx = pixel_gx(px, two_over_w);
            // Rows are taken in groups of consecutive rows whose table entries fit the table (near field: all rows in one
            // group, a handful of records); a row with more entries than the table holds is a group of its own, shaded in
            // one step per pixel (resolve_varyings), as every row was in round 1.
            int32_t r0 = 0;
            uint32_t gbase = 0;            // table entries of the groups before this one
            bool one_group = n_all <= kRecCap;
#define TOPO_X(r, m) one_group = one_group && n_row.m <= kRecCap;
            TOPO_ROWS(TOPO_X)
#undef TOPO_X
#pragma unroll 1
            while (r0 < n_rows) {
                int32_t r1;
                uint32_t cnt;
                bool table = true;
                if (one_group) {
                    r1 = n_rows;
                    cnt = n_all;
                } else {
                    r1 = r0;
                    cnt = 0;
#define TOPO_X(r, m) if (r >= r0 && r == r1 && n_row.m <= kRecCap && cnt + n_row.m <= kRecCap) { cnt += n_row.m; r1 = r + 1; }
                    TOPO_ROWS(TOPO_X)
#undef TOPO_X
                    if (r1 > n_rows) r1 = n_rows;
                    if (r1 == r0) {            // the row at r0 alone exceeds the table
                        table = false;
                        r1 = r0 + 1;
                    } else if (gbase != 0u) {
                        // a later group: its ids were not listed above (their entry numbers lie beyond the table): listed now -- a
                        // lane is the leader of its entry iff its left neighbour has another one; the id is read again
                        for (int32_t r = r0; r < r1; ++r) {
                            const uint32_t e = slot_tile[r][tx], el = (uint32_t)__shfl_up((int)e, 1);
                            if (e != 0xFFu && (lane == 0 || e != el) && TOPO_CHK(P.counters, e - gbase < kRecCap, 15u, e)) s_uid[wave][e - gbase] = resolve_reload_id(P, B, px, y0 + r);
                        }
                    }
                }
#ifdef TOPO_RESOLVE_STATS
                if (lane == 0 && cnt) atomicAdd(&P.counters[14], 1u);
#endif
                if (table && cnt) {            // one triangle per lane: everything that depends on the triangle alone
                    wave_lds_fence();
                    if (lane < cnt) {
                        const uint32_t id = s_uid[wave][lane];
                        const uint32_t draw = id >> 1, fan = id & 1u;
                        const uint32_t rank = fastdiv(draw, P.div_tris), tri = draw - rank * P.tris_per_tile;
                        TriRecord rec;
                        if (TOPO_CHK(P.counters, rank < P.n_tiles, 13u, id)) resolve_setup<true>(P.tiles[rank], P.tile_w, P.div_hm1, P.tile_h - 1, view, P.W, P.H, tri, fan, s_ndec, B.bx, y0, rec);
                        else rec = TriRecord{};
                        int k = 0;
#define TOPO_X(f) TOPO_REC_AT(wave, lane, k++) = rec.f;
                        TOPO_TRIREC_WORDS(TOPO_X)
#undef TOPO_X
                    }
                    wave_lds_fence();
                }
                TOPO_PROF(5)  // winners: entries, records (gathers)
                // One row of the group.  The two ways a row gets its varyings -- from the wave's record table, or in one step per pixel
                // with memory loads of its own (rows with more winners than the table holds: 0.5 % at c4) -- are two INSTANCES of
                // this body, each in a loop of its own: in one loop the compiler had to assume the memory loads of the second form
                // pending in the first as well, and every table row began by waiting for the previous row's output store to land
                // (s_waitcnt vmcnt(0)).
                auto shade_row = [&](auto table_tag, int32_t r) __attribute__((always_inline)) {
                    constexpr bool kTable = decltype(table_tag)::value;
                        const int32_t py = y0 + r;
                        // the pixel's entry number (rows shaded from the table) or its winner id
                        const uint32_t sel = kTable ? (uint32_t)slot_tile[r][tx] : (in_x ? resolve_reload_id(P, B, px, py) : kNoTri);
                        // the contour taps first: they depend on nothing, so their LDS trip overlaps the record's
                        float ln[8];
                        {
                            int k = 0;
    #pragma unroll
                            for (int i = -1; i <= 1; ++i)
    #pragma unroll
                                for (int j = -1; j <= 1; ++j) {
                                    if (i == 0 && j == 0) continue;
                                    ln[k++] = lin_tile[r + 1 + j][tx + 1 + i];
                                }
                        }
                        const float lin_c = lin_tile[r + 1][tx + 1];
                        // render target texel (Rgba8UnormSrgb): the cleared value or the shaded winner
                        uint32_t c8 = P.sky_c8;
                        if (sel != (kTable ? 0xFFu : kNoTri)) {
                            float lin[4] = {0.0f, 0.71f, 0.885f, 1.0f};
                            f3 wpos = {0.0f, 0.0f, 0.0f}, wnrm;
                            bool ok;
                            if (kTable) {
                                const uint32_t sl = sel - gbase;
                                TriRecord rec;
                                int k = 0;
    #define TOPO_X(f) rec.f = TOPO_REC_AT(wave, sl, k++);
                                TOPO_TRIREC_WORDS(TOPO_X)
    #undef TOPO_X
                                const PixelAt at = {px, py, lane_d, (double)r, gx, pixel_gy(py, two_over_h)};
                                ok = resolve_pixel(rec, at, wpos.x, wpos.y, wnrm);
    #ifdef TOPO_RESOLVE_STATS
                                { const uint32_t n3 = (uint32_t)__popcll(__ballot(rec.kind == 3u)); if (n3 && lane == (uint32_t)__builtin_ctzll(__ballot(true))) atomicAdd(&P.counters[11], n3); }
    #endif
                            } else {
                                const uint32_t draw = sel >> 1, fan = sel & 1u;
                                const uint32_t rank = fastdiv(draw, P.div_tris), tri = draw - rank * P.tris_per_tile;
                                ok = TOPO_CHK(P.counters, rank < P.n_tiles, 13u, sel) &&
                                     resolve_varyings<true>(P.tiles[rank], P.tile_w, P.div_hm1, P.tile_h - 1, view, P.W, P.H, tri, fan, s_ndec, px, py, wpos, wnrm);
                            }
                            if (ok) shade_fragment(view_mode, sun, cam_x, cam_y, (float)px + 0.5f, (float)py + 0.5f, wpos, wnrm, lin);
                            c8 = (kSrgb ? srgb_encode_lut3(s_thresh, lut, lin[0], lin[1], lin[2]) : to_unorm8(lin[0]) | (to_unorm8(lin[1]) << 8) | (to_unorm8(lin[2]) << 16)) |
                                 (to_unorm8(lin[3]) << 24);
                        }
                        // The post pass.  Its contour factor a is 0 iff RN(contour / centre) <= 0.05f; contour <= 0.0499f * centre
                        // (centre is a linear depth: 50 .. 5e5) puts the quotient below 0.04991: such a pixel returns its texel
                        // unchanged, and a row of them skips the divisions.  (A NaN fails the comparison and takes the long route.)
                        float contour = 8.0f * lin_c;
    #pragma unroll
                        for (int k = 0; k < 8; ++k) contour -= ln[k];
                        uint32_t out = c8;
                        const bool long_post = !P.post_off && __ballot(!(contour <= 0.0499f * lin_c)) != 0ull;      // (post_off: the render-target texel itself)
    #ifdef TOPO_RESOLVE_STATS
                        if (lane == 0) { atomicAdd(&P.counters[kTable ? 8 : 9], 1u); if (long_post) atomicAdd(&P.counters[10], 1u); }
    #endif
                        if (long_post) out = post_pixel_t<true>(s_thresh, s_decode, c8, lin_c, ln, lut, kSrgb);
                        if (in_x) *reinterpret_cast<uint32_t*>(rgba_p) = surface_order<kBgra>(out);
                };
                if (table) {
#pragma unroll 1
                    for (int32_t r = r0; r < r1; ++r, rgba_p += O.rgba_pitch) shade_row(std::true_type{}, r);
                } else {
#pragma unroll 1
                    for (int32_t r = r0; r < r1; ++r, rgba_p += O.rgba_pitch) shade_row(std::false_type{}, r);
                }
                TOPO_PROF(6)  // pixels
                gbase += table ? cnt : 0u;
                r0 = r1;
            }
            have = more;
            j_cur = j_next;
#ifdef TOPO_RESOLVE_EARLY_PREFETCH
            K = Kn;
#endif
        }
        while (mc) resolve_fill_sky<kBgra>(P, O, block_of(pop_bit(mc)), lane, wave);
        TOPO_PROF(1)
    }
#ifdef TOPO_RESOLVE_PROF
    if (lane == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&P.counters[9 + k], pf_acc[k] >> 10);
        atomicAdd(&P.counters[8], ((uint32_t)__builtin_amdgcn_s_memtime() - pf_start) >> 10);
    }
#endif
#undef TOPO_PROF
}

// The post pass with the pixelise branch on (postprocessing_shader.wgsl:70-74; never in the reference, which pins pixelize_n to
// 100): the colour is a sample of the render target AWAY from the pixel's own texel, so the frame takes two passes -- k_resolve
// stores the render-target texels (post_off), this kernel samples them (sample_pixelized), takes the contour from the depth
// image and stores the surface texel.  One lane per pixel; nothing here is tuned.
__global__ __launch_bounds__(256) void k_post_pixelize(int32_t W, int32_t H, float vw, float vh, float n, const uint8_t* __restrict__ pre, OutputParams O,
                                                       const float* __restrict__ depth, size_t depth_view_stride, size_t depth_pitch, uint32_t linear_target,
                                                       uint32_t bgra) {
    __shared__ float s_thresh[256], s_decode[256];
    s_thresh[threadIdx.x] = bits_f(TOPO_SRGB_THRESH_BITS[threadIdx.x]);
    s_decode[threadIdx.x] = bits_f(TOPO_SRGB_DECODE_BITS[threadIdx.x]);
    __syncthreads();
    const int32_t px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const uint32_t view = blockIdx.z;
    if (px >= W || py >= H) return;
    const uint8_t* img = pre + (size_t)view * W * H * 4;
    const uint8_t* dimg = reinterpret_cast<const uint8_t*>(depth) + (size_t)view * depth_view_stride;
    auto texel = [&](int32_t x, int32_t y, float out[4]) {
        const uint32_t c8 = *reinterpret_cast<const uint32_t*>(img + ((size_t)y * W + x) * 4);
        out[0] = linear_target ? from_unorm8(c8 & 255u) : s_decode[c8 & 255u];
        out[1] = linear_target ? from_unorm8((c8 >> 8) & 255u) : s_decode[(c8 >> 8) & 255u];
        out[2] = linear_target ? from_unorm8((c8 >> 16) & 255u) : s_decode[(c8 >> 16) & 255u];
        out[3] = from_unorm8(c8 >> 24);
    };
    float rc[4];
    sample_pixelized(px, py, vw, vh, n, W, H, texel, rc);
    auto lin_at = [&](int32_t x, int32_t y) {
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        return linear_depth(*reinterpret_cast<const float*>(dimg + (size_t)y * depth_pitch + (size_t)x * 4));
    };
    float ln[8];
    int k = 0;
#pragma unroll
    for (int i = -1; i <= 1; ++i)
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
            if (i == 0 && j == 0) continue;
            ln[k++] = lin_at(px + i, py + j);
        }
    uint32_t out = post_mix(s_thresh, rc, lin_at(px, py), ln, linear_target == 0u);
    if (bgra) out = (out & 0xFF00FF00u) | ((out >> 16) & 0xFFu) | ((out & 0xFFu) << 16);
    *reinterpret_cast<uint32_t*>(O.rgba + (size_t)view * O.rgba_view_stride + (size_t)py * O.rgba_pitch + (size_t)px * 4) = out;
}

// ---- overlay pass (line_shader.wgsl) ------------------------------------------------------------------
// Overlay geometry is a few hundred CPU-tessellated triangles: one lane per triangle walks its pixel box and raises the
// pixel's overlay key (depth bits << 32 | ~triangle index) with a 64-bit atomic MAX -- `Greater` plus "the earlier draw
// keeps an equal depth" -- over a key image that starts at the post quad's depth 1/4096; a second kernel colours the
// pixels whose key moved.
__global__ __launch_bounds__(64) void k_overlay_raster(const OverlayVertex* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t n_tris,
                                                       uint32_t n_verts, float width, int32_t W, int32_t H, uint64_t* __restrict__ keys) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_tris) return;
    const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
    if (i0 >= n_verts || i1 >= n_verts || i2 >= n_verts) return;      // (wgpu rejects such a draw; here the triangle is skipped)
    SVert s0, s1, s2;
    if (overlay_vertex(verts[i0], width, (float)W, (float)H, s0) != kVtxOk || overlay_vertex(verts[i1], width, (float)W, (float)H, s1) != kVtxOk ||
        overlay_vertex(verts[i2], width, (float)W, (float)H, s2) != kVtxOk)
        return;
    TriSetup ts;
    if (!triangle_setup(s0, s1, s2, W, H, ts)) return;
    for (int32_t py = ts.py0; py <= ts.py1; ++py)
        for (int32_t px = ts.px0; px <= ts.px1; ++px) {
            const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
            int64_t F[3];
            bool in = true;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                F[e] = ts.dy[e] * (cx - ts.ax[e]) - ts.dx[e] * (cy - ts.ay[e]);
                in = in && F[e] + ts.bias[e] >= 0;
            }
            if (!in) continue;
            const float z = fmaf((float)F[1] * ts.iA, ts.dz1, fmaf((float)F[2] * ts.iA, ts.dz2, ts.z0));
            if (!(z >= 0.0f && z <= 1.0f)) continue;      // clip volume 0 <= z <= w
            atomicMax(reinterpret_cast<unsigned long long*>(keys + (size_t)py * W + px), (unsigned long long)overlay_key(z, t));
        }
}

__global__ __launch_bounds__(256) void k_overlay_resolve(const OverlayVertex* __restrict__ verts, const uint32_t* __restrict__ idx, float width,
                                                         int32_t W, int32_t H, uint64_t* __restrict__ keys, uint8_t* __restrict__ rgba, size_t pitch,
                                                         uint32_t linear_target, uint32_t bgra) {
    const int32_t px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const uint64_t key = keys[(size_t)py * W + px];
    keys[(size_t)py * W + px] = kOverlayClear;      // ready for the next frame's overlay
    if (key <= kOverlayClear) return;                // nothing passed `Greater` here (an equal depth has a smaller low word)
    const uint32_t t = 0xFFFFFFFFu - (uint32_t)key;
    float rgb[3];
    if (!overlay_color(verts[idx[3 * t]], verts[idx[3 * t + 1]], verts[idx[3 * t + 2]], width, W, H, px, py, rgb)) return;
    uint32_t out;
    if (linear_target) {
        out = to_unorm8(rgb[0]) | (to_unorm8(rgb[1]) << 8) | (to_unorm8(rgb[2]) << 16);
    } else {
        float thresh[1];      // (the 8-probe search reads the table from constant memory: overlays are a few thousand pixels)
        (void)thresh;
        auto enc = [](float l) {
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = 128; step >= 1; step >>= 1)
                if (bits_f(TOPO_SRGB_THRESH_BITS[lo + step - 1]) <= l) lo += step;
            return lo;
        };
        out = enc(rgb[0]) | (enc(rgb[1]) << 8) | (enc(rgb[2]) << 16);
    }
    out |= to_unorm8(1.0f) << 24;
    if (bgra) out = (out & 0xFF00FF00u) | ((out >> 16) & 0xFFu) | ((out & 0xFFu) << 16);
    *reinterpret_cast<uint32_t*>(rgba + (size_t)py * pitch + (size_t)px * 4) = out;
}

// Text: one 64-thread workgroup per glyph quad.  Pass 1 raises the key of every pixel the quad covers (the lines' key image:
// depth bits << 32 | ~glyph index, 64-bit atomic max = Greater + "the earlier draw keeps an equal depth"); pass 2 lets the
// glyph that owns a pixel blend into it and puts the key back to the post quad's depth.
__global__ __launch_bounds__(64) void k_glyph_raster(const GlyphInstance* __restrict__ glyphs, uint32_t n_glyphs, float depth, int32_t W, int32_t H,
                                                     uint64_t* __restrict__ keys) {
    const uint32_t g = blockIdx.x;
    if (g >= n_glyphs) return;
    const GlyphInstance gi = glyphs[g];
    const uint32_t gw = gi.dim[0], n = gw * gi.dim[1];
    const unsigned long long key = (unsigned long long)overlay_key(depth, g);
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        const int64_t px = (int64_t)gi.pos[0] + (int64_t)(i % gw), py = (int64_t)gi.pos[1] + (int64_t)(i / gw);
        if (px < 0 || py < 0 || px >= W || py >= H) continue;
        atomicMax(reinterpret_cast<unsigned long long*>(keys + (size_t)py * W + (size_t)px), key);
    }
}

__global__ __launch_bounds__(64) void k_glyph_resolve(const GlyphInstance* __restrict__ glyphs, uint32_t n_glyphs, float depth, const uint8_t* __restrict__ atlas,
                                                      uint32_t aw, uint32_t ah, int32_t W, int32_t H, uint64_t* __restrict__ keys, uint8_t* __restrict__ rgba,
                                                      size_t pitch, uint32_t linear_target, uint32_t bgra) {
    __shared__ float s_thresh[256], s_decode[256];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        s_thresh[i] = bits_f(TOPO_SRGB_THRESH_BITS[i]);
        s_decode[i] = bits_f(TOPO_SRGB_DECODE_BITS[i]);
    }
    __syncthreads();
    const uint32_t g = blockIdx.x;
    if (g >= n_glyphs) return;
    const GlyphInstance gi = glyphs[g];
    const uint32_t gw = gi.dim[0], n = gw * gi.dim[1];
    const uint64_t mine = overlay_key(depth, g);
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        const uint32_t dx = i % gw, dy = i / gw;
        const int64_t px = (int64_t)gi.pos[0] + dx, py = (int64_t)gi.pos[1] + dy;
        if (px < 0 || py < 0 || px >= W || py >= H) continue;
        uint64_t* k = keys + (size_t)py * W + (size_t)px;
        if (*k != mine) continue;                 // another glyph (an earlier one) owns the pixel, or the depth never passed
        *k = kOverlayClear;                        // ready for the next overlay call
        const uint32_t ax = gi.uv[0] + dx, ay = gi.uv[1] + dy;
        const uint32_t mask = ax < aw && ay < ah ? atlas[(size_t)ay * aw + ax] : 0u;      // (outside the atlas: transparent)
        uint32_t* out = reinterpret_cast<uint32_t*>(rgba + (size_t)py * pitch + (size_t)px * 4);
        *out = glyph_blend(gi, mask, *out, linear_target == 0u, bgra != 0u, s_thresh, s_decode);
    }
}

__global__ __launch_bounds__(256) void k_overlay_init(uint64_t* __restrict__ keys, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = kOverlayClear;
}

// One lane per peak: project, one depth lookup, one comparison (render_engine.rs:338-396).
__global__ __launch_bounds__(256) void k_visible_peaks(const float* __restrict__ proj, uint32_t w, uint32_t h,
                                                       const float* __restrict__ depth, size_t depth_pitch, uint32_t n,
                                                       const float* __restrict__ peaks, uint8_t* __restrict__ visible,
                                                       uint32_t* __restrict__ xy) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x_pos = 0, y_pos = 0;
    float peak_dist = 0.0f;
    bool vis = false;
    if (project_peak(proj, peaks[3 * i], peaks[3 * i + 1], peaks[3 * i + 2], (float)w, (float)h, x_pos, y_pos, peak_dist) &&
        x_pos < w && y_pos < h) {       // the reference's buffer lookup would panic outside ("Failed depth buffer lookup")
        const float d = *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(depth) + (size_t)y_pos * depth_pitch + (size_t)x_pos * 4);
        vis = peak_dist - 10.0f < linear_depth(d);
    }
    visible[i] = vis ? 1 : 0;
    xy[2 * i] = vis ? x_pos : 0u;
    xy[2 * i + 1] = vis ? y_pos : 0u;
}

// ---- GeoTIFF rows: predictor, byte order, placement --------------------------------------------------------
// Inclusive prefix sum over `n` elements of a row held in global memory, in place, by one 256-thread workgroup: each
// thread sums a contiguous chunk, the 256 partial sums are scanned in LDS, each thread rewrites its chunk.
template <typename T, typename Load, typename Store>
__device__ void row_prefix_sum(uint32_t n, Load load, Store store) {
    __shared__ uint32_t part[256];
    const uint32_t per = (n + 255) / 256, lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += load(i);
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t a = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;      // exclusive prefix of this chunk
    for (uint32_t i = lo; i < hi; ++i) {
        run += load(i);
        store(i, (T)run);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_tiff_rows(uint8_t* __restrict__ bytes, const TiffSegDev* __restrict__ segs,
                                                   const uint32_t* __restrict__ row_seg, float* __restrict__ out, uint32_t W, uint32_t H,
                                                   uint32_t predictor, int big_endian) {
    const TiffSegDev sg = segs[row_seg[blockIdx.x]];
    const uint32_t r = blockIdx.x - sg.row0, y = sg.y0 + r;
    uint8_t* row = bytes + sg.byte_off + (size_t)r * sg.w * 4;
    auto word = [&](uint32_t i) {           // sample i of the row in the file's byte order -> native
        const uint8_t* p = row + 4 * (size_t)i;
        return big_endian ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
                          : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
    };
    if (predictor == 3) {
        // floating-point predictor: the row is stored as four byte planes, most significant first, the whole 4w-byte
        // sequence differenced byte-wise (Adobe Photoshop TIFF Technical Note 3)
        row_prefix_sum<uint8_t>(sg.w * 4, [&](uint32_t i) { return (uint32_t)row[i]; }, [&](uint32_t i, uint8_t v) { row[i] = v; });
        for (uint32_t i = threadIdx.x; i < sg.w; i += 256) {
            const uint32_t x = sg.x0 + i;
            if (x < W && y < H)
                out[(size_t)y * W + x] = __uint_as_float(((uint32_t)row[i] << 24) | ((uint32_t)row[sg.w + i] << 16) |
                                                         ((uint32_t)row[2 * sg.w + i] << 8) | row[3 * sg.w + i]);
        }
        return;
    }
    if (predictor == 2) {                   // horizontal differencing of the 32-bit words
        uint32_t* wrow = reinterpret_cast<uint32_t*>(row);      // segments start 4-byte aligned in the staging buffer
        for (uint32_t i = threadIdx.x; i < sg.w; i += 256) wrow[i] = word(i);
        __syncthreads();
        row_prefix_sum<uint32_t>(sg.w, [&](uint32_t i) { return wrow[i]; }, [&](uint32_t i, uint32_t v) { wrow[i] = v; });
        for (uint32_t i = threadIdx.x; i < sg.w; i += 256) {
            const uint32_t x = sg.x0 + i;
            if (x < W && y < H) out[(size_t)y * W + x] = __uint_as_float(wrow[i]);
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < sg.w; i += 256) {
        const uint32_t x = sg.x0 + i;
        if (x < W && y < H) out[(size_t)y * W + x] = __uint_as_float(word(i));
    }
}

__global__ void k_probe_sincos(const float* x, float* s, float* c, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_f(x[i], s[i], c[i]);
}

__global__ void k_probe_div(int kind, const float* x, const float* y, float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = kind == 0   ? div_f(x[i], y[i])
             : kind == 1 ? div_const(x[i], 255.0f, 1.0f / 255.0f)
             : kind == 2 ? div_const(x[i], 0.15f - 0.05f, 1.0f / (0.15f - 0.05f))
             : kind == 4 ? __builtin_amdgcn_fractf(x[i])      // v_fract_f32, the instruction itself
             : kind == 5 ? fract_f(x[i])                       // the spec's fract as the kernels evaluate it
                         : sqrt_f(x[i]);
    if (kind >= 6 && kind <= 8) {      // fs_main's dither (mode 0) of channel kind - 6 at p = (x, y), shading 0.25: the wave-level choice of the fraction's form
        float c[4];
        shade_fragment(0, f3{0.0f, 0.0f, 0.25f / 0.7f}, 0.0f, 0.0f, x[i], y[i], f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 1.0f}, c);
        out[i] = c[kind - 6];
    }
}

}  // namespace

// ======================================================================================================
// launchers
// ======================================================================================================

void launch_block_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s) {
    if (count == 0) return;
    const uint32_t bxc = (w - 1 + kBCX - 1) / kBCX, byc = (h - 1 + kBCY - 1) / kBCY;
    const uint32_t runs = (bxc + kTblBlocks - 1) / kTblBlocks * byc;
    hipLaunchKernelGGL(k_block_tables, dim3((runs + 3) / 4, count), dim3(256), 0, s, tiles, first, w, h, bxc, byc);
}

// The load path that reads the DEM once (normals_tables_fused()): sin/cos tables -> normals + block minima / maxima -> f64 bounds.
bool normals_tables_fused(uint32_t w, uint32_t h, int lds_rows) {
    static const bool off = getenv("TOPO_LOAD_FUSED") && atoi(getenv("TOPO_LOAD_FUSED")) == 0;      // experiments: the separate kernels
    return !off && lds_rows == 0 && w >= 4 * kBCX && w % (4 * kBCX) == 0 && h >= 3;
}
void launch_trig_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_trig_tables, dim3((w + h + 255) / 256, count), dim3(256), 0, s, tiles, first, w, h);
}
void launch_normals_tables(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s) {
    if (count == 0) return;
    const uint32_t bxc = (w - 1 + kBCX - 1) / kBCX, byc = (h - 1 + kBCY - 1) / kBCY;
    const uint32_t pieces = (w / (4 * kBCX)) * ((h + kBCY - 1) / kBCY) * count;
    // rows per wave x waves of the 15-row workgroup (TOPO_FUSED_SHAPE = rows * 10 + waves; a wave's rows are loaded in one round)
    static const int shape = getenv("TOPO_FUSED_SHAPE") ? atoi(getenv("TOPO_FUSED_SHAPE")) : 44;
    const dim3 grid(((pieces + 7) / 8) * 8);
    switch (shape) {
        case 53: hipLaunchKernelGGL((k_normals_rolling<5, 3, true, 5>), grid, dim3(192), 0, s, tiles, first, count, (int)w, (int)h, bxc, byc); break;
        case 35: hipLaunchKernelGGL((k_normals_rolling<3, 5, true, 3>), grid, dim3(320), 0, s, tiles, first, count, (int)w, (int)h, bxc, byc); break;
        case 82: hipLaunchKernelGGL((k_normals_rolling<8, 2, true, 4>), grid, dim3(128), 0, s, tiles, first, count, (int)w, (int)h, bxc, byc); break;
        case 151: hipLaunchKernelGGL((k_normals_rolling<15, 1, true, 5>), grid, dim3(64), 0, s, tiles, first, count, (int)w, (int)h, bxc, byc); break;
        default: hipLaunchKernelGGL((k_normals_rolling<4, 4, true, 4>), grid, dim3(256), 0, s, tiles, first, count, (int)w, (int)h, bxc, byc); break;
    }
}
void launch_block_bounds(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, hipStream_t s) {
    if (count == 0) return;
    const uint32_t bxc = (w - 1 + kBCX - 1) / kBCX, byc = (h - 1 + kBCY - 1) / kBCY;
    if (count <= 4) hipLaunchKernelGGL(k_block_bounds<8>, dim3((bxc * byc + 31) / 32, count), dim3(256), 0, s, tiles, first, w, h, bxc, byc);      // latency
    else hipLaunchKernelGGL(k_block_bounds<1>, dim3((bxc * byc + 255) / 256, count), dim3(256), 0, s, tiles, first, w, h, bxc, byc);               // throughput
}

void launch_normals_interior(const TileDev* tiles, uint32_t first, uint32_t count, uint32_t w, uint32_t h, int lds_rows,
                             hipStream_t s) {
    if (count == 0) return;
    const dim3 block(256);
    if (lds_rows == 0 && w % 4 == 0) {      // the register-rolling form: no LDS tile
        // rows per wave x waves per workgroup.  Measured on the c4 load phase (tools/exp_roll.py, ms for K1-K3 of 100 tiles, +-2 %):
        // 4x1 0.211, 4x4 0.217, 8x1 0.222, 8x4 0.223, 16x4 0.238, 48x4 0.242; one row per wave 0.232; the LDS form with 32 rows
        // 0.225-0.233.  Short-lived single-wave workgroups stream best (a plain 16-byte copy of the same bytes: 6.3 TB/s as one
        // load and one store per thread, 5.0-5.3 as a grid-stride loop: tools/calib.hip copy).  TOPO_ROLL_SHAPE = rows * 10 + waves.
        static const int shape = getenv("TOPO_ROLL_SHAPE") ? atoi(getenv("TOPO_ROLL_SHAPE")) : 41;
#define TOPO_ROLL(R, WV) hipLaunchKernelGGL((k_normals_rolling<R, WV, false>), dim3(((((w + 255) / 256) * (((h + (R)-1) / (R) + (WV)-1) / (WV)) * count + 7) / 8) * 8), dim3(64 * (WV)), 0, s, tiles, first, count, (int)w, (int)h, 0u, 0u)
        switch (shape) {
            case 44: TOPO_ROLL(4, 4); break;
            case 81: TOPO_ROLL(8, 1); break;
            case 84: TOPO_ROLL(8, 4); break;
            case 164: TOPO_ROLL(16, 4); break;
            case 484: TOPO_ROLL(48, 4); break;
            default: TOPO_ROLL(4, 1); break;
        }
#undef TOPO_ROLL
        return;
    }
#define TOPO_K1(R)                                                                                                                    \
    hipLaunchKernelGGL(k_normals_interior<R>, dim3(((((w + 127) / 128) * ((h + (R)-1) / (R)) * count + 7) / 8) * 8), block, 0, s, tiles, first, \
                       count, (int)w, (int)h)
    switch (lds_rows) {
        case 4: TOPO_K1(4); break;
        case 8: TOPO_K1(8); break;
        case 16: TOPO_K1(16); break;
        case 64: TOPO_K1(64); break;
        default: TOPO_K1(32); break;
    }
#undef TOPO_K1
}

void launch_normals_border(const TileDev* tiles, const EdgeJob* edges, uint32_t n_edges, const CornerJob* corners, uint32_t n_corners, uint32_t w,
                           uint32_t h, hipStream_t s) {
    const uint32_t chunks = (w + 63) / 64, blocks = chunks * n_edges + (n_corners + 63) / 64;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_normals_border, dim3(blocks), dim3(64), 0, s, tiles, edges, n_edges, chunks, corners, n_corners, (int)w, (int)h);
}

void launch_clear(const FrameParams& p, uint32_t* zero, hipStream_t s, hipEvent_t start) {
    const size_t n = (size_t)p.n_views * p.W * p.H;
    if (start) hipExtLaunchKernelGGL(k_clear, dim3(2048), dim3(256), 0, s, start, nullptr, 0, p.vis, p.dirty, n, p.counters, zero);
    else hipLaunchKernelGGL(k_clear, dim3(2048), dim3(256), 0, s, p.vis, p.dirty, n, p.counters, zero);
}
// The view constants of a submission, from the kernel's own argument segment into the device slot the frame's kernels read.
__global__ __launch_bounds__(256) void k_put_views(ViewPack pack, uint32_t n_words, uint32_t* __restrict__ dst) {
    static_assert(sizeof(ViewPack) % 4 == 0 && sizeof(ViewPack) / 4 <= 256, "one word per lane");
    const auto src = (const __attribute__((address_space(4))) uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();      // `pack` is the first argument
    if (threadIdx.x < n_words) dst[threadIdx.x] = src[threadIdx.x];
}

void launch_put_views(const ViewPack& pack, uint32_t n, ViewDev* dst, hipStream_t s) {
    // (hipExtAnyOrderLaunch, which would let this launch pass under the tail of the frame before, is not honoured on gfx9: measured, no change)
    hipLaunchKernelGGL(k_put_views, dim3(1), dim3(256), 0, s, pack, n * (uint32_t)(sizeof(ViewDev) / 4), (uint32_t*)dst);
}

void launch_clear_cull(const FrameParams& p, uint32_t* zero, hipStream_t s, hipEvent_t start, const ViewPack* pack, uint32_t n_pack_views) {
    const size_t total = (size_t)p.n_views * p.n_tiles * p.bx_count * p.by_count;
    const unsigned n_cull = (unsigned)((total + 255) / 256), n_clear = 2048;
    static const ViewPack none{};
    const ViewPack& pk = pack ? *pack : none;
    const uint32_t words = pack ? n_pack_views * (uint32_t)(sizeof(ViewDev) / 4) : 0u;
    if (start) hipExtLaunchKernelGGL(k_clear_cull, dim3(n_cull + n_clear), dim3(256), 0, s, start, nullptr, 0, p, n_cull, n_clear, zero, pk, words);
    else hipLaunchKernelGGL(k_clear_cull, dim3(n_cull + n_clear), dim3(256), 0, s, p, n_cull, n_clear, zero, pk, words);
}

void launch_cull(const FrameParams& p, hipStream_t s) {
    const size_t total = (size_t)p.n_views * p.n_tiles * p.bx_count * p.by_count;
    if (total == 0) return;
    hipLaunchKernelGGL(k_cull, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
}

// Persistent-style grids: exactly as many workgroups as are resident at once (occupancy x CUs), each wave striding
// over its queue, so there is no partially filled second round of workgroups.
// The size is a property of (kernel, device): cached per device id, so one process can drive several GPUs.
template <int kSite, typename K>
static unsigned resident_grid(K kernel, unsigned fallback) {
    constexpr int kMaxDev = 64;
    static std::atomic<unsigned> cache[kMaxDev];      // 0 = not computed yet; one array per call site (kSite)
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return fallback;
    if (unsigned g = cache[dev].load(std::memory_order_relaxed)) return g;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return fallback;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) return fallback;
    const unsigned g = (unsigned)(cus * per_cu);
    cache[dev].store(g, std::memory_order_relaxed);
    return g;
}

void launch_raster(const FrameParams& p, int phase, hipStream_t s) {
    if (p.n_tiles == 0) return;
    const unsigned grid = resident_grid<0>(k_raster, 256 * 5);
    hipLaunchKernelGGL(k_raster, dim3(grid), dim3(256), 0, s, p, phase);
}


void launch_occlusion(const FrameParams& p, hipStream_t s) {
    if (p.n_tiles == 0) return;
    const unsigned grid = resident_grid<1>(k_occlusion, 256 * 8);
    hipLaunchKernelGGL(k_occlusion, dim3(grid), dim3(256), 0, s, p);
}

void launch_raster_rare(const FrameParams& p, hipStream_t s) {
    if (p.n_tiles == 0) return;
    hipLaunchKernelGGL(k_raster_rare, dim3(256), dim3(256), 0, s, p);
}

void launch_raster_big(const FrameParams& p, hipStream_t s) {
    if (p.n_tiles == 0) return;
    const unsigned grid = resident_grid<2>(k_raster_big, 256 * 4);
    hipLaunchKernelGGL(k_raster_big, dim3(grid), dim3(256), 0, s, p);
}

void launch_resolve(const FrameParams& p, const OutputParams& o, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    const unsigned n_blocks = p.rblock_count;
    if (n_blocks == 0) return;
    // four times the resident workgroups: the hardware then hands out workgroups as others finish, which evens out what the
    // static stride leaves uneven (c3: 0.21 -> 0.16 ms; c4, with 128 blocks per resident workgroup, does not care)
    // (the four instantiations differ by a few instructions: one occupancy query serves them all)
    unsigned resident = 4u * resident_grid<3>(k_resolve<true, false>, 256 * TOPO_RESOLVE_WGS);
    if (const char* e = getenv("TOPO_RESOLVE_GRID")) resident = (unsigned)atoi(e) ? (unsigned)atoi(e) : n_blocks;      // experiments: 0 = one block per workgroup
    const dim3 grid(n_blocks < resident ? n_blocks : resident), block(256);
    const bool bgra = p.bgra && !p.post_off;      // (the render-target image of the pixelise path is always R G B A)
    auto launch = [&](auto kernel) {
        if (start || stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, start, stop, 0, p, o);
        else hipLaunchKernelGGL(kernel, grid, block, 0, s, p, o);
    };
    if (!p.linear_target && !bgra) launch(k_resolve<true, false>);
    else if (!p.linear_target) launch(k_resolve<true, true>);
    else if (!bgra) launch(k_resolve<false, false>);
    else launch(k_resolve<false, true>);
}

void launch_overlay(const OverlayVertex* verts, const uint32_t* idx, uint32_t n_tris, uint32_t n_verts, float width, int32_t W, int32_t H, uint64_t* keys,
                    bool keys_fresh, uint8_t* rgba, size_t pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s) {
    const size_t n = (size_t)W * H;
    if (keys_fresh) hipLaunchKernelGGL(k_overlay_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n);
    if (n_tris == 0) return;
    hipLaunchKernelGGL(k_overlay_raster, dim3((n_tris + 63) / 64), dim3(64), 0, s, verts, idx, n_tris, n_verts, width, W, H, keys);
    hipLaunchKernelGGL(k_overlay_resolve, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, verts, idx, width, W, H, keys, rgba, pitch, linear_target, bgra);
}

void launch_overlay_glyphs(const GlyphInstance* glyphs, uint32_t n_glyphs, float depth, const uint8_t* atlas, uint32_t aw, uint32_t ah, int32_t W, int32_t H,
                           uint64_t* keys, bool keys_fresh, uint8_t* rgba, size_t pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s) {
    const size_t n = (size_t)W * H;
    if (keys_fresh) hipLaunchKernelGGL(k_overlay_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n);
    if (n_glyphs == 0) return;
    hipLaunchKernelGGL(k_glyph_raster, dim3(n_glyphs), dim3(64), 0, s, glyphs, n_glyphs, depth, W, H, keys);
    hipLaunchKernelGGL(k_glyph_resolve, dim3(n_glyphs), dim3(64), 0, s, glyphs, n_glyphs, depth, atlas, aw, ah, W, H, keys, rgba, pitch, linear_target, bgra);
}

void launch_post_pixelize(uint32_t n_views, int32_t W, int32_t H, float vw, float vh, float pixelize_n, const uint8_t* pre_rgba, const OutputParams& out,
                          const float* depth, size_t depth_view_stride, size_t depth_pitch, uint32_t linear_target, uint32_t bgra, hipStream_t s) {
    hipLaunchKernelGGL(k_post_pixelize, dim3((W + 63) / 64, (H + 3) / 4, n_views), dim3(256), 0, s, W, H, vw, vh, pixelize_n, pre_rgba, out, depth, depth_view_stride,
                       depth_pitch, linear_target, bgra);
}

void launch_visible_peaks(const float* proj16_dev, uint32_t w, uint32_t h, const float* depth, size_t depth_pitch, uint32_t n,
                          const float* peaks_xyz, uint8_t* visible, uint32_t* xy, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_visible_peaks, dim3((n + 255) / 256), dim3(256), 0, s, proj16_dev, w, h, depth, depth_pitch, n, peaks_xyz,
                       visible, xy);
}

void launch_tiff_rows(uint8_t* bytes, const TiffSegDev* segs, const uint32_t* row_seg, uint32_t n_rows, float* out, uint32_t W, uint32_t H,
                      uint32_t predictor, bool big_endian, hipStream_t s) {
    if (n_rows) hipLaunchKernelGGL(k_tiff_rows, dim3(n_rows), dim3(256), 0, s, bytes, segs, row_seg, out, W, H, predictor, big_endian ? 1 : 0);
}

void launch_probe_sincos(const float* x, float* s, float* c, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_probe_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, s, c, n);
}

void launch_probe_div(int kind, const float* x, const float* y, float* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_probe_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kind, x, y, out, n);
}

}  // namespace topo
