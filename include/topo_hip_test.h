/* topo_hip_test.h -- TEST HOOKS of libtopo_hip.so.  Not part of the drop-in boundary (include/topo_hip.h): nothing here
 * has a counterpart in the reference, a host application has no use for any of it, and the Rust bindings
 * (rust/topo-hip-sys, generated from topo_hip.h) do not declare it.  The library exports these symbols so that the parity
 * tests (tests/test_gpu_parity.py, through the ctypes binding) can drive the product's own code paths: the queue overflow
 * branches, the device forms of the arithmetic spec, the normal texture as the kernels left it, and the synthetic
 * COP90-shaped tiles of the benchmark. */
#ifndef TOPO_HIP_TEST_H
#define TOPO_HIP_TEST_H

#include "topo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: capacities (entries) of the big-triangle and rare-triangle queues; 0 restores the default (4 Mi each,
 * the rare queue growing on demand under topo_render).  Lets the tests drive the overflow paths: a full big queue is
 * handled exactly (slower); a full rare queue of an explicitly set capacity drops triangles and makes the call that waits
 * for the frame fail with TOPO_ERR_CAPACITY.  rare_cap with bit 31 set = "start at (rare_cap & 0x7FFFFFFF) entries and
 * grow on demand", i.e. the default behaviour from a small starting size. */
int topo_debug_set_queue_caps(topo_ctx* ctx, uint32_t big_cap, uint32_t rare_cap);

/* Test accessor: 1 if the last submission launched its far phase (occlusion test + second raster pass), 0 if the host could
 * prove that no block lies beyond the occlusion split (a lone tile around the viewpoint) and left the four launches out.
 * TOPO_FAR_SKIP=0 in the environment always launches it. */
int topo_debug_far_phase_launched(topo_ctx* ctx, int32_t* out);

/* Test accessor: the tile's Rgba8Unorm normal texture, w*h*4 bytes, host pointer. */
int topo_read_normals(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg, uint8_t* out);

/* The load-time tables of a tile, read back (unit tests: the two load paths must leave the same tables).  minmax_out: 2 floats per
 * raster block (60 x 15 cells); trig_out: 2 (w + h) floats, the (sin, cos) pairs of the w vertex longitudes, then of the h vertex
 * latitudes; bounds_out: 17 doubles per raster block (4 sphere | 12 corner directions | 1 sagitta, each part contiguous over the
 * blocks).  Any pointer may be null.  n_blocks_out: the number of raster blocks. */
int topo_read_tile_tables(topo_ctx* ctx, int32_t lat_deg, int32_t lon_deg, float* minmax_out, float* trig_out, double* bounds_out, uint32_t* n_blocks_out);

/* Synthetic COP90-shaped tile for tests and benches (integer-hash fBm): w*h floats. */
void topo_synth_tile(int32_t lat_deg, int32_t lon_deg, uint32_t w, uint32_t h, uint32_t seed, float* out);

/* GPU unit-test probe: the device sin/cos of the arithmetic spec over n host floats. */
int topo_probe_sincos(topo_ctx* ctx, const float* x, float* s, float* c, size_t n);

/* GPU unit-test probe: the device forms of the spec's IEEE divisions over n host floats.  kind 0: x / y (general
 * form, operands inside 2^-96 .. 2^96); kind 1: x / 255; kind 2: x / (0.15f - 0.05f); kind 3: sqrt(x) (y ignored for 1..5);
 * kind 4: the v_fract_f32 instruction; kind 5: the spec's x - floor(x); kinds 6..8: channel kind - 6 of fs_main's mode-0 colour
 * (render_shader.wgsl:75-87,106) for the dither argument p = (x, y) and a shading of 0.25. */
int topo_probe_div(topo_ctx* ctx, int32_t kind, const float* x, const float* y, float* out, size_t n);

#ifdef __cplusplus
}
#endif

#endif /* TOPO_HIP_TEST_H */
