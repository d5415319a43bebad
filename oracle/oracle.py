"""ctypes wrapper over oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see topo_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_last_error.argtypes = [C.c_void_p]
        L.oracle_add_terrain.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_unload_terrain.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.oracle_update.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_render_views.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.oracle_read_normals.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.oracle_render_winners.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render_views_tiled.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                                C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.oracle_camera_uniforms.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.c_float, C.c_float, C.c_int32, C.c_void_p]
        L.oracle_terrain_uniforms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_geometry_transform.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.oracle_dist_from_depth.restype = C.c_float
        L.oracle_dist_from_depth.argtypes = [C.c_float]
        L.oracle_pad_256.restype = C.c_uint32
        L.oracle_pad_256.argtypes = [C.c_uint32]
        L.oracle_sincos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_srgb_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_srgb_encode.restype = C.c_uint8
        L.oracle_srgb_encode.argtypes = [C.c_float]
        L.oracle_vs_main_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def camera_uniforms(eye, yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode) -> np.ndarray:
    out = np.zeros(40, dtype=np.float32)
    e = np.asarray(eye, dtype=np.float32)
    lib().oracle_camera_uniforms(_p(e), yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode, _p(out))
    return out


def terrain_uniforms(raster_point, model_point, pixel_scale, w, h) -> np.ndarray:
    out = np.zeros(24, dtype=np.float32)
    rp, mp, ps = (np.asarray(a, dtype=np.float32) for a in (raster_point, model_point, pixel_scale))
    lib().oracle_terrain_uniforms(_p(rp), _p(mp), _p(ps), w, h, _p(out))
    return out


def geometry_transform(h, lon_deg, lat_deg) -> np.ndarray:
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_geometry_transform(h, lon_deg, lat_deg, _p(out))
    return out


def sincos(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().oracle_sincos(_p(x), _p(s), _p(c), x.size)
    return s, c


def srgb_tables():
    d, t = np.empty(256, np.float32), np.empty(255, np.float32)
    lib().oracle_srgb_tables(_p(d), _p(t))
    return d, t


def locations_range(latitude, longitude, range_dist=100_000.0):
    L = lib()
    L.oracle_locations_range.restype = C.c_uint32
    L.oracle_locations_range.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_uint32]
    buf = np.zeros((4096, 2), np.int32)
    n = int(L.oracle_locations_range(latitude, longitude, range_dist, _p(buf), 4096))
    return [(int(a), int(b)) for a, b in buf[:min(n, 4096)]]


def pad_256(n: int) -> int:
    return int(lib().oracle_pad_256(n))


class OracleRenderer:
    """CPU restatement of TerrainRenderer (terrain_renderer.rs): new/update/add_terrain/unload_terrain/render."""

    def __init__(self, width: int, height: int, color_format: int = 1):
        self._h = lib().oracle_create(width, height)
        self.size = (width, height)
        if color_format != 1:
            L = lib()
            L.oracle_set_format.argtypes = [C.c_void_p, C.c_uint32]
            self._check(L.oracle_set_format(self._h, color_format))

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error(self._h).decode())

    def add_terrain(self, lat, lon, heights, raster_point, model_point, pixel_scale):
        hts = np.ascontiguousarray(heights, dtype=np.float32)
        h, w = hts.shape
        rp, mp, ps = (np.ascontiguousarray(a, dtype=np.float32) for a in (raster_point, model_point, pixel_scale))
        self._check(lib().oracle_add_terrain(self._h, lat, lon, _p(hts), w, h, _p(rp), _p(mp), _p(ps)))

    def unload_terrain(self, lat, lon):
        self._check(lib().oracle_unload_terrain(self._h, lat, lon))

    def update(self, width, height, uniforms: np.ndarray, post_uniforms: np.ndarray):
        u = np.ascontiguousarray(uniforms).view(np.uint8)
        pu = np.ascontiguousarray(post_uniforms, dtype=np.float32)
        assert u.nbytes == 160 and pu.nbytes == 16
        self._check(lib().oracle_update(self._h, width, height, _p(u), _p(pu)))
        self.size = (width, height)

    def render(self, want_pre_post=False):
        w, h = self.size
        rgba = np.empty((h, w, 4), np.uint8)
        depth = np.empty((h, w), np.float32)
        pre = np.empty((h, w, 4), np.uint8) if want_pre_post else None
        self._check(lib().oracle_render(self._h, _p(rgba), w * 4, _p(depth), w * 4, _p(pre) if pre is not None else None))
        return (rgba, depth, pre) if want_pre_post else (rgba, depth)

    def render_views_tiled(self, uniforms_list, threads, groups):
        """render_views with more threads than frames: each frame's tiles in `groups` runs of the draw order, merged per pixel."""
        w, h = self.size
        n = len(uniforms_list)
        us = np.ascontiguousarray(np.stack([np.ascontiguousarray(u).view(np.uint8).reshape(160) for u in uniforms_list]))
        rgba = np.empty((n, h, w, 4), np.uint8)
        depth = np.empty((n, h, w), np.float32)
        self._check(lib().oracle_render_views_tiled(self._h, n, _p(us), _p(rgba), w * h * 4, w * 4, _p(depth), w * h * 4, w * 4, threads, groups))
        return rgba, depth

    def overlay_lines(self, vertices, indices, rgba, line_width=0.5):
        """LineRenderer::render over `rgba` (h, w, 4) in place: vertices = structured/(n, 8) 32-byte GpuVertex records, indices u32."""
        v = np.ascontiguousarray(vertices)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        L = lib()
        L.oracle_overlay_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_size_t]
        self._check(L.oracle_overlay_lines(self._h, _p(v), v.nbytes // 32, _p(ix), ix.size, line_width, _p(rgba), rgba.strides[0]))
        return rgba

    def overlay_glyphs(self, glyphs, atlas, rgba, depth=100.0 / 4096.0):
        """TextRenderer::render over `rgba` (h, w, 4) in place: glyphs = structured/(n, 7) 28-byte GlyphToRender records,
        atlas (ah, aw) u8 mask."""
        g = np.ascontiguousarray(glyphs)
        a = np.ascontiguousarray(atlas, dtype=np.uint8)
        L = lib()
        L.oracle_overlay_glyphs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        self._check(L.oracle_overlay_glyphs(self._h, _p(g), g.nbytes // 28, depth, _p(a), a.shape[1], a.shape[0], _p(rgba), rgba.strides[0]))
        return rgba

    def render_winners(self):
        """(depth (h,w) f32, winner (h,w) u32): winner = tile rank in draw order * 2(w-1)(h-1) + index-buffer triangle,
        0xFFFFFFFF where nothing was drawn.  Bookkeeping for oracle/ray_check.py."""
        w, h = self.size
        depth = np.empty((h, w), np.float32)
        win = np.empty((h, w), np.uint32)
        self._check(lib().oracle_render_winners(self._h, _p(depth), _p(win)))
        return depth, win

    def render_views(self, uniforms_list, threads=1):
        """n frames -> rgba (n,h,w,4), depth (n,h,w)"""
        w, h = self.size
        n = len(uniforms_list)
        us = np.ascontiguousarray(np.stack([np.ascontiguousarray(u).view(np.uint8).reshape(160) for u in uniforms_list]))
        rgba = np.empty((n, h, w, 4), np.uint8)
        depth = np.empty((n, h, w), np.float32)
        self._check(lib().oracle_render_views(self._h, n, _p(us), _p(rgba), w * h * 4, w * 4, _p(depth), w * h * 4, w * 4, threads))
        return rgba, depth

    def visible_peaks(self, peaks_xyz):
        pk = np.ascontiguousarray(peaks_xyz, dtype=np.float32).reshape(-1, 3)
        n = pk.shape[0]
        vis = np.zeros(n, np.uint8)
        xy = np.zeros((n, 2), np.uint32)
        L = lib()
        L.oracle_visible_peaks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(L.oracle_visible_peaks(self._h, n, _p(pk), _p(vis), _p(xy)))
        return vis.astype(bool), xy

    def read_normals(self, lat, lon, w, h):
        out = np.empty((h, w, 4), np.uint8)
        self._check(lib().oracle_read_normals(self._h, lat, lon, _p(out)))
        return out

    def vs_main_probe(self, lat, lon, x, y):
        out = np.zeros(10, np.float32)
        lib().oracle_vs_main_probe(self._h, lat, lon, x, y, _p(out))
        return out


def coverage_probe(W, H, tris):
    """Accumulated coverage counts of triangles given in framebuffer pixels [(x0,y0,x1,y1,x2,y2), ...]."""
    L = lib()
    L.oracle_coverage_probe.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    counts = np.zeros((H, W), np.uint32)
    for t in tris:
        xy = np.asarray(t, dtype=np.float32)
        L.oracle_coverage_probe(W, H, _p(xy), _p(counts))
    return counts
