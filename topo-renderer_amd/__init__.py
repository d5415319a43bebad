"""topo-renderer_amd -- MI355X-native terrain-panorama render path of krzyz/topo-renderer.

Python is only the test/bench binding here: the product is `libtopo_hip.so` (hand-written gfx950 HIP kernels
behind the C ABI of include/topo_hip.h, host orchestration in C++).  This module mirrors the reference's
`TerrainRenderer` method set (topo-renderer/src/render/terrain_renderer.rs: new / update / add_terrain /
unload_terrain / render) over ctypes.

The directory name carries a hyphen, so it is imported through the loader `topo_renderer_amd.py` at the
repository root (``import topo_renderer_amd``).

There is no CPU fallback: if the shared library is missing, or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

from . import panorama, synth  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOPO_HIP_LIB") or os.path.join(_HERE, "libtopo_hip.so")   # TOPO_HIP_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "topo_hip.h")
TEST_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "topo_hip_test.h")      # test hooks: not part of the boundary

TOPO_OK = 0
TOPO_ERR_INVALID, TOPO_ERR_UNSUPPORTED, TOPO_ERR_HIP, TOPO_ERR_NOT_FOUND, TOPO_ERR_CAPACITY = -1, -2, -3, -4, -5
FORMAT_RGBA8_UNORM_SRGB, FORMAT_BGRA8_UNORM_SRGB, FORMAT_RGBA8_UNORM, FORMAT_BGRA8_UNORM = 1, 2, 3, 4
TIMING_SLOTS = 9
TIMING_NAMES = ("clear", "cull", "raster", "occlusion", "raster_big", "resolve", "total", "load", "load_tables")

NEAR, FAR = 50.0, 500000.0           # data/camera.rs:6-7
N_SECTORS = 8                        # fixed panorama sector count


class TopoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"topo_hip error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile libtopo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "all"] + ([] if verbose else ["-s"]))
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    process cannot both own the GPU, so when torch is installed its runtime is loaded first and
    libtopo_hip.so binds to it; processes that never import torch are unaffected."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded C ABI.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no Python/CPU fallback for the render path)")
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, u32, i32, f32, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_size_t
        sigs = {
            "topo_create": (C.c_int, [C.POINTER(vp), C.c_int, u32, u32, u32]),
            "topo_destroy": (None, [vp]),
            "topo_last_error": (C.c_char_p, [vp]),
            "topo_add_terrain": (C.c_int, [vp, i32, i32, vp, u32, u32, vp, vp, vp]),
            "topo_add_terrain_device": (C.c_int, [vp, i32, i32, vp, u32, u32, vp, vp, vp]),
            "topo_unload_terrain": (C.c_int, [vp, i32, i32]),
            "topo_update": (C.c_int, [vp, u32, u32, vp, vp]),
            "topo_render": (C.c_int, [vp, vp, sz, vp, sz]),
            "topo_recompute_normals": (C.c_int, [vp]),
            "topo_render_views_device": (C.c_int, [vp, u32, vp, u32, u32, vp, sz, sz, vp, sz, sz]),
            "topo_set_stream": (C.c_int, [vp, vp]),
            "topo_synchronize": (C.c_int, [vp]),
            "topo_set_normals_lds_rows": (C.c_int, [vp, C.c_int]),
            "topo_debug_set_queue_caps": (C.c_int, [vp, u32, u32]),
            "topo_debug_far_phase_launched": (C.c_int, [vp, vp]),
            "topo_pin_host_buffer": (C.c_int, [vp, vp, sz]),
            "topo_unpin_host_buffer": (C.c_int, [vp, vp]),
            "topo_get_timings": (C.c_int, [vp, vp]),
            "topo_get_timing_history": (C.c_int, [vp, C.c_uint32, vp, vp]),
            "topo_get_counters": (C.c_int, [vp, vp]),
            "topo_set_occlusion_split": (C.c_int, [vp, f32]),
            "topo_set_timing_slots": (C.c_int, [vp, u32]),
            "topo_read_normals": (C.c_int, [vp, i32, i32, vp]),
            "topo_read_tile_tables": (C.c_int, [vp, i32, i32, vp, vp, vp, vp]),
            "topo_probe_sincos": (C.c_int, [vp, vp, vp, vp, sz]),
            "topo_probe_div": (C.c_int, [vp, i32, vp, vp, vp, sz]),
            "topo_set_pipeline_depth": (C.c_int, [vp, i32]),
            "topo_join": (C.c_int, [vp]),
            "topo_frame_status": (C.c_int, [vp, vp]),
            "topo_overlay_lines": (C.c_int, [vp, vp, u32, vp, u32, f32, vp, sz]),
            "topo_overlay_lines_device": (C.c_int, [vp, vp, u32, vp, u32, f32, vp, sz]),
            "topo_overlay_glyphs": (C.c_int, [vp, vp, u32, C.c_float, vp, u32, u32, vp, sz]),
            "topo_overlay_glyphs_device": (C.c_int, [vp, vp, u32, C.c_float, vp, u32, u32, vp, sz]),
            "topo_change_location_plan": (None, [f32, f32, f32, vp, u32, vp, u32, vp, vp, u32, vp]),
            "topo_change_location": (C.c_int, [vp, f32, f32, f32, vp, u32, vp, vp]),
            "topo_comm_unique_id": (C.c_int, [vp]),
            "topo_comm_init": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int]),
            "topo_comm_from_nccl": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int]),
            "topo_comm_destroy": (None, [vp]),
            "topo_panorama_sector_range": (None, [C.c_int, C.c_int, vp, vp]),
            "topo_panorama_slots": (u32, [C.c_int, u32, u32, vp, u32]),
            "topo_render_panorama": (C.c_int, [vp, vp, vp, f32, f32, u32, u32, f32, f32, i32, vp, vp]),
            "topo_render_batch": (C.c_int, [vp, u32, vp, vp, vp, f32, u32, u32, i32, vp, vp]),
            "topo_visible_peaks": (C.c_int, [vp, u32, vp, vp, vp]),
            "topo_visible_peaks_device": (C.c_int, [vp, vp, u32, u32, vp, sz, u32, vp, vp, vp]),
            "topo_camera_uniforms": (None, [vp, f32, f32, f32, f32, f32, f32, f32, i32, vp]),
            "topo_terrain_uniforms": (None, [vp, vp, vp, u32, u32, vp]),
            "topo_geometry_transform": (None, [f32, f32, f32, vp]),
            "topo_dist_from_depth": (f32, [f32]),
            "topo_pad_256": (u32, [u32]),
            "topo_synth_tile": (None, [i32, i32, u32, u32, u32, vp]),
            "topo_locations_range": (u32, [f32, f32, f32, vp, u32]),
            "topo_coordinate_transform": (C.c_int, [vp, u32, vp, u32, vp, vp, vp, vp]),
            "topo_geotiff_info": (C.c_int, [vp, sz, vp, vp, vp, vp, vp]),
            "topo_sector_fov_y": (f32, [u32, u32, u32]),
            "topo_panorama_uniforms": (None, [vp, f32, f32, u32, u32, f32, f32, i32, u32, vp]),
            "topo_render_device": (C.c_int, [vp, vp, sz, vp, sz]),
            "topo_geotiff_decode": (C.c_int, [vp, vp, sz, vp, sz]),
            "topo_add_terrain_geotiff": (C.c_int, [vp, i32, i32, vp, sz]),
            "topo_to_model": (None, [vp, vp, vp, f32, f32, vp]),
            "topo_to_raster": (None, [vp, vp, vp, f32, f32, vp]),
            "topo_height_value_at": (C.c_int, [vp, u32, u32, vp, vp, vp, C.c_double, C.c_double, vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L._topo_symbols = tuple(sigs)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- host-side helpers (reference CPU code restated in the library) ------------------------------------

def camera_uniforms(eye, yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode=0) -> np.ndarray:
    """Uniforms::new(&camera, bounds) -> 40 x f32 view of the 160-byte struct (render/data.rs:33-58)."""
    out = np.zeros(40, dtype=np.float32)
    e = np.ascontiguousarray(eye, dtype=np.float32)
    lib().topo_camera_uniforms(_p(e), yaw, pitch, fov_y, width, height, sun_theta_deg, sun_phi_deg, view_mode, _p(out))
    return out


def terrain_uniforms(raster_point, model_point, pixel_scale, w, h) -> np.ndarray:
    out = np.zeros(24, dtype=np.float32)
    rp, mp, ps = (np.ascontiguousarray(a, dtype=np.float32) for a in (raster_point, model_point, pixel_scale))
    lib().topo_terrain_uniforms(_p(rp), _p(mp), _p(ps), w, h, _p(out))
    return out


def geometry_transform(h, lon_deg, lat_deg) -> np.ndarray:
    out = np.zeros(3, dtype=np.float32)
    lib().topo_geometry_transform(h, lon_deg, lat_deg, _p(out))
    return out


def dist_from_depth(d: float) -> float:
    return float(lib().topo_dist_from_depth(d))


def pad_256(n: int) -> int:
    return int(lib().topo_pad_256(n))


def synth_tile(lat_deg, lon_deg, w=1200, h=1200, seed=synth.SEED_DEFAULT) -> np.ndarray:
    """C++ twin of synth.synth_tile (bit-identical, ~50x faster)."""
    out = np.empty((h, w), dtype=np.float32)
    lib().topo_synth_tile(lat_deg, lon_deg, w, h, seed & 0xFFFFFFFF, _p(out))
    return out


def locations_range(latitude: float, longitude: float, range_dist: float = 100_000.0):
    """UiController::get_locations_range (ui_controller.rs:61-83): [(lat_deg, lon_deg), ...] in the reference's order."""
    buf = np.zeros((4096, 2), np.int32)
    n = int(lib().topo_locations_range(latitude, longitude, range_dist, _p(buf), 4096))
    return [(int(a), int(b)) for a, b in buf[:min(n, 4096)]]


def change_location_plan(latitude: float, longitude: float, loaded, range_dist: float = 100_000.0):
    """UiController::change_location's set arithmetic (ui_controller.rs:23-59): (to_unload, to_request) for a list of loaded
    (lat_deg, lon_deg) tiles."""
    ld = np.ascontiguousarray(np.array(list(loaded), dtype=np.int32).reshape(-1, 2))
    un, rq = np.zeros((4096, 2), np.int32), np.zeros((4096, 2), np.int32)
    nu, nr = C.c_uint32(), C.c_uint32()
    lib().topo_change_location_plan(latitude, longitude, range_dist, _p(ld) if len(ld) else None, len(ld), _p(un), 4096, C.byref(nu), _p(rq), 4096, C.byref(nr))
    return [(int(a), int(b)) for a, b in un[:nu.value]], [(int(a), int(b)) for a, b in rq[:nr.value]]


class CoordinateTransform:
    """CoordinateTransform (common/coordinate_transform.rs:16-71): the six f32 a tile's GeoTIFF tags reduce to."""

    def __init__(self, raster_point, model_point, pixel_scale):
        self.raster_point = np.asarray(raster_point, np.float32)
        self.model_point = np.asarray(model_point, np.float32)
        self.pixel_scale = np.asarray(pixel_scale, np.float32)

    @classmethod
    def from_geo_tag_data(cls, pixel_scale_data, tie_points_data, model_transformation_data=None):
        """Raises TopoError(UNSUPPORTED) for IncorrectGeoTags and TopoError(INVALID) for IncorrectGeoTagData."""
        arr = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float64)
        ps, tp, mt = arr(pixel_scale_data), arr(tie_points_data), arr(model_transformation_data)
        rp, mp, sc = (np.zeros(2, np.float32) for _ in range(3))
        ptr = lambda a: None if a is None else _p(a)
        rc = lib().topo_coordinate_transform(ptr(ps), 0 if ps is None else ps.size, ptr(tp), 0 if tp is None else tp.size, ptr(mt),
                                             _p(rp), _p(mp), _p(sc))
        if rc != 0:
            raise TopoError(rc, "IncorrectGeoTags" if rc == -2 else "IncorrectGeoTagData")
        return cls(rp, mp, sc)

    def to_model(self, x, y):
        out = np.zeros(2, np.float32)
        lib().topo_to_model(_p(self.raster_point), _p(self.model_point), _p(self.pixel_scale), x, y, _p(out))
        return float(out[0]), float(out[1])

    def to_raster(self, lon, lat):
        out = np.zeros(2, np.float32)
        lib().topo_to_raster(_p(self.raster_point), _p(self.model_point), _p(self.pixel_scale), lon, lat, _p(out))
        return float(out[0]), float(out[1])

    def height_value_at(self, heights: np.ndarray, longitude: float, latitude: float):
        """get_height_value_at (coordinate_transform.rs:73-90): the f32 height, or None where the reference yields None."""
        hts = np.ascontiguousarray(heights, dtype=np.float32)
        h, w = hts.shape
        out = np.zeros(1, np.float32)
        rc = lib().topo_height_value_at(_p(hts), w, h, _p(self.raster_point), _p(self.model_point), _p(self.pixel_scale),
                                        float(longitude), float(latitude), _p(out))
        return None if rc != 0 else float(out[0])


def geotiff_info(data: bytes):
    """(width, height, CoordinateTransform) of a GeoTIFF's first image (topo_geotiff_info); no GPU involved."""
    buf = np.frombuffer(data, np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    rp, mp, sc = (np.zeros(2, np.float32) for _ in range(3))
    rc = lib().topo_geotiff_info(_p(buf), buf.size, C.byref(w), C.byref(h), _p(rp), _p(mp), _p(sc))
    if rc != 0:
        raise TopoError(rc, "GeoTIFF container or geo tags not usable")
    return int(w.value), int(h.value), CoordinateTransform(rp, mp, sc)


def post_uniforms(width, height, pixelize_n=100.0) -> np.ndarray:
    """PostprocessingUniforms::new (render/data.rs:82-89)."""
    return np.array([width, height, pixelize_n, 0.0], dtype=np.float32)


def sector_fov_y(sector_w: int, sector_h: int, n_sectors: int = N_SECTORS) -> float:
    """Vertical FOV that gives each sector a horizontal FOV of 360/n degrees."""
    return 2.0 * math.atan(math.tan(math.pi / n_sectors) * sector_h / sector_w)


def panorama_uniforms(eye, yaw0, sector_w, sector_h, sun_theta_deg, sun_phi_deg, view_mode=0,
                      n_sectors: int = N_SECTORS, pitch: float = 0.0):
    """The n_sectors reference cameras of a 360-degree strip.  The reference's yaw grows counter-clockwise seen
    from above (Camera::direction, camera.rs:101-109), so sector k looks at yaw0 - k*(360/n) degrees: the strip then
    reads left to right, each sector's right edge meeting the next one's left edge."""
    out = np.zeros((n_sectors, 40), dtype=np.float32)
    e = np.ascontiguousarray(eye, dtype=np.float32)
    lib().topo_panorama_uniforms(_p(e), yaw0, pitch, sector_w, sector_h, sun_theta_deg, sun_phi_deg, view_mode, n_sectors, _p(out))
    return [out[k].copy() for k in range(n_sectors)]


# ---- multi-GPU communicator (topo_comm_*: RCCL behind the C ABI) ---------------------------------------

def comm_unique_id() -> np.ndarray:
    """128 bytes from rank 0 (ncclGetUniqueId) for the host to hand to the other ranks."""
    out = np.zeros(128, np.uint8)
    rc = lib().topo_comm_unique_id(_p(out))
    if rc != TOPO_OK:
        raise TopoError(rc, lib().topo_last_error(None).decode())
    return out


class Comm:
    """topo_comm: this process's place in a group of one-process-per-GPU renderers.  world == 1 needs no RCCL."""

    def __init__(self, rank: int = 0, world: int = 1, unique_id=None, device: int = 0, nccl_comm: int = 0):
        h = C.c_void_p()
        if nccl_comm:
            rc = lib().topo_comm_from_nccl(C.byref(h), C.c_void_p(nccl_comm), rank, world)
        else:
            uid = None if unique_id is None else np.ascontiguousarray(unique_id, dtype=np.uint8)
            rc = lib().topo_comm_init(C.byref(h), device, None if uid is None else _p(uid), rank, world)
        if rc != TOPO_OK:
            raise TopoError(rc, lib().topo_last_error(None).decode())
        self._h, self.rank, self.world = h, rank, world

    def close(self):
        if getattr(self, "_h", None):
            lib().topo_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def panorama_slots(world: int, sector_w: int, sector_h: int):
    """The resolve / exchange slots of one panorama (topo_panorama_slots): [(sector within the rank's range, row0, rows)]."""
    n = lib().topo_panorama_slots(world, sector_w, sector_h, None, 0)
    out = np.zeros((max(1, n), 3), np.uint32)
    lib().topo_panorama_slots(world, sector_w, sector_h, _p(out), n)
    return [tuple(int(v) for v in row) for row in out[:n]]


def panorama_sector_range(rank: int, world: int):
    a, b = C.c_uint32(), C.c_uint32()
    lib().topo_panorama_sector_range(rank, world, C.byref(a), C.byref(b))
    return range(a.value, a.value + b.value)


# ---- the TerrainRenderer mirror ------------------------------------------------------------------------

class TerrainRenderer:
    """Mirror of the reference's TerrainRenderer (terrain_renderer.rs) on one MI355X."""

    def __init__(self, width: int, height: int, device: int = 0, color_format: int = FORMAT_RGBA8_UNORM_SRGB):
        h = C.c_void_p()
        rc = lib().topo_create(C.byref(h), device, width, height, color_format)
        if rc != TOPO_OK:
            raise TopoError(rc, lib().topo_last_error(None).decode())
        self._h = h
        self.size = (width, height)
        self.tile_size = None

    def close(self):
        if getattr(self, "_h", None):
            lib().topo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != TOPO_OK:
            raise TopoError(rc, lib().topo_last_error(self._h).decode())

    # add_terrain(location, height_map_data, coordinate_transform, size)   terrain_renderer.rs:173-182
    def add_terrain(self, lat_deg, lon_deg, heights, raster_point, model_point, pixel_scale):
        hts = np.ascontiguousarray(heights, dtype=np.float32)
        h, w = hts.shape
        rp, mp, ps = (np.ascontiguousarray(a, dtype=np.float32) for a in (raster_point, model_point, pixel_scale))
        self._check(lib().topo_add_terrain(self._h, lat_deg, lon_deg, _p(hts), w, h, _p(rp), _p(mp), _p(ps)))
        self.tile_size = (w, h)

    def add_terrain_device(self, lat_deg, lon_deg, heights_ptr: int, w, h, raster_point, model_point, pixel_scale):
        rp, mp, ps = (np.ascontiguousarray(a, dtype=np.float32) for a in (raster_point, model_point, pixel_scale))
        self._check(lib().topo_add_terrain_device(self._h, lat_deg, lon_deg, C.c_void_p(heights_ptr), w, h, _p(rp), _p(mp), _p(ps)))
        self.tile_size = (w, h)

    def unload_terrain(self, lat_deg, lon_deg):
        self._check(lib().topo_unload_terrain(self._h, lat_deg, lon_deg))

    # update(target_size, &uniforms, &postprocessing_uniforms)              terrain_renderer.rs:151-158
    def update(self, width, height, uniforms: np.ndarray, post: np.ndarray):
        u = np.ascontiguousarray(uniforms).view(np.uint8)
        pu = np.ascontiguousarray(post, dtype=np.float32)
        if u.nbytes != 160 or pu.nbytes != 16:
            raise ValueError("uniforms must be 160 bytes and post uniforms 16 bytes")
        self._check(lib().topo_update(self._h, width, height, _p(u), _p(pu)))
        self.size = (width, height)

    # render(...) + depth read-back with the reference's pad_256 row pitch    terrain_renderer.rs:365, render_engine.rs:219-249
    def render(self, want_depth=True, padded_depth=False):
        w, h = self.size
        rgba = np.empty((h, w, 4), np.uint8)
        if not want_depth:
            self._check(lib().topo_render(self._h, _p(rgba), w * 4, None, 0))
            return rgba, None
        pitch = pad_256(4 * w) if padded_depth else 4 * w
        depth = np.zeros((h, pitch // 4), np.float32)
        self._check(lib().topo_render(self._h, _p(rgba), w * 4, _p(depth), pitch))
        return rgba, (depth if padded_depth else depth[:, :w])

    def render_into(self, rgba: np.ndarray, depth: np.ndarray = None):
        """topo_render into caller-owned arrays (rgba (h, w, 4) u8; depth (h, pitch / 4) f32 or None) -- the arrays a caller
        reuses frame after frame and may have pinned (pin_host_buffer)."""
        self._check(lib().topo_render(self._h, _p(rgba), rgba.strides[0], _p(depth) if depth is not None else None,
                                      depth.strides[0] if depth is not None else 0))

    def pin_host_buffer(self, a: np.ndarray):
        self._check(lib().topo_pin_host_buffer(self._h, _p(a), a.nbytes))

    def unpin_host_buffer(self, a: np.ndarray):
        self._check(lib().topo_unpin_host_buffer(self._h, _p(a)))

    def decode_geotiff(self, data: bytes) -> np.ndarray:
        """The f32 raster of a GeoTIFF's first image (topo_geotiff_decode: host container work, GPU predictor/layout)."""
        w, h, _ = geotiff_info(data)
        buf = np.frombuffer(data, np.uint8)
        out = np.empty((h, w), np.float32)
        self._check(lib().topo_geotiff_decode(self._h, _p(buf), buf.size, _p(out), out.size))
        return out

    def add_terrain_geotiff(self, lat_deg: int, lon_deg: int, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        self._check(lib().topo_add_terrain_geotiff(self._h, lat_deg, lon_deg, _p(buf), buf.size))
        w, h, _ = geotiff_info(data)
        self.tile_size = (w, h)

    def render_device(self, rgba_ptr: int, rgba_pitch: int, depth_ptr: int = 0, depth_pitch: int = 0):
        """topo_render_device: the frame of the last update() into device buffers."""
        self._check(lib().topo_render_device(self._h, C.c_void_p(rgba_ptr), rgba_pitch, C.c_void_p(depth_ptr) if depth_ptr else None, depth_pitch))

    def recompute_normals(self):
        self._check(lib().topo_recompute_normals(self._h))

    # LineRenderer::render (line_renderer.rs:200-212): overlay triangles over an image of this renderer's size and format
    def overlay_lines(self, vertices, indices, rgba: np.ndarray, line_width: float = 0.5) -> np.ndarray:
        """vertices: (n, 8) 4-byte words / structured array of 32-byte GpuVertex records; indices u32; rgba (h, w, 4) u8, in place."""
        v = np.ascontiguousarray(vertices)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        assert rgba.dtype == np.uint8 and rgba.flags.c_contiguous
        self._check(lib().topo_overlay_lines(self._h, _p(v), v.nbytes // 32, _p(ix), ix.size, line_width, _p(rgba), rgba.strides[0]))
        return rgba

    def overlay_lines_device(self, vertices, indices, rgba_ptr: int, rgba_pitch: int, line_width: float = 0.5):
        v = np.ascontiguousarray(vertices)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        self._check(lib().topo_overlay_lines_device(self._h, _p(v), v.nbytes // 32, _p(ix), ix.size, line_width, C.c_void_p(rgba_ptr), rgba_pitch))

    def overlay_glyphs(self, glyphs, atlas: np.ndarray, rgba: np.ndarray, depth: float = 100.0 / 4096.0) -> np.ndarray:
        """glyphs: structured array / (n, 7) words of 28-byte GlyphToRender records; atlas (ah, aw) u8 mask; rgba (h, w, 4) u8, in place."""
        g = np.ascontiguousarray(glyphs)
        a = np.ascontiguousarray(atlas, dtype=np.uint8)
        assert rgba.dtype == np.uint8 and rgba.flags.c_contiguous
        self._check(lib().topo_overlay_glyphs(self._h, _p(g), g.nbytes // 28, depth, _p(a), a.shape[1], a.shape[0], _p(rgba), rgba.strides[0]))
        return rgba

    def overlay_glyphs_device(self, glyphs, atlas: np.ndarray, rgba_ptr: int, rgba_pitch: int, depth: float = 100.0 / 4096.0):
        g = np.ascontiguousarray(glyphs)
        a = np.ascontiguousarray(atlas, dtype=np.uint8)
        self._check(lib().topo_overlay_glyphs_device(self._h, _p(g), g.nbytes // 28, depth, _p(a), a.shape[1], a.shape[0], C.c_void_p(rgba_ptr), rgba_pitch))

    def change_location(self, latitude: float, longitude: float, range_dist: float = 100_000.0):
        """UiController::change_location on this renderer's tile set: unloads what left the range, returns (tiles to fetch, n unloaded)."""
        rq = np.zeros((4096, 2), np.int32)
        nr, nu = C.c_uint32(), C.c_uint32()
        self._check(lib().topo_change_location(self._h, latitude, longitude, range_dist, _p(rq), 4096, C.byref(nr), C.byref(nu)))
        return [(int(a), int(b)) for a, b in rq[:nr.value]], int(nu.value)

    def render_panorama(self, comm, eye, yaw0, sector_w, sector_h, sun_theta_deg, sun_phi_deg, strip_ptr: int, depth_ptr: int = 0,
                        view_mode: int = 0, pitch: float = 0.0):
        """topo_render_panorama: this rank's sectors into the sector-major strip [8][sector_h][sector_w][4] (+ depth), then the
        in-place RCCL all-gather (comm None / world 1: all 8 sectors, no collective).  Asynchronous on the context's stream."""
        e = np.ascontiguousarray(eye, dtype=np.float32)
        self._check(lib().topo_render_panorama(self._h, comm._h if comm is not None else None, _p(e), yaw0, pitch, sector_w, sector_h,
                                               sun_theta_deg, sun_phi_deg, view_mode, C.c_void_p(strip_ptr),
                                               C.c_void_p(depth_ptr) if depth_ptr else None))

    def render_batch(self, eyes, yaw0s, suns_theta_phi_deg, sector_w, sector_h, rgba_ptr: int, depth_ptr: int = 0, view_mode: int = 0,
                     pitch: float = 0.0):
        """topo_render_batch: n independent panoramas (eyes (n,3), yaw0s (n,), suns (n,2) degrees) into rgba [n][8][h][w][4]."""
        e = np.ascontiguousarray(eyes, dtype=np.float32).reshape(-1, 3)
        y = np.ascontiguousarray(yaw0s, dtype=np.float32).reshape(-1)
        sn = np.ascontiguousarray(suns_theta_phi_deg, dtype=np.float32).reshape(-1, 2)
        assert len(e) == len(y) == len(sn)
        self._check(lib().topo_render_batch(self._h, len(e), _p(e), _p(y), _p(sn), pitch, sector_w, sector_h, view_mode,
                                            C.c_void_p(rgba_ptr), C.c_void_p(depth_ptr) if depth_ptr else None))

    def render_views_device(self, uniforms_list, width, height, rgba_ptr: int, rgba_view_stride: int, rgba_pitch: int,
                            depth_ptr: int = 0, depth_view_stride: int = 0, depth_pitch: int = 0):
        if isinstance(uniforms_list, np.ndarray) and uniforms_list.dtype == np.uint8 and uniforms_list.ndim == 2:
            us = np.ascontiguousarray(uniforms_list)           # already packed: (n_views, 160) bytes
            assert us.shape[1] == 160
        else:
            us = np.ascontiguousarray(np.stack([np.ascontiguousarray(u).view(np.uint8).reshape(160) for u in uniforms_list]))
        self._check(lib().topo_render_views_device(self._h, len(us), _p(us), width, height,
                                                   C.c_void_p(rgba_ptr), rgba_view_stride, rgba_pitch,
                                                   C.c_void_p(depth_ptr) if depth_ptr else None, depth_view_stride, depth_pitch))

    def set_stream(self, hip_stream: int):
        self._check(lib().topo_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_pipeline_depth(self, depth: int):
        """Frames in flight for render_views_device (topo_set_pipeline_depth); outputs are complete after join()."""
        self._check(lib().topo_set_pipeline_depth(self._h, depth))

    def join(self):
        self._check(lib().topo_join(self._h))

    def synchronize(self):
        self._check(lib().topo_synchronize(self._h))

    def frame_status(self) -> dict:
        """Status of the last frame waited for (topo_frame_status): bits + the bounds record of the check build."""
        out = np.zeros(4, np.uint32)
        self._check(lib().topo_frame_status(self._h, _p(out)))
        return {"status": int(out[0]), "big_overflow": bool(out[0] & 1), "rare_overflow": bool(out[0] & 2), "bounds_violation": bool(out[0] & 4),
                "bounds_site": int(out[1]), "bounds_value": int(out[2]) | (int(out[3]) << 32)}

    def set_normals_lds_rows(self, rows: int):
        self._check(lib().topo_set_normals_lds_rows(self._h, rows))

    def debug_set_queue_caps(self, big_cap: int, rare_cap: int):
        self._check(lib().topo_debug_set_queue_caps(self._h, big_cap, rare_cap))

    def debug_far_phase_launched(self) -> bool:
        """Whether the last submission launched its far phase (include/topo_hip_test.h)."""
        out = np.zeros(1, np.int32)
        self._check(lib().topo_debug_far_phase_launched(self._h, _p(out)))
        return bool(out[0])

    def timings(self) -> dict:
        out = np.zeros(TIMING_SLOTS, np.float32)
        self._check(lib().topo_get_timings(self._h, _p(out)))
        return {k: float(v) for k, v in zip(TIMING_NAMES, out) if k != "_"}

    def timing_history(self, n_frames: int) -> list:
        """Per-kernel durations of the last n_frames frames (at most 32 per frame in flight), oldest first; waits for them."""
        out, n = np.zeros((max(1, n_frames), 7), np.float32), C.c_uint32(0)
        self._check(lib().topo_get_timing_history(self._h, n_frames, _p(out), C.byref(n)))
        return [{k: float(v) for k, v in zip(TIMING_NAMES[:7], row)} for row in out[:n.value]]

    def counters(self) -> dict:
        out = np.zeros(6, np.uint32)
        self._check(lib().topo_get_counters(self._h, _p(out)))
        return {"blocks_rastered": int(out[0]) + int(out[5]), "big_items": int(out[1]), "status": int(out[2]), "rare_items": int(out[3]),
                "near_blocks": int(out[0]), "far_tested": int(out[4]), "far_survived": int(out[5])}   # near/survivors: in strips of 1 or 2 cell rows (terrain_renderer.cpp: near_strip)

    def set_timing_slots(self, names=None, total=True):
        """Measure only the named per-kernel durations (TIMING_NAMES[:6]); None = all, () = just the total; total=False: not
        even that (TOPO_TIMING_NO_TOTAL: no event in front of the frame)."""
        mask = 0x3F if names is None else sum(1 << TIMING_NAMES.index(n) for n in names)
        if not total:
            mask |= 0x80
        self._check(lib().topo_set_timing_slots(self._h, mask))

    def set_occlusion_split(self, metres: float):
        self._check(lib().topo_set_occlusion_split(self._h, metres))

    def read_tile_tables(self, lat_deg, lon_deg) -> dict:
        """The load-time tables of a tile (topo_read_tile_tables, test hook): block min/max, sin/cos tables, f64 cull bounds."""
        w, h = self.tile_size
        n = C.c_uint32(0)
        self._check(lib().topo_read_tile_tables(self._h, lat_deg, lon_deg, None, None, None, C.byref(n)))
        minmax, trig, bounds = np.empty((n.value, 2), np.float32), np.empty((w + h, 2), np.float32), np.empty(n.value * 17, np.float64)
        self._check(lib().topo_read_tile_tables(self._h, lat_deg, lon_deg, _p(minmax), _p(trig), _p(bounds), C.byref(n)))
        return {"minmax": minmax, "trig": trig, "bounds": bounds}

    def read_normals(self, lat_deg, lon_deg) -> np.ndarray:
        w, h = self.tile_size
        out = np.empty((h, w, 4), np.uint8)
        self._check(lib().topo_read_normals(self._h, lat_deg, lon_deg, _p(out)))
        return out

    # RenderEngine::get_visible_labels                                       render_engine.rs:338-396
    def visible_peaks(self, peaks_xyz: np.ndarray):
        """peaks (n,3) f32 ECEF -> (visible (n,) bool, xy (n,2) u32) against the depth of the last render()."""
        pk = np.ascontiguousarray(peaks_xyz, dtype=np.float32).reshape(-1, 3)
        n = pk.shape[0]
        vis = np.zeros(n, np.uint8)
        xy = np.zeros((n, 2), np.uint32)
        self._check(lib().topo_visible_peaks(self._h, n, _p(pk), _p(vis), _p(xy)))
        return vis.astype(bool), xy

    def visible_peaks_device(self, uniforms, width, height, depth_ptr, depth_pitch, n, peaks_ptr, visible_ptr, xy_ptr):
        u = np.ascontiguousarray(uniforms).view(np.uint8)
        self._check(lib().topo_visible_peaks_device(self._h, _p(u), width, height, C.c_void_p(depth_ptr), depth_pitch, n,
                                                    C.c_void_p(peaks_ptr), C.c_void_p(visible_ptr), C.c_void_p(xy_ptr)))

    def probe_div(self, kind: int, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = np.empty_like(x)
        self._check(lib().topo_probe_div(self._h, kind, _p(x), _p(y), _p(out), x.size))
        return out

    def probe_sincos(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s, c = np.empty_like(x), np.empty_like(x)
        self._check(lib().topo_probe_sincos(self._h, _p(x), _p(s), _p(c), x.size))
        return s, c
