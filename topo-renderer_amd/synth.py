"""Synthetic COP90-shaped DEM tiles and viewpoints.

Heights are 5-octave value-noise fBm driven by an integer hash of *global* texel coordinates, so the
field is continuous across tile borders and bit-identical wherever it is generated (numpy here, C++ in
libtopo_hip.so's `topo_synth_tile`; both use IEEE f32 ops in the same order).  This is input data for
tests and benches, not part of the render path.
"""
from __future__ import annotations

import numpy as np

SEED_DEFAULT = 0x5EED0001
_WAVELENGTHS = (512, 256, 128, 64, 32)          # texels
_AMPS = (1.0, 0.5, 0.25, 0.125, 0.0625)
_NORM = np.float32(3000.0 / 1.9375)             # -> [0, 3000] m


def _hash(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    """uint32 mix of lattice coordinates -> float32 in [0,1) with 24 significant bits."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (iy.astype(np.uint32) * np.uint32(0x85EBCA77)) \
            ^ np.uint32((seed * 0xC2B2AE3D) & 0xFFFFFFFF)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def synth_tile(lat_deg: int, lon_deg: int, w: int = 1200, h: int = 1200, seed: int = SEED_DEFAULT) -> np.ndarray:
    """Heights (h, w) float32, row 0 = north, for the 1x1 degree tile whose SW corner is (lat_deg, lon_deg)."""
    gx = (np.int64(lon_deg) + 180) * w + np.arange(w, dtype=np.int64)
    gy = (89 - np.int64(lat_deg)) * h + np.arange(h, dtype=np.int64)
    acc = np.zeros((h, w), dtype=np.float32)
    for o, (wl, amp) in enumerate(zip(_WAVELENGTHS, _AMPS)):
        cx, cy = gx // wl, gy // wl
        fx = ((gx % wl).astype(np.float32) / np.float32(wl))[None, :]
        fy = ((gy % wl).astype(np.float32) / np.float32(wl))[:, None]
        ux = fx * fx * (np.float32(3.0) - np.float32(2.0) * fx)
        uy = fy * fy * (np.float32(3.0) - np.float32(2.0) * fy)
        s = seed + o
        X0, Y0 = np.meshgrid(cx, cy)
        v00 = _hash(X0, Y0, s)
        v10 = _hash(X0 + 1, Y0, s)
        v01 = _hash(X0, Y0 + 1, s)
        v11 = _hash(X0 + 1, Y0 + 1, s)
        a = v00 + ux * (v10 - v00)
        b = v01 + ux * (v11 - v01)
        v = a + uy * (b - a)
        acc = acc + np.float32(amp) * v
    return (acc * _NORM).astype(np.float32)


def tile_transform(lat_deg: int, lon_deg: int, w: int = 1200, h: int = 1200):
    """(raster_point, model_point, pixel_scale) as the GeoTIFF tags of a COP90 tile give them
    (coordinate_transform.rs:23-55): tie point = NW corner at raster (0,0), pixel scale 1/w, 1/h degrees."""
    raster_point = np.array([0.0, 0.0], dtype=np.float32)
    model_point = np.array([float(lon_deg), float(lat_deg + 1)], dtype=np.float32)
    pixel_scale = np.array([1.0 / w, 1.0 / h], dtype=np.float32)
    return raster_point, model_point, pixel_scale


def mosaic_locations(lat0: int, lon0: int, n_lat: int, n_lon: int):
    """Tile SW corners of an n_lat x n_lon mosaic in the insertion order of SURVEY.md 8d:
    row-major, north to south, west to east."""
    return [(lat0 + n_lat - 1 - r, lon0 + c) for r in range(n_lat) for c in range(n_lon)]


def height_at(heights: np.ndarray, lat_deg: int, lon_deg: int, lon: float, lat: float) -> float:
    """get_height_value_at (coordinate_transform.rs:72-86): truncating raster lookup."""
    h, w = heights.shape
    rp, mp, ps = tile_transform(lat_deg, lon_deg, w, h)
    rx = (np.float32(lon) - mp[0]) / ps[0] + rp[0]
    ry = (np.float32(lat) - mp[1]) / -ps[1] + rp[1]
    return float(heights[int(ry), int(rx)])
