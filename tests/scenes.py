"""Seeded synthetic scenes shared by the CPU and GPU parity tests."""
from __future__ import annotations

import math

import numpy as np

import topo_renderer_amd as T


class Scene:
    """A mosaic of synthetic tiles plus one viewpoint, in the insertion order of SURVEY.md 8d."""

    def __init__(self, tile=64, n_lat=1, n_lon=1, lat0=45, lon0=15, vfrac=(0.5123, 0.5217), eye_dh=50.0, seed=T.synth.SEED_DEFAULT,
                 height_fn=None):
        self.tile = tile
        self.locs = T.synth.mosaic_locations(lat0, lon0, n_lat, n_lon)
        if height_fn is None:
            self.heights = {loc: T.synth_tile(loc[0], loc[1], tile, tile, seed) for loc in self.locs}
        else:       # analytic relief h(lat_deg, lon_deg) sampled at the vertices (row 0 = north, tile_transform's layout)
            x = np.arange(tile, dtype=np.float64) / tile
            self.heights = {loc: np.ascontiguousarray(height_fn((loc[0] + 1 - x)[:, None], (loc[1] + x)[None, :]), dtype=np.float32)
                            for loc in self.locs}
        self.vlat = lat0 + n_lat * vfrac[0]
        self.vlon = lon0 + n_lon * vfrac[1]
        tl, to = int(math.floor(self.vlat)), int(math.floor(self.vlon))
        self.ground = T.synth.height_at(self.heights[(tl, to)], tl, to, self.vlon, self.vlat)
        self.eye = T.geometry_transform(self.ground + eye_dh, self.vlon, self.vlat)   # render_engine.rs:327

    def transform(self, loc):
        return T.synth.tile_transform(loc[0], loc[1], self.tile, self.tile)

    def load(self, renderer, order=None):
        for loc in (order or self.locs):
            renderer.add_terrain(loc[0], loc[1], self.heights[loc], *self.transform(loc))

    def uniforms(self, W, H, yaw_deg=0.0, pitch_deg=0.0, fov_deg=60.0, mode=0):
        # Camera::reset puts the sun at the zenith of the viewpoint (camera.rs:89-95)
        return T.camera_uniforms(self.eye, math.radians(yaw_deg), math.radians(pitch_deg), math.radians(fov_deg), W, H,
                                 self.vlon, self.vlat, mode)

    def panorama(self, sector_w, sector_h, yaw0_deg=0.0, mode=0):
        return T.panorama_uniforms(self.eye, math.radians(yaw0_deg), sector_w, sector_h, self.vlon, self.vlat, mode)


def assert_same_frame(a, b, what=""):
    (ra, da), (rb, db) = a, b
    dm = np.argwhere(da.view(np.uint32) != db.view(np.uint32))
    cm = np.argwhere((ra != rb).any(axis=-1))
    msg = f"{what}: {len(dm)} depth and {len(cm)} colour pixels differ"
    if len(dm):
        y, x = dm[0][-2:]
        msg += f"; first depth @({x},{y}) {da[tuple(dm[0])]!r} vs {db[tuple(dm[0])]!r}"
    if len(cm):
        msg += f"; first colour @{tuple(cm[0])} {ra[tuple(cm[0])]} vs {rb[tuple(cm[0])]}"
    assert len(dm) == 0 and len(cm) == 0, msg


def random_peaks(sc: "Scene", n=400, seed=5):
    """Peak positions (ECEF f32, as PeakInstance.position: background_runner.rs peak -> geometry::transform) scattered
    over the scene, some above and some below the terrain surface."""
    rng = np.random.default_rng(seed)
    lats = np.array([l[0] for l in sc.locs])
    lons = np.array([l[1] for l in sc.locs])
    out = np.zeros((n, 3), np.float32)
    for i in range(n):
        la = rng.uniform(lats.min() + 0.02, lats.max() + 0.98)
        lo = rng.uniform(lons.min() + 0.02, lons.max() + 0.98)
        tl, to = int(math.floor(la)), int(math.floor(lo))
        ground = T.synth.height_at(sc.heights[(tl, to)], tl, to, lo, la)
        out[i] = T.geometry_transform(ground + rng.choice([-400.0, -30.0, 0.0, 5.0, 60.0, 800.0]), lo, la)
    return out
