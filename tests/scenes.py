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


OVERLAY_VERTEX = np.dtype([("position", np.float32, 2), ("normal", np.float32, 2), ("color", np.float32, 3), ("z_index", np.int32)])


def overlay_geometry(W, H, n_lines=40, n_boxes=12, seed=3):
    """Overlay triangles shaped like what the reference's LineRenderer tessellates (line_renderer.rs): leader lines as quads of
    stroke vertices (position on the path + unit normal, widened by line_width in the shader, z_index 2) and label boxes as
    filled rectangles (normal 0, z_index 3), plus a few triangles on the text layer (100), some clockwise (culled), some
    partly off the target, some at z_index 0/1 (never visible) and some overlapping on the same layer (first one wins)."""
    rng = np.random.default_rng(seed)
    verts, idx = [], []

    def quad(p, nrm, color, z):
        base = len(verts)
        for q, n in zip(p, nrm):
            verts.append((tuple(q), tuple(n), tuple(color), z))
        return base

    for i in range(n_lines):
        a, b = rng.uniform(-0.1, 1.1, 2) * [W, H], rng.uniform(-0.1, 1.1, 2) * [W, H]
        d = (b - a) / max(1e-6, np.linalg.norm(b - a))
        n = np.array([-d[1], d[0]], np.float32)
        base = quad([a, a, b, b], [n, -n, n, -n], rng.uniform(0, 1, 3), 2 if i % 9 else int(rng.choice([0, 1, 5])))
        tri = [base, base + 1, base + 2, base + 2, base + 1, base + 3]
        if i % 7 == 3:
            tri = tri[::-1]
        idx += tri
    for i in range(n_boxes):
        x0, y0 = rng.uniform(0, 0.8, 2) * [W, H]
        w, h = rng.uniform(5, 0.3 * W), rng.uniform(4, 20)
        base = quad([(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)], [(0, 0)] * 4, rng.uniform(0, 1, 3), 3)
        idx += [base, base + 3, base + 2, base, base + 2, base + 1] if i % 2 else [base, base + 2, base + 1, base, base + 3, base + 2]
    for i in range(6):
        p = rng.uniform(0, 1, (3, 2)) * [W, H]
        base = len(verts)
        cols = rng.uniform(0, 1.2, (3, 3))              # per-vertex colours (interpolated), some above 1 (clamped by the store)
        for q, c in zip(p, cols):
            verts.append((tuple(q), (0.0, 0.0), tuple(c), 100))
        idx += [base, base + 1, base + 2, base, base + 2, base + 1]      # both windings: one of them is front-facing
    v = np.array(verts, dtype=OVERLAY_VERTEX)
    return v, np.array(idx, np.uint32)


GLYPH = np.dtype([("pos", np.int32, 2), ("dim", np.uint16, 2), ("uv", np.uint16, 2), ("color", np.uint32), ("content_type_with_srgb", np.uint16, 2),
                  ("depth", np.float32)])      # glyphon 0.10 GlyphToRender, 28 bytes


def glyph_scene(W, H, n_labels=14, seed=7):
    """A mask atlas of synthetic 'glyphs' (antialiased blobs: coverage 0 .. 255) and label-like runs of glyph quads shaped like
    what the reference's text renderer hands glyphon: rows of small quads advancing to the right, black text (the reference's
    default colour) and a few coloured / translucent ones, some runs overlapping (the first quad keeps the pixel), some partly
    or wholly off the target, one glyph whose atlas rectangle sticks out of the atlas."""
    assert GLYPH.itemsize == 28
    rng = np.random.default_rng(seed)
    aw, ah, cell = 96, 64, 16
    atlas = np.zeros((ah, aw), np.uint8)
    yy, xx = np.mgrid[0:cell, 0:cell]
    cells = []
    for cy in range(ah // cell):
        for cx in range(aw // cell):
            gw, gh = int(rng.integers(5, cell)), int(rng.integers(7, cell))
            c = rng.uniform(2, [gw - 2, gh - 2])
            r = rng.uniform(2.0, 6.0)
            cov = np.clip(r - np.hypot(xx - c[0], yy - c[1]) + 0.5, 0, 1) * (xx < gw) * (yy < gh)
            atlas[cy * cell:(cy + 1) * cell, cx * cell:(cx + 1) * cell] = np.round(255 * cov).astype(np.uint8)
            cells.append((cx * cell, cy * cell, gw, gh))
    glyphs = []
    for i in range(n_labels):
        x, y = int(rng.integers(-20, W - 10)), int(rng.integers(-10, H - 4))
        if i % 5 == 4:
            x, y = glyphs[-1]["pos"][0] - 6, glyphs[-1]["pos"][1] + 3          # a run laid over the previous one
        color = 0xFF000000 if i % 3 else (int(rng.integers(0, 256)) << 24) | int(rng.integers(0, 1 << 24))
        for k in range(int(rng.integers(3, 12))):
            ux, uy, gw, gh = cells[int(rng.integers(0, len(cells)))]
            g = np.zeros((), GLYPH)
            g["pos"], g["dim"], g["uv"], g["color"] = (x, y + int(rng.integers(-2, 3))), (gw, gh), (ux, uy), color
            g["content_type_with_srgb"] = (1, 1 if i % 4 else 0)
            g["depth"] = 100.0 / 4096.0
            glyphs.append(g)
            x += gw + 1
    g = np.zeros((), GLYPH)                                                      # its atlas rectangle leaves the atlas: transparent there
    g["pos"], g["dim"], g["uv"], g["color"], g["content_type_with_srgb"], g["depth"] = (W // 2, H // 2), (20, 20), (aw - 8, ah - 8), 0xFF2040FF, (1, 1), 100.0 / 4096.0
    glyphs.append(g)
    return np.array(glyphs, dtype=GLYPH), atlas
