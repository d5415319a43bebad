"""oracle/shade_check.py -- INDEPENDENT f64 restatements of the parts of the path oracle/ray_check.py does not reach
(test infrastructure, never shipped; nothing here calls into topo_oracle.cpp or shares a line with it).

The oracle and the product come from one reading of the reference by one author; the reference holds no fixture for this
path ("parity unpinned").  ray_check.py re-derives the geometry (which triangle a pixel sees, at what
depth).  This module re-derives, each from the reference's source text and in plain numpy float64:

  * fs_main, view modes 1 and 2      resources/shaders/render_shader.wgsl:96-115  (Lambert 0.7 max(n.sun, 0) + 0.01; 0.5 (n + 1))
      with vs_main's normal decode    render_shader.wgsl:66-69 (2 rgb - 1, rotated by normal_to_world_rotation),
      the per-tile rotation           topo-renderer/src/render/data.rs:125-133 (Mat3::from_euler(XYZEx, 0, 90 deg - lat, lon) = Rz(lon) Ry(90 deg - lat)),
      the sun of Camera::reset        topo-renderer/src/data/camera.rs:44-53,89-95 (the zenith of the viewpoint),
      perspective-correct varyings = the barycentrics of the ray's hit point in the triangle's own plane,
      and the sRGB transfer function  IEC 61966-2-1 (the *Srgb surface format, render_engine.rs:77-84)
  * the contour post pass             resources/shaders/postprocessing_shader.wgsl:52-54,68-95 (linear depth, 8-neighbour
                                      Laplacian, smoothstep(0.05, 0.15, contour / centre), mix with black)
  * the normal stencil                resources/shaders/compute_normals_shader.wgsl:22-58 (+ the seam and corner passes'
                                      WRITE PATTERN: compute_normals_edge_shader.wgsl:33,74, compute_normals_corner_shader.wgsl:52-62,
                                      terrain_renderer.rs:204-347)
  * the rasteriser's fill rule        WebGPU / D3D11 "top-left rule" in exact integer arithmetic on the 1/256 px grid

Each function returns what the reference's arithmetic gives in exact / f64 terms; the tests (tests/test_independent_cpu.py)
hold the oracle's f32 results to within the rounding the f32 path itself introduces (<= 1 LSB of an 8-bit code, +-1 normal
code) on pixels that are not on a knife edge.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

R0 = 6371000.0
NEAR, FAR = 50.0, 500000.0


# ---- colour ---------------------------------------------------------------------------------------------------------
def srgb_encode8(lin):
    """linear [0,1] -> 8-bit sRGB code, IEC 61966-2-1, round to nearest."""
    lin = np.clip(np.asarray(lin, np.float64), 0.0, 1.0)
    s = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(np.maximum(lin, 1e-300), 1.0 / 2.4) - 0.055)
    return np.floor(255.0 * s + 0.5).astype(np.int64)


def srgb_decode8(code):
    c = np.asarray(code, np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4))


# ---- fs_main (modes 1, 2) from a ray hit ------------------------------------------------------------------------------
def tile_rotation(model_lon_deg, model_lat_deg):
    """Mat3::from_euler(EulerRot::XYZEx, 0, rad(90 - lat), rad(lon)): extrinsic rotations about x (0), then y, then z."""
    b, c = math.radians(90.0 - model_lat_deg), math.radians(model_lon_deg)
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return rz @ ry


def zenith(lon_deg, lat_deg):
    """LightAngle{theta: lon, phi: lat}.to_vec3() as Camera::reset sets it: the direction geometry::transform gives (lon, lat)."""
    la, lo = math.radians(lat_deg), math.radians(lon_deg)
    return np.array([math.cos(la) * math.cos(lo), math.cos(la) * math.sin(lo), math.sin(la)])


def vertex_normals_world(normal_texels, model_point):
    """vs_main: 2 rgb - 1 of the Rgba8Unorm texel, rotated by the tile's normal_to_world_rotation.  [h, w, 3], indexed [y, x]."""
    n = 2.0 * (normal_texels[..., :3].astype(np.float64) / 255.0) - 1.0
    return n @ tile_rotation(float(model_point[0]), float(model_point[1])).T


def shade_from_hits(winner, bary_u, bary_v, tiles_normals, tri_index_fn, sun, mode):
    """Expected LINEAR rgb per pixel from the ray caster's hits.  winner [H, W] (-1 = none); bary_u / bary_v = the weights of
    the hit triangle's second / third vertex; tiles_normals[rank] = world normals [h, w, 3]; tri_index_fn(winner) ->
    (rank, (x0, y0), (x1, y1), (x2, y2)) arrays of the three vertices in index-buffer order."""
    H, W = winner.shape
    out = np.zeros((H, W, 3))
    hit = winner >= 0
    rank, v0, v1, v2 = tri_index_fn(winner[hit])
    n = np.zeros((hit.sum(), 3))
    for r in np.unique(rank):
        m = rank == r
        tn = tiles_normals[int(r)]
        w0 = 1.0 - bary_u[hit][m] - bary_v[hit][m]
        n[m] = (w0[:, None] * tn[v0[1][m], v0[0][m]] + bary_u[hit][m][:, None] * tn[v1[1][m], v1[0][m]] +
                bary_v[hit][m][:, None] * tn[v2[1][m], v2[0][m]])
    if mode == 2:
        rgb = 0.5 * (n + 1.0)                                   # the un-normalised interpolated normal
    else:
        nn = n / np.linalg.norm(n, axis=1, keepdims=True)
        rgb = np.repeat((0.01 + 0.7 * np.maximum(nn @ sun, 0.0))[:, None], 3, axis=1)
    out[hit] = rgb
    return out, hit


# ---- contour post pass ------------------------------------------------------------------------------------------------
def linear_depth(d):
    return FAR * NEAR / (FAR - np.asarray(d, np.float64) * (FAR - NEAR))


def contour_factor(depth_ndc):
    """smoothstep(0.05, 0.15, (8 lin(c) - sum of the 8 neighbours' lin) / lin(c)); depth taps clamp to the edge."""
    lin = linear_depth(depth_ndc)
    p = np.pad(lin, 1, mode="edge")
    H, W = lin.shape
    s = np.zeros_like(lin)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                s += p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    ratio = (8.0 * lin - s) / lin
    t = np.clip((ratio - 0.05) / (0.15 - 0.05), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t), ratio


# ---- normal stencil ----------------------------------------------------------------------------------------------------
def normal_codes_interior(heights, raster_point, model_point, pixel_scale):
    """compute_normals: the three 8-bit codes (as real numbers BEFORE the floor, i.e. 255 * 0.5 (n + 1) + 0.5) of every
    interior texel, [h, w, 3]; border texels NaN (not written by the interior pass)."""
    h, w = heights.shape
    hh = heights.astype(np.float64)
    x = math.radians(float(pixel_scale[0])) * R0
    rows = np.arange(h, dtype=np.float64)
    lat = (rows - float(raster_point[1])) * -float(pixel_scale[1]) + float(model_point[1])
    y = math.radians(float(pixel_scale[1])) * R0 * np.cos(np.radians(lat))          # the cos factor sits on y, as written
    out = np.full((h, w, 3), np.nan)
    hT, hB = hh[:-2, 1:-1], hh[2:, 1:-1]
    hL, hR = hh[1:-1, :-2], hh[1:-1, 2:]
    yy = y[1:-1, None]
    # cross(right - left, top - bottom) with left = (-x, 0, hL), right = (x, 0, hR), top = (0, y, hT), bottom = (0, -y, hB)
    dx = np.stack([np.full_like(hL, 2.0 * x), np.zeros_like(hL), hR - hL], -1)
    dy = np.stack([np.zeros_like(hL), np.broadcast_to(2.0 * yy, hL.shape), hT - hB], -1)
    n = np.cross(dx, dy)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    out[1:-1, 1:-1] = 255.0 * (0.5 * (n + 1.0)) + 0.5
    return out


def seam_write_pattern(order, w, h):
    """Which texels of which tile the seam and corner passes have written after the tiles of `order` (a list of (lat, lon))
    were added one after the other -- from the orchestration of add_terrain (terrain_renderer.rs:204-347) and the shaders'
    guards: seams write 1 <= id < dim - 1 of the shared border in BOTH tiles for every already loaded edge neighbour;
    a corner texel of all four tiles is written when the other three tiles of a 2 x 2 block were loaded before.
    Returns {loc: bool mask [h, w]} of the border texels written."""
    loaded, masks = [], {loc: np.zeros((h, w), bool) for loc in order}
    for (la, lo) in order:
        have = set(loaded)
        def on(l):
            return l in have
        left, right, top, bottom = (la, lo - 1), (la, lo + 1), (la + 1, lo), (la - 1, lo)
        if on(left):
            masks[left][1:h - 1, w - 1] = True
            masks[(la, lo)][1:h - 1, 0] = True
        if on(right):
            masks[(la, lo)][1:h - 1, w - 1] = True
            masks[right][1:h - 1, 0] = True
        if on(top):
            masks[top][h - 1, 1:w - 1] = True
            masks[(la, lo)][0, 1:w - 1] = True
        if on(bottom):
            masks[(la, lo)][h - 1, 1:w - 1] = True
            masks[bottom][0, 1:w - 1] = True
        # 2 x 2 blocks this tile completes: (lt, rt, lb, rb)
        for lt, rt, lb, rb in (((la + 1, lo - 1), (la + 1, lo), (la, lo - 1), (la, lo)),
                               ((la + 1, lo), (la + 1, lo + 1), (la, lo), (la, lo + 1)),
                               ((la, lo - 1), (la, lo), (la - 1, lo - 1), (la - 1, lo)),
                               ((la, lo), (la, lo + 1), (la - 1, lo), (la - 1, lo + 1))):
            if all(t == (la, lo) or on(t) for t in (lt, rt, lb, rb)):
                masks[lt][h - 1, w - 1] = True
                masks[rt][h - 1, 0] = True
                masks[lb][0, w - 1] = True
                masks[rb][0, 0] = True
        loaded.append((la, lo))
    return masks


# ---- fill rule ----------------------------------------------------------------------------------------------------------
def coverage_exact(W, H, tri_sub):
    """Pixels whose centre a triangle covers under the top-left rule, in exact integer arithmetic.  tri_sub = three (X, Y)
    integer vertices in 1/256 px (y down).  Geometric statement of the rule (WebGPU / D3D11 rasterisation rules): a pixel
    centre strictly inside is covered; one exactly on an edge is covered iff that edge is a TOP edge (horizontal, the
    triangle below it) or a LEFT edge (not horizontal, the triangle to its right).  Only front faces are drawn: counter-
    clockwise as seen on screen (FrontFace::Ccw, cull Back: pipeline.rs:221-223), i.e. with y down the signed area
    (x1 - x0)(y2 - y0) - (y1 - y0)(x2 - x0) is negative.  Returns a bool mask [H, W]."""
    (x0, y0), (x1, y1), (x2, y2) = [(int(a), int(b)) for a, b in tri_sub]
    out = np.zeros((H, W), bool)
    area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)
    if area >= 0:
        return out
    V = [(x0, y0), (x1, y1), (x2, y2)]
    xs, ys = [v[0] for v in V], [v[1] for v in V]
    for py in range(max(0, min(ys) // 256 - 1), min(H, max(ys) // 256 + 2)):
        for px in range(max(0, min(xs) // 256 - 1), min(W, max(xs) // 256 + 2)):
            cx, cy = 256 * px + 128, 256 * py + 128
            inside = True
            for k in range(3):
                (ax, ay), (bx, by), (ox, oy) = V[k], V[(k + 1) % 3], V[(k + 2) % 3]
                side_p = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)          # which side of edge a->b the centre is on ...
                side_o = (bx - ax) * (oy - ay) - (by - ay) * (ox - ax)          # ... and the opposite vertex (never 0: area != 0)
                if side_p == 0:
                    if ay == by:
                        own = oy > ay                                            # top edge: the triangle lies below (larger y)
                    else:
                        # left edge: at the opposite vertex's height the edge's line is to the LEFT of that vertex
                        x_line = Fraction(ax) + Fraction(bx - ax) * Fraction(oy - ay, by - ay)
                        own = x_line < ox
                    inside = inside and own
                else:
                    inside = inside and (side_p > 0) == (side_o > 0)
                if not inside:
                    break
            out[py, px] = inside
    return out
