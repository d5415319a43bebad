"""oracle/ray_check.py -- an INDEPENDENT f64 cross-check of the CPU oracle (test infrastructure, never shipped).

The oracle (topo_oracle.cpp) and the product were written by the same hand from the same reading of the reference, and
the reference holds no fixture for this path ("parity unpinned").  This module shrinks the risk of a
shared misreading: it renders the same scene a completely different way -- one f64 ray per pixel centre against the
triangle mesh, no projection matrix, no rasteriser, no fixed point, no f32 -- from its own derivation of the
reference's geometry, and compares depth and per-pixel winner with the oracle's.

What it derives itself (each from the reference's source text, not from the oracle):
  * sphere mapping           resources/shaders/render_shader.wgsl:35-45,58-64   (to_model, R0 + h, lon/lat -> ECEF)
  * mesh topology + winding  topo-renderer/src/render/render_buffer.rs:185-219  (vertex (i,j) = texel (x=i,y=j); index order)
  * camera                   topo-renderer/src/data/camera.rs:97-128            (up = eye/|eye|; direction = arc(-Y -> up) * (cos yaw cos
                             pitch, sin pitch, sin yaw cos pitch); look_to_rh; perspective_rh(fov_y, w/h, 50, 500000))
  * face culling             topo-renderer/src/render/pipeline.rs:221-223       (front = counter-clockwise, cull back)
  * depth                    perspective_rh's z in [0,1] (glam), depth test Less (pipeline.rs:226-232): nearest hit wins
  * draw order               terrain_renderer.rs:407-420 (BTreeMap order) -- only matters for exact depth ties, which an f64 ray
                             cast has none of; NOT cross-checked here.

A misreading of handedness, of the yaw or pitch sign, of the row/column meaning of the vertex grid, of the triangle
winding, of the depth range or of the field of view moves or empties the ray-cast picture and shows up as a gross
mismatch.  NOT covered: everything sub-pixel (fill rule, 1/256 px snapping, f32 rounding), shading, the post pass,
normals -- those stay pinned only by the authored known-answer tests.
"""
from __future__ import annotations

import math

import numpy as np

R0 = 6371000.0            # render_shader.wgsl:1
NEAR, FAR = 50.0, 500000.0   # data/camera.rs:6-7


def tile_vertices(heights, raster_point, model_point, pixel_scale):
    """ECEF f64 position of every vertex (i = x = column, j = y = row): array [h, w, 3] indexed [j, i]."""
    h, w = heights.shape
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    lon = np.radians((x - float(raster_point[0])) * float(pixel_scale[0]) + float(model_point[0]))
    lat = np.radians((y - float(raster_point[1])) * -float(pixel_scale[1]) + float(model_point[1]))
    r = R0 + heights.astype(np.float64)
    return np.stack([r * np.cos(lat) * np.cos(lon), r * np.cos(lat) * np.sin(lon), r * np.sin(lat) * np.ones_like(lon)], axis=-1)


def tile_triangles(w, h):
    """Index-buffer order of generate_indices: for i (outer), j (inner): two triangles; returns [(n,3,2)] (i, j) vertex ids."""
    tris = []
    for i in range(w - 1):
        for j in range(h - 1):
            a, b, c, d = (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1)
            if (i + j) % 2 == 0:
                tris += [(a, b, d), (d, c, a)]
            else:
                tris += [(a, b, c), (d, c, b)]
    return np.asarray(tris, dtype=np.int64)


def rotation_arc(frm, to):
    """The minimal rotation taking unit vector `frm` to unit vector `to` (what Quat::from_rotation_arc represents), as a matrix."""
    frm, to = np.asarray(frm, np.float64), np.asarray(to, np.float64)
    c = float(np.dot(frm, to))
    axis = np.cross(frm, to)
    s = float(np.linalg.norm(axis))
    if s < 1e-15:
        if c > 0:
            return np.eye(3)
        raise ValueError("antiparallel arc: not used by the test scenes")
    k = axis / s
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + s * K + (1.0 - c) * (K @ K)          # Rodrigues


def camera_basis(eye, yaw, pitch):
    """(forward, right, up_cam) of Camera::get_view: look_to_rh(eye, direction, up) looks along `direction`."""
    eye = np.asarray(eye, np.float64)
    up = eye / np.linalg.norm(eye)
    local = np.array([math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)])
    f = rotation_arc([0.0, -1.0, 0.0], up) @ local
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    return f, s, u


def ray_cast(tiles, eye, yaw, pitch, fov_y, W, H, chunk=256, with_bary=False):
    """tiles: [(heights f32 [h,w], raster_point, model_point, pixel_scale)] in DRAW order.
    Returns per pixel: depth_ndc (1.0 = no hit), winner (tile_rank * 2(w-1)(h-1) + triangle, -1 = none), the smallest
    barycentric of the winning hit (how far inside its triangle the pixel centre lies), and the view depth of the
    nearest hit when back faces are NOT culled (np.inf = none).  with_bary: also the weights of the winning triangle's second
    and third vertex at the hit point (barycentrics in the triangle's own plane = perspective-correct interpolation weights)."""
    eye = np.asarray(eye, np.float64)
    f, s, u = camera_basis(eye, yaw, pitch)
    th = math.tan(0.5 * fov_y)
    aspect = W / H
    V0, E1, E2, ids = [], [], [], []
    base = 0
    for rank, (hts, rp, mp, ps) in enumerate(tiles):
        h, w = hts.shape
        P = tile_vertices(hts, rp, mp, ps)
        T = tile_triangles(w, h)
        v0 = P[T[:, 0, 1], T[:, 0, 0]]
        v1 = P[T[:, 1, 1], T[:, 1, 0]]
        v2 = P[T[:, 2, 1], T[:, 2, 0]]
        V0.append(v0 - eye)
        E1.append(v1 - v0)
        E2.append(v2 - v0)
        ids.append(base + np.arange(len(T)))
        base += 2 * (w - 1) * (h - 1)
    V0, E1, E2, ids = np.concatenate(V0), np.concatenate(E1), np.concatenate(E2), np.concatenate(ids)
    N = np.cross(E1, E2)                                   # geometric normal; counter-clockwise seen from where N points
    px = (np.arange(W) + 0.5) / W * 2.0 - 1.0
    py = 1.0 - (np.arange(H) + 0.5) / H * 2.0              # framebuffer y runs down, NDC y up
    gx, gy = np.meshgrid(px, py)
    D = f[None, :] + (gx.reshape(-1, 1) * th * aspect) * s[None, :] + (gy.reshape(-1, 1) * th) * u[None, :]     # not normalised: t = view depth
    n_pix = D.shape[0]
    depth = np.ones(n_pix)
    winner = np.full(n_pix, -1, np.int64)
    minbary = np.zeros(n_pix)
    any_depth = np.full(n_pix, np.inf)
    wu, wv = np.zeros(n_pix), np.zeros(n_pix)
    for lo in range(0, n_pix, chunk):
        d = D[lo:lo + chunk]                               # [c,3]
        # Moeller-Trumbore with the ray origin at the (translated) origin: o - v0 = -V0
        pvec = np.cross(d[:, None, :], E2[None, :, :])     # [c,n,3]
        det = np.einsum("nk,cnk->cn", E1, pvec)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tvec = -V0[None, :, :]
            bu = np.einsum("cnk,cnk->cn", np.broadcast_to(tvec, pvec.shape), pvec) * inv
            qvec = np.cross(np.broadcast_to(tvec, pvec.shape), E1[None, :, :])
            bv = np.einsum("ck,cnk->cn", d, qvec) * inv
            t = np.einsum("nk,cnk->cn", E2, qvec) * inv
        hit = (bu >= 0) & (bv >= 0) & (bu + bv <= 1) & (t >= NEAR) & (t < FAR) & np.isfinite(t)
        front = np.einsum("nk,ck->cn", N, d) < 0           # the viewer is on the side N points to
        t_any = np.where(hit, t, np.inf)
        any_depth[lo:lo + chunk] = t_any.min(axis=1)
        t_front = np.where(hit & front, t, np.inf)
        k = t_front.argmin(axis=1)
        rows = np.arange(len(d))
        tw = t_front[rows, k]
        ok = np.isfinite(tw)
        with np.errstate(invalid="ignore"):
            z = FAR * (tw - NEAR) / ((FAR - NEAR) * tw)    # perspective_rh depth of a point at view depth t
        depth[lo:lo + chunk] = np.where(ok, z, 1.0)
        winner[lo:lo + chunk] = np.where(ok, ids[k], -1)
        b1, b2 = bu[rows, k], bv[rows, k]
        minbary[lo:lo + chunk] = np.where(ok, np.minimum(np.minimum(b1, b2), 1.0 - b1 - b2), 0.0)
        wu[lo:lo + chunk] = np.where(ok, b1, 0.0)
        wv[lo:lo + chunk] = np.where(ok, b2, 0.0)
    if with_bary:
        return depth.reshape(H, W), winner.reshape(H, W), minbary.reshape(H, W), any_depth.reshape(H, W), wu.reshape(H, W), wv.reshape(H, W)
    return depth.reshape(H, W), winner.reshape(H, W), minbary.reshape(H, W), any_depth.reshape(H, W)


def compare(oracle_depth, oracle_winner, ray_depth, ray_winner, minbary, interior=0.03, slack_clip=4.0):
    """Statistics of oracle vs ray cast.  `interior` pixels: the ray hit lies at least that far (in barycentric units)
    inside its triangle, so 1/256 px snapping and the fill rule cannot change the owner.

    Depth tolerance.  The reference's vertex shader forms clip = (projection * view) * position in f32 with the matrix
    product precomposed on the CPU (camera.rs:122-128) and positions of magnitude 6.4e6 m: z_clip and w each come out of a
    sum of ~6e6-sized terms and carry up to about one unit (metre) of cancellation noise, so z_ndc = z_clip / w is only
    good to about 2 / w -- at 20 km that is 200 m of view depth, inherent to the reference's own pipeline on any GPU.  The
    tolerance is therefore slack_clip / (view depth) + 4 ulp: loose in metres far away, but a misread depth range
    (-1..1 instead of 0..1), near/far pair or field of view is off by orders of magnitude more."""
    od = oracle_depth.astype(np.float64)
    ow = oracle_winner.astype(np.int64)
    ow = np.where(ow == 0xFFFFFFFF, -1, ow)
    inner = (ray_winner >= 0) & (minbary >= interior)
    same = ow == ray_winner
    with np.errstate(divide="ignore"):
        dist = FAR * NEAR / (FAR - ray_depth * (FAR - NEAR))       # view depth of the ray hit
    tol = slack_clip / dist + 4 * float(np.spacing(np.float32(1.0)))
    derr = np.abs(od - ray_depth)
    sky_o, sky_r = od >= 1.0, ray_winner < 0
    return {
        "pixels": int(od.size),
        "terrain_pixels_ray": int((~sky_r).sum()),
        "interior_pixels": int(inner.sum()),
        "interior_winner_agree": float(same[inner].mean()) if inner.any() else 1.0,
        "interior_depth_within_tol": float((derr[inner] <= tol[inner]).mean()) if inner.any() else 1.0,
        "all_winner_agree": float(same.mean()),
        "sky_agree": float((sky_o == sky_r).mean()),
        "max_interior_depth_err_clip_units": float((derr[inner] * dist[inner]).max()) if inner.any() else 0.0,
        "median_interior_depth_err_clip_units": float(np.median(derr[inner] * dist[inner])) if inner.any() else 0.0,
    }
