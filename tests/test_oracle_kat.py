"""Known-answer tests that pin the CPU oracle (the reference has no test on this path, so these
are authored from its source text) and the committed golden vectors."""
import hashlib
import math
import os

import numpy as np
import pytest

from scenes import Scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R0 = 6371000.0


def flat(tile, h=1000.0):
    return np.full((tile, tile), h, np.float32)


def tr(topo, lat, lon, tile):
    return topo.synth.tile_transform(lat, lon, tile, tile)


def test_flat_tile_normals(topo, orc):
    # (1) flat tile -> interior texel (128,128,255,0), border ring untouched (compute_normals_shader.wgsl:30-33)
    o = orc.OracleRenderer(8, 8)
    o.add_terrain(45, 15, flat(16), *tr(topo, 45, 15, 16))
    n = o.read_normals(45, 15, 16, 16)
    assert (n[1:-1, 1:-1] == np.array([128, 128, 255, 0], np.uint8)).all()
    border = np.ones((16, 16), bool)
    border[1:-1, 1:-1] = False
    assert (n[border] == 0).all()


def expected_normal(hT, hL, hR, hB, lat_deg, scale):
    # closed form of compute_normals_shader.wgsl:35-45 in f64
    x = math.radians(scale) * R0
    y = math.radians(scale) * R0 * math.cos(math.radians(lat_deg))
    n = np.array([-2 * y * (hR - hL), -2 * x * (hT - hB), 4 * x * y])
    n /= np.linalg.norm(n)
    return np.floor(np.clip(0.5 * (n + 1), 0, 1) * 255 + 0.5)


def test_ramp_normals_closed_form(topo, orc):
    # (2) planar east / north ramps
    tile = 32
    xs = np.arange(tile, dtype=np.float32)
    east = np.tile(100.0 * xs, (tile, 1)).astype(np.float32)            # h grows eastwards
    north = np.tile((100.0 * (tile - 1 - xs))[:, None], (1, tile)).astype(np.float32)   # h grows northwards (row 0 = north)
    for hts, (dT_B, dR_L) in ((east, (0.0, 200.0)), (north, (200.0, 0.0))):
        o = orc.OracleRenderer(8, 8)
        o.add_terrain(45, 15, hts, *tr(topo, 45, 15, tile))
        n = o.read_normals(45, 15, tile, tile)
        for (yy, xx) in ((1, 1), (10, 20), (30, 30)):
            lat = 46.0 - yy / tile
            e = expected_normal(dT_B, 0.0, dR_L, 0.0, lat, 1.0 / tile)
            assert np.abs(n[yy, xx, :3].astype(int) - e).max() <= 1, (yy, xx, n[yy, xx], e)
            assert n[yy, xx, 3] == 0


def test_seams_and_corner_write_pattern(topo, orc):
    # (3) 2x2 block: which texels each seam / corner pass writes, and that the corner's `top` tap comes from the
    # bottom-right tile (compute_normals_corner_shader.wgsl:49)
    tile = 12
    sc = Scene(tile, 2, 2)
    o = orc.OracleRenderer(8, 8)
    nw, ne, sw, se = sc.locs          # N->S, W->E insertion order
    o.add_terrain(*nw, sc.heights[nw], *sc.transform(nw))
    n = o.read_normals(*nw, tile, tile)
    assert (n[:, -1] == 0).all() and (n[-1, :] == 0).all()
    o.add_terrain(*ne, sc.heights[ne], *sc.transform(ne))
    a, b = o.read_normals(*nw, tile, tile), o.read_normals(*ne, tile, tile)
    assert (a[1:-1, -1, :3] != 0).any(axis=-1).all() and np.array_equal(a[1:-1, -1], b[1:-1, 0])   # shared seam normal
    assert (a[0, -1] == 0).all() and (a[-1, -1] == 0).all()            # seam end points skipped (edge_shader.wgsl:33)
    o.add_terrain(*sw, sc.heights[sw], *sc.transform(sw))
    c = o.read_normals(*sw, tile, tile)
    a = o.read_normals(*nw, tile, tile)
    assert np.array_equal(a[-1, 1:-1], c[0, 1:-1]) and (a[-1, -1] == 0).all()
    o.add_terrain(*se, sc.heights[se], *sc.transform(se))
    a, b, c, d = (o.read_normals(*l, tile, tile) for l in (nw, ne, sw, se))
    corner = a[-1, -1]
    assert (corner[:3] != 0).any()
    assert np.array_equal(corner, b[-1, 0]) and np.array_equal(corner, c[0, -1]) and np.array_equal(corner, d[0, 0])
    H = sc.heights
    hT = float(H[se][tile - 2, 0])       # `top` from terrain_heightmap_rb at (0, H-2)
    hL = float(H[nw][tile - 1, tile - 2])
    hR = float(H[ne][tile - 1, 1])
    hB = float(H[sw][1, tile - 1])
    lat_row = (se[0] + 1) - (tile - 1) / tile          # uniforms of the NEW tile (se), row H-1
    e = expected_normal(hT, hL, hR, hB, lat_row, 1.0 / tile)
    assert np.abs(corner[:3].astype(int) - e).max() <= 1


def test_seam_depends_on_insertion_order(topo, orc):
    # top/bottom seam latitude is row H-1 of whichever tile arrives last (terrain_renderer.rs:275)
    tile = 10
    sc = Scene(tile, 2, 1)
    north, south = sc.locs
    rows = 8000.0 * np.arange(2 * tile, dtype=np.float32)                 # one north-south ramp across both tiles
    hts = {north: np.tile(rows[:tile, None], (1, tile)), south: np.tile(rows[tile:, None], (1, tile))}
    res = []
    for order in ((north, south), (south, north)):
        o = orc.OracleRenderer(8, 8)
        for loc in order:
            o.add_terrain(*loc, hts[loc], *sc.transform(loc))
        res.append(o.read_normals(*north, tile, tile)[-1, 1:-1].copy())
        new = order[1]
        lat_row = (new[0] + 1) - (tile - 1) / tile
        e = expected_normal(float(rows[tile - 2]), 0.0, 0.0, float(rows[tile + 1]), lat_row, 1.0 / tile)
        assert np.abs(res[-1][3, :3].astype(int) - e).max() <= 1
    assert not np.array_equal(res[0], res[1])


def test_projection_depth_and_dist_roundtrip(topo, orc):
    # (4) a point on the view axis at distance d has depth f(d-n)/(d(f-n)); dist_from_depth inverts it
    sc = Scene(16, 1, 1)
    u = orc.camera_uniforms(sc.eye, 0.3, 0.1, math.radians(60), 640, 480, 15.0, 45.0, 0)
    m = u[:16].astype(np.float64).reshape(4, 4).T          # column-major -> matrix
    eye = sc.eye.astype(np.float64)
    up = eye / np.linalg.norm(eye)
    # direction = third row of the view rotation, negated; recover it from the matrix: clip.w = -z_view = dot(f, p - eye)
    f = m[3, :3]
    assert abs(np.linalg.norm(f) - 1) < 1e-5 and abs(f @ up) < 0.2
    n, fa = 50.0, 500000.0
    for d in (60.0, 1000.0, 40000.0, 400000.0):
        p = np.append(eye + d * f, 1.0)
        clip = m @ p
        depth = clip[2] / clip[3]
        assert abs(clip[3] - d) < 2.0                       # f32 matrix at 6.4e6 magnitude: metre-level noise
        expect = fa * (clip[3] - n) / (clip[3] * (fa - n))
        assert abs(depth - expect) < 2.0 * n / clip[3] ** 2 + 2e-6    # a metre of f32 matrix noise moves depth by n/d^2
        assert abs(orc.lib().oracle_dist_from_depth(np.float32(expect)) - clip[3]) / clip[3] < 5e-3


def test_yaw_turns_left_and_positive_pitch_looks_down(topo, orc):
    sc = Scene(16, 1, 1)
    eye = sc.eye.astype(np.float64)
    up = eye / np.linalg.norm(eye)

    def fwd(yaw, pitch):
        u = orc.camera_uniforms(sc.eye, yaw, pitch, math.radians(60), 100, 100, 0, 0, 0)
        return u[:16].astype(np.float64).reshape(4, 4).T[3, :3]
    f0, f1 = fwd(0.0, 0.0), fwd(0.2, 0.0)
    assert np.cross(f0, f1) @ up > 0                        # +yaw turns counter-clockwise seen from above (to the left)
    assert fwd(0.0, 0.3) @ up < -0.25                       # +pitch looks down (rotation_arc maps -Y to up)
    # panorama sectors therefore step by -45 degrees so that the strip reads left to right (clockwise)
    us = topo.panorama_uniforms(sc.eye, 0.4, 64, 128, 0.0, 0.0)
    fs = [u[:16].astype(np.float64).reshape(4, 4).T[3, :3] for u in us]
    for k in range(8):
        a, b = fs[k], fs[(k + 1) % 8]
        assert np.cross(a, b) @ up < 0 and abs(math.degrees(math.acos(np.clip(a @ b, -1, 1))) - 45.0) < 1e-3


def test_top_left_rule_watertight(topo, orc):
    # (5) single-triangle coverage with the top-left rule: a quad split either way covers every pixel centre once
    W = H = 16
    quad = [(2.0, 2.0), (2.0, 12.0), (12.0, 12.0), (12.0, 2.0)]          # a, b(below), d, c -- screen CCW = a,b,d
    a, b, d, c = quad
    counts = orc.coverage_probe(W, H, [a + b + d, d + c + a])
    inside = np.zeros((H, W), np.uint32)
    inside[2:12, 2:12] = 1            # pixel centres 2.5..11.5; edges at 2.0 (owned: left/top) and 12.0 (not owned)
    assert np.array_equal(counts, inside)
    counts = orc.coverage_probe(W, H, [(2.5, 2.5, 2.5, 12.5, 12.5, 12.5), (12.5, 12.5, 12.5, 2.5, 2.5, 2.5)])
    exp = np.zeros((H, W), np.uint32)
    exp[2:12, 2:12] = 1               # centres exactly on the left/top edges are in, on the right/bottom edges out
    assert np.array_equal(counts, exp)
    assert orc.coverage_probe(W, H, [a + d + b]).sum() == 0             # clockwise on screen = back face, culled
    rng = np.random.default_rng(7)
    for _ in range(50):                # random fans around a shared vertex: no pixel hit twice, none lost
        ctr = rng.uniform(4, 12, 2)
        ang = np.sort(rng.uniform(0, 2 * math.pi, 7))
        pts = [tuple(np.round((ctr + 6 * np.array([math.cos(t), -math.sin(t)])) * 256) / 256) for t in ang]   # CCW on screen (y down)
        tris = [tuple(ctr) + pts[i] + pts[(i + 1) % 7] for i in range(7)]
        tris = [(t[0], t[1], t[2], t[3], t[4], t[5]) for t in tris]
        cnt = orc.coverage_probe(W, H, [tuple(np.float32(v) for v in t) for t in tris])
        assert cnt.max() <= 1


def test_sky_colour_and_contour(topo, orc):
    # (7) sky = sRGB8(0, 0.71, 0.885) after encode -> decode -> encode; (8) contour 0 on constant depth, black at a step
    def enc(l):
        s = 12.92 * l if l <= 0.0031308 else 1.055 * l ** (1 / 2.4) - 0.055
        return int(math.floor(s * 255 + 0.5))
    sky = np.array([enc(0.0), enc(0.71), enc(0.885), 255], np.uint8)
    sc = Scene(64, 1, 1)
    o = orc.OracleRenderer(128, 64)
    sc.load(o)
    o.update(128, 64, sc.uniforms(128, 64, mode=1), topo.post_uniforms(128, 64))
    rgba, depth, pre = o.render(want_pre_post=True)
    assert (rgba[0, 0] == sky).all() and (rgba[depth == 1.0][:, 3] == 255).all()
    is_sky = depth == 1.0
    interior_sky = is_sky.copy()
    interior_sky[1:, :] &= is_sky[:-1, :]; interior_sky[:-1, :] &= is_sky[1:, :]
    interior_sky[:, 1:] &= is_sky[:, :-1]; interior_sky[:, :-1] &= is_sky[:, 1:]
    interior_sky[[0, -1], :] = False; interior_sky[:, [0, -1]] = False
    assert (rgba[interior_sky] == sky).all()
    # first terrain row under the horizon: centre near, sky neighbours at 500 km -> contour/center << 0 -> unchanged;
    # last sky row above terrain: centre = 500 km, neighbours near -> contour/center ~ 3/8.. -> black
    col = 64
    ys = np.where(~is_sky[:, col])[0]
    y_top = ys.min()
    assert (rgba[y_top - 1, col, :3] == 0).all() and rgba[y_top - 1, col, 3] == 255
    assert np.array_equal(rgba[y_top + 3, col], pre[y_top + 3, col]) or (rgba[y_top + 3, col, :3] <= pre[y_top + 3, col, :3]).all()


def test_srgb_tables_match_committed(topo, orc):
    import emul
    d, t = orc.srgb_tables()
    L = emul.lib()
    d2, t2 = np.empty(256, np.float32), np.empty(255, np.float32)
    L.emul_srgb_tables(emul._p(d2), emul._p(t2))
    assert np.array_equal(d.view(np.uint32), d2.view(np.uint32))
    assert np.array_equal(t.view(np.uint32), t2.view(np.uint32))
    assert (np.diff(t) > 0).all()
    for c in range(256):
        assert orc.lib().oracle_srgb_encode(d[c]) == c


@pytest.mark.parametrize("name", ["single_64", "block2x2_24", "nearfield_16"])
def test_oracle_reproduces_golden(topo, orc, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    tile, n_lat, n_lon, W, H, yaw, pitch, fov, dh = g["params"]
    sc = Scene(int(tile), int(n_lat), int(n_lon), eye_dh=float(dh))
    sha = hashlib.sha256(b"".join(sc.heights[l].tobytes() for l in sc.locs)).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), g["heights_sha"]), "synthetic heights changed"
    o = orc.OracleRenderer(int(W), int(H))
    sc.load(o)
    for i, loc in enumerate(sc.locs):
        assert np.array_equal(o.read_normals(loc[0], loc[1], int(tile), int(tile)), g[f"normals_{i}"])
    for mode in (0, 1, 2):
        u = sc.uniforms(int(W), int(H), float(yaw), float(pitch), float(fov), mode)
        assert np.array_equal(u.view(np.uint32), g[f"uniforms_{mode}"].view(np.uint32)), "host camera math changed"
        o.update(int(W), int(H), u, topo.post_uniforms(int(W), int(H)))
        rgba, depth = o.render()
        assert np.array_equal(rgba, g[f"rgba_{mode}"])
        if mode == 0:
            assert np.array_equal(depth.view(np.uint32), g["depth_bits"])


def test_tiled_many_thread_rendering_equals_plain(orc):
    """oracle_render_views_tiled (bench.py's all-cores CPU baseline: a frame's tiles split into runs of the draw order, each into
    its own z-buffer, merged per pixel) gives the bytes of oracle_render_views -- including equal-depth ties between tiles,
    which the merge must hand to the earlier run."""
    from scenes import Scene
    import topo_renderer_amd as T
    sc = Scene(24, 3, 3, eye_dh=300.0)
    sw, sh = 48, 40
    o = orc.OracleRenderer(sw, sh)
    sc.load(o)
    # a duplicate of one tile's heights under a second location would not tie; an exact tie needs the same geometry drawn
    # twice, which the BTreeMap forbids -- so ties are exercised through flat sea-level tiles sharing edge vertices' depth
    views = sc.panorama(sw, sh, yaw0_deg=12.0)
    o.update(sw, sh, views[0], T.post_uniforms(sw, sh))
    ra, da = o.render_views(views, threads=4)
    for groups in (1, 2, 4, 9, 13):
        rb, db = o.render_views_tiled(views, threads=8, groups=groups)
        assert np.array_equal(ra, rb) and np.array_equal(da.view(np.uint32), db.view(np.uint32)), groups


def test_surface_formats_of_the_oracle(orc):
    """render_engine.rs:77-84 picks formats[0] (+ sRGB suffix when offered): Rgba/Bgra x Srgb/plain.  Bgra = the same texels,
    B and R swapped in memory; the plain formats store round(clamp(v) * 255) -- the cleared sky (0, 0.71, 0.885, 1) becomes
    (0, 181, 226, 255) instead of the sRGB codes -- and sample c / 255."""
    from scenes import Scene
    import topo_renderer_amd as T
    sc = Scene(32, 1, 1, eye_dh=200.0)
    W, H = 64, 48
    frames = {}
    for fmt in (1, 2, 3, 4):
        o = orc.OracleRenderer(W, H, color_format=fmt)
        sc.load(o)
        o.update(W, H, sc.uniforms(W, H, 30, 20, 70, 0), T.post_uniforms(W, H))
        frames[fmt] = o.render()
    for a, b in ((1, 2), (3, 4)):
        assert np.array_equal(frames[a][0][..., [2, 1, 0, 3]], frames[b][0]) and np.array_equal(frames[a][1], frames[b][1])
    assert np.array_equal(frames[1][1], frames[3][1])                       # depth does not depend on the colour format
    sky = frames[3][1] == 1.0
    interior = sky & np.roll(sky, 1, 0) & np.roll(sky, -1, 0) & np.roll(sky, 1, 1) & np.roll(sky, -1, 1)
    assert interior.any() and (frames[3][0][interior] == [0, 181, 226, 255]).all()
    srgb_sky = frames[1][0][interior][0]
    assert tuple(srgb_sky) == (0, 219, 242, 255)                            # the exact sRGB codes of 0.71 and 0.885
    assert (frames[1][0] != frames[3][0]).any()


def test_overlay_layering_known_answers(orc):
    """LineRenderer's pass (line_shader.wgsl, depth Greater against the post quad's 1/4096): a counter-clockwise (in y-down
    pixel space: as lyon emits them... front = CCW after the shader's y flip) rectangle on layer 3 covers exactly its pixel
    centres; layer 2 stays under it, layer 100 goes over it, layers 0 and 1 never show; an equal layer keeps the first."""
    from scenes import OVERLAY_VERTEX
    W, H = 32, 16
    o = orc.OracleRenderer(W, H)

    def rect(x0, y0, x1, y1, color, z, flip=False):
        v = np.array([((x0, y0), (0, 0), color, z), ((x1, y0), (0, 0), color, z), ((x1, y1), (0, 0), color, z), ((x0, y1), (0, 0), color, z)], dtype=OVERLAY_VERTEX)
        ix = [0, 3, 2, 0, 2, 1] if not flip else [0, 1, 2, 0, 2, 3]
        return v, np.array(ix, np.uint32)

    def draw(parts):
        vs, ixs, base = [], [], 0
        for v, ix in parts:
            vs.append(v); ixs.append(ix + base); base += len(v)
        img = np.zeros((H, W, 4), np.uint8)
        return o.overlay_lines(np.concatenate(vs), np.concatenate(ixs), img)

    red, green, blue = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)
    a = draw([rect(4, 2, 12, 10, red, 3)])
    cov = (a[..., 3] == 255)
    front = cov.sum() > 0
    if not front:                                     # the other winding is the front-facing one
        a = draw([rect(4, 2, 12, 10, red, 3, flip=True)])
        cov = (a[..., 3] == 255)
    flip = not front
    assert cov.sum() == 8 * 8 and cov[2:10, 4:12].all() and (a[cov][:, :3] == [255, 0, 0]).all()
    assert draw([rect(4, 2, 12, 10, red, 3, flip=not flip)])[..., 3].sum() == 0          # back face: culled
    for z in (0, 1):
        assert draw([rect(4, 2, 12, 10, red, z, flip=flip)])[..., 3].sum() == 0           # not Greater than 1/4096
    over = draw([rect(4, 2, 12, 10, red, 3, flip=flip), rect(8, 4, 16, 12, green, 2, flip=flip), rect(0, 0, 6, 6, blue, 100, flip=flip)])
    assert (over[5, 9, :3] == [255, 0, 0]).all() and (over[11, 9, :3] == [0, 255, 0]).all() and (over[3, 5, :3] == [0, 0, 255]).all()
    same = draw([rect(4, 2, 12, 10, red, 3, flip=flip), rect(4, 2, 12, 10, green, 3, flip=flip)])
    assert (same[5, 9, :3] == [255, 0, 0]).all()                                           # equal depth: the first draw stays


def test_text_overlay_known_answers(orc):
    """glyphon's pipeline as the reference drives it (text_renderer.rs:259-291), on answers that can be worked out by hand:
    opaque black text replaces the pixel; a zero mask leaves it -- but still owns it: a later quad over the same pixel is
    rejected by depth Greater at equal depth; half coverage mixes in LINEAR light (decode, mix, encode: the IEC curve in f64);
    a colour given as sRGB bytes is decoded when the flag says so; the alpha channel follows src.a + dst.a (1 - src.a)."""
    from scenes import GLYPH
    W, H = 16, 8
    atlas = np.zeros((4, 8), np.uint8)
    atlas[:, 0:2] = 255
    atlas[:, 2:4] = 128
    atlas[:, 4:6] = 0

    def glyph(x, y, u, color, srgb=1, w=2, h=2):
        g = np.zeros((), GLYPH)
        g["pos"], g["dim"], g["uv"], g["color"], g["content_type_with_srgb"], g["depth"] = (x, y), (w, h), (u, 0), color, (1, srgb), 100 / 4096
        return g
    base = np.zeros((H, W, 4), np.uint8)
    base[...] = (200, 120, 40, 255)
    gs = np.array([glyph(0, 0, 0, 0xFF000000),              # opaque black
                   glyph(4, 0, 4, 0xFF000000),              # transparent (mask 0) ...
                   glyph(4, 0, 0, 0xFFFFFFFF),              # ... but it owns its pixels: this opaque white one is rejected
                   glyph(8, 0, 2, 0xFF000000),              # mask 128/255 of black
                   glyph(12, 0, 0, 0xFF8040C0, srgb=1),     # an sRGB colour, opaque
                   glyph(12, 4, 0, 0xFF8040C0, srgb=0),     # the same bytes taken as linear
                   glyph(0, 4, 0, 0x80FFFFFF),              # white at alpha 128/255
                   glyph(14, 6, 0, 0xFF000000, w=4, h=4)],  # partly off the target
                  dtype=GLYPH)
    img = orc.OracleRenderer(W, H).overlay_glyphs(gs, atlas, base.copy())
    enc = lambda l: int(np.floor(255 * (12.92 * l if l <= 0.0031308 else 1.055 * l ** (1 / 2.4) - 0.055) + 0.5))
    dec = lambda c: (c / 255) / 12.92 if c / 255 <= 0.04045 else ((c / 255 + 0.055) / 1.055) ** 2.4
    assert (img[0:2, 0:2] == (0, 0, 0, 255)).all()
    assert (img[0:2, 4:6] == base[0, 0]).all()
    a = 128 / 255
    assert tuple(img[0, 8]) == (enc(dec(200) * (1 - a)), enc(dec(120) * (1 - a)), enc(dec(40) * (1 - a)), 255)
    assert tuple(img[0, 12]) == (0x80, 0x40, 0xC0, 255)                       # decode then encode: the bytes come back
    assert tuple(img[4, 12]) == (enc(0x80 / 255), enc(0x40 / 255), enc(0xC0 / 255), 255)
    assert tuple(img[4, 0]) == (enc(a + dec(200) * (1 - a)), enc(a + dec(120) * (1 - a)), enc(a + dec(40) * (1 - a)), 255)
    assert (img[6:8, 14:16] == (0, 0, 0, 255)).all() and (img[2:4] == base[0, 0]).all()
    # a destination that is not opaque: alpha = src.a + dst.a (1 - src.a)
    base2 = base.copy()
    base2[..., 3] = 100
    img2 = orc.OracleRenderer(W, H).overlay_glyphs(gs[3:4], atlas, base2)
    assert img2[0, 8, 3] == int(np.floor(255 * (a + (100 / 255) * (1 - a)) + 0.5))
    # plain (non-sRGB) surface: no decode, no encode
    img3 = orc.OracleRenderer(W, H, color_format=3).overlay_glyphs(gs[3:4], atlas, base.copy())
    assert tuple(img3[0, 8][:3]) == tuple(int(np.floor(c * (1 - a) + 0.5)) for c in (200, 120, 40))
