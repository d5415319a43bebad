"""The product's device arithmetic (topo_math.h / topo_pipeline.h, the very headers the HIP kernels compile)
executed on the CPU by tests/host_emul.cpp and compared bit for bit with the independently written oracle.
This is what lets a CPU-only run vouch for the kernels' arithmetic; the GPU run (-m gpu) then only has to show
that hipcc's code generation agrees with g++'s."""
import math

import numpy as np
import pytest

import emul
from scenes import Scene, assert_same_frame


def test_sincos_matches_oracle_and_libm(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-3.3, 3.3, 300000), np.linspace(-10, 10, 100001), np.float32([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi])]).astype(np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    emul.lib().emul_sincos(emul._p(x), emul._p(s), emul._p(c), x.size)
    so, co = orc.sincos(x)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32)) and np.array_equal(c.view(np.uint32), co.view(np.uint32))
    x64 = x.astype(np.float64)
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7 and np.abs(c - np.cos(x64)).max() < 1.5e-7


def test_from_unorm8_is_the_exact_quotient():
    import ctypes as C
    L = emul.lib()
    L.emul_from_unorm8.restype = C.c_float
    L.emul_from_unorm8.argtypes = [C.c_uint32]
    for c in range(256):
        assert np.float32(L.emul_from_unorm8(c)) == np.float32(c) / np.float32(255.0)


def test_constant_division():
    """div_const's device form (one Markstein correction of x * RN(1/C)) is the IEEE quotient: for C = 255 at every
    magnitude, for C = 0.15f - 0.05f inside 2^-97 .. 2^123 (an exhaustive sweep of all 2^32 inputs was run once when the
    form was introduced; this is a dense sample of the same claim)."""
    import ctypes as C
    L = emul.lib()
    L.emul_markstein_div.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    n = 4_000_000
    mant = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    sign = rng.integers(0, 2, n, dtype=np.uint32) << 31
    for c, lo, hi in ((np.float32(255.0), 1, 254), (np.float32(0.15) - np.float32(0.05), 30, 249)):
        expo = rng.integers(lo, hi + 1, n, dtype=np.uint32) << 23
        x = (sign | expo | mant).view(np.float32)
        x[:4] = [0.0, 1.0, -1.0, 0.05]
        out = np.empty_like(x)
        L.emul_markstein_div(x.ctypes.data, float(c), out.ctypes.data, n)
        with np.errstate(all="ignore"):
            ref = x / c
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_srgb_lut_encode_equals_probes():
    """srgb_encode_lut (12-bit table + two probes) counts the same thresholds as the eight-probe search: every f32 in
    [0, 1.01] at a stride coprime to the bin size, all 255 thresholds and their neighbours, the bin edges, and the
    values outside the unit interval."""
    import ctypes as C
    L = emul.lib()
    L.emul_srgb_encode_both.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hi = int(np.float32(1.01).view(np.uint32))
    bits = np.arange(0, hi, 37, dtype=np.uint32)
    edges = (np.arange(0, 4097, dtype=np.float32) / np.float32(4096.0)).view(np.uint32)
    edges = np.concatenate([edges, edges + 1, np.maximum(edges, 1) - 1])
    special = np.array([0.0, -0.0, -1.0, 2.0, 1e30, np.inf, -np.inf, np.nan, 1.0, 0.99999994], np.float32).view(np.uint32)
    x = np.concatenate([bits, edges, special]).view(np.float32)
    a, b = np.empty(x.size, np.uint8), np.empty(x.size, np.uint8)
    L.emul_srgb_encode_both(x.ctypes.data, a.ctypes.data, b.ctypes.data, x.size)
    assert np.array_equal(a, b)
    # around every threshold (the emulation's probe version defines them): find them by bisection of the codes
    xs = np.sort(x[np.isfinite(x) & (x >= 0) & (x <= 1.01)])
    L.emul_srgb_encode_both(xs.ctypes.data, a[:xs.size].ctypes.data, b[:xs.size].ctypes.data, xs.size)
    steps = np.nonzero(np.diff(a[:xs.size].astype(np.int32)))[0]
    assert steps.size == 255
    near = np.concatenate([(xs[steps].view(np.uint32)[:, None] + np.arange(0, 40, dtype=np.uint32)[None, :]).reshape(-1)]).view(np.float32)
    near = np.ascontiguousarray(near)
    a2, b2 = np.empty(near.size, np.uint8), np.empty(near.size, np.uint8)
    L.emul_srgb_encode_both(near.ctypes.data, a2.ctypes.data, b2.ctypes.data, near.size)
    assert np.array_equal(a2, b2)


def test_fastdiv_is_exact():
    import ctypes as C
    L = emul.lib()
    L.emul_fastdiv.restype = C.c_uint32
    L.emul_fastdiv.argtypes = [C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(4)
    ds = [2, 3, 7, 8, 1199, 1200, 3599, 2 * 1199 * 1199, 2 * 3599 * 3599, (1 << 30) + 1, (1 << 31) - 1] + list(rng.integers(2, 1 << 31, 40))
    for d in ds:
        d = int(d)
        ns = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, ((1 << 31) - 1) // d * d, max(((1 << 31) - 1) // d * d - 1, 0)}
        ns |= {int(v) for v in rng.integers(0, 1 << 31, 200)}
        for nn in ns:
            if nn < (1 << 31):
                assert L.emul_fastdiv(nn, d) == nn // d, (nn, d)


def test_vis_key_orders_like_less_then_draw_order():
    L = emul.lib()
    import ctypes as C
    L.emul_vis_key.restype = C.c_uint64
    L.emul_vis_key.argtypes = [C.c_float, C.c_uint32]
    zs = np.float32([0.0, 1e-30, 0.25, 0.5, 0.99999994])
    for i in range(len(zs) - 1):
        assert L.emul_vis_key(zs[i], 0xFFFFFFFE) < L.emul_vis_key(zs[i + 1], 0)        # nearer always wins
    assert L.emul_vis_key(0.5, 10) < L.emul_vis_key(0.5, 11)                            # equal depth: earlier draw wins
    assert L.emul_vis_key(0.99999994, 0xFFFFFFFE) < 0x3F800000FFFFFFFF                  # anything < 1.0 beats the clear value


def test_coverage_matches_oracle(orc):
    import ctypes as C
    L = emul.lib()
    rng = np.random.default_rng(11)
    W, H = 48, 40
    for _ in range(300):
        scale = rng.choice([3.0, 20.0, 200.0])
        p = (rng.uniform(-0.3, 1.3, (3, 2)) * [W, H] * rng.uniform(0.2, 1.0) + rng.uniform(-scale, scale, 2)).astype(np.float32)
        xy = np.ascontiguousarray(p.reshape(6))
        a = orc.coverage_probe(W, H, [xy])
        b = np.zeros((H, W), np.uint32)
        L.emul_coverage_probe(W, H, emul._p(xy), emul._p(b))
        assert np.array_equal(a, b)


CASES = [
    # tile, n_lat, n_lon, W, H, yaw, pitch, fov, mode, eye_dh
    (64, 1, 1, 128, 64, 0, 0, 60, 0, 50),
    (64, 1, 1, 128, 64, 0, 0, 60, 1, 50),
    (64, 1, 1, 128, 64, 0, 0, 60, 2, 50),
    (48, 3, 3, 160, 96, 200, 30, 79.28, 0, 50),
    (64, 2, 2, 128, 128, 77, 60, 100, 0, 100),
    (64, 2, 2, 128, 128, 77, 85, 100, 2, 400),
    (100, 1, 2, 131, 67, 300, 20, 120, 0, 80),
]


@pytest.mark.parametrize("cfg", CASES, ids=[f"e{i}" for i in range(len(CASES))])
def test_header_pipeline_matches_oracle(topo, orc, cfg):
    tile, n_lat, n_lon, W, H, yaw, pitch, fov, mode, dh = cfg
    sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
    e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
    sc.load(e)
    sc.load(o)
    for loc in sc.locs:
        assert np.array_equal(e.read_normals(*loc), o.read_normals(loc[0], loc[1], tile, tile))
    u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
    e.update(W, H, u, pu)
    o.update(W, H, u, pu)
    assert_same_frame(e.render(), o.render(), f"emul {cfg}")


def test_header_pipeline_random_frames(topo, orc):
    """The product headers on the CPU against the oracle over a seeded sweep of poses (steep pitches put triangles across the
    near plane: the homogeneous-varyings path), sizes and view modes."""
    rng = np.random.default_rng(77)
    for i in range(12):
        tile = int(rng.choice([16, 24, 40]))
        n_lat, n_lon = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        dh = float(rng.choice([3.0, 12.0, 50.0, 400.0, 5000.0]))
        sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
        W, H = int(rng.integers(17, 120)), int(rng.integers(9, 90))
        e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
        sc.load(e)
        sc.load(o)
        yaw, pitch = float(rng.uniform(0, 360)), float(rng.choice([rng.uniform(-20, 20), rng.uniform(20, 89)]))
        fov, mode = float(rng.uniform(12, 150)), int(rng.integers(0, 3))
        u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
        e.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(e.render(), o.render(), f"emul random {i}: tile {tile} dh {dh} {W}x{H} yaw {yaw:.1f} pitch {pitch:.1f} fov {fov:.1f} mode {mode}")


def test_full_size_tile_matches_oracle(topo, orc):
    # one real-size COP90 tile (1200x1200, 2.9 M triangles) into a 512x256 frame
    sc = Scene(1200, 1, 1)
    W, H = 512, 256
    e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
    sc.load(e)
    sc.load(o)
    u, pu = sc.uniforms(W, H, 140.0, 8.0, 79.2785, 0), topo.post_uniforms(W, H)
    e.update(W, H, u, pu)
    o.update(W, H, u, pu)
    fe, fo = e.render(), o.render()
    assert_same_frame(fe, fo, "1200x1200 tile")
    assert 0.2 < float((fo[1] < 1).mean()) < 0.9


def test_peak_visibility_matches_oracle(topo, orc):
    # RenderEngine::get_visible_labels (render_engine.rs:338-396)
    import ctypes as C
    from scenes import random_peaks
    sc = Scene(64, 2, 2, eye_dh=400.0)
    W, H = 200, 120                                   # pad_256(4*200) = 1024 > 800: exercises the reference's row pitch
    o = orc.OracleRenderer(W, H)
    sc.load(o)
    peaks = random_peaks(sc, 600)
    seen = 0
    for yaw, pitch in ((0, 20), (120, 35), (250, 10)):
        u = sc.uniforms(W, H, yaw, pitch, 90, 1)
        o.update(W, H, u, topo.post_uniforms(W, H))
        _, depth = o.render()
        vo, xo = o.visible_peaks(peaks)
        ve, xe = np.zeros(len(peaks), np.uint8), np.zeros((len(peaks), 2), np.uint32)
        proj = np.ascontiguousarray(u[:16])
        emul.lib().emul_visible_peaks(emul._p(proj), W, H, emul._p(np.ascontiguousarray(depth)), len(peaks), emul._p(peaks), emul._p(ve), emul._p(xe))
        assert np.array_equal(vo, ve.astype(bool)) and np.array_equal(xo, xe)
        seen += int(vo.sum())
        assert (xo[vo][:, 0] < W).all() and (xo[vo][:, 1] < H).all()
    assert 10 < seen < 3 * len(peaks) - 10          # both outcomes occur


@pytest.mark.parametrize("tw,th", [(40, 30), (30, 40), (3, 3)])
def test_non_square_tiles_match_oracle(topo, orc, tw, th):
    W, H = 96, 64
    e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
    hts = {}
    for (la, lo) in topo.synth.mosaic_locations(-1, -1, 2, 2):       # straddles the equator and the prime meridian
        h = topo.synth_tile(la + 46, lo + 16, max(tw, th), max(tw, th))[:th, :tw].copy()
        tr = (np.float32([0, 0]), np.float32([lo, la + 1]), np.float32([1.0 / tw, 1.0 / th]))
        hts[(la, lo)] = h
        e.add_terrain(la, lo, h, *tr)
        o.add_terrain(la, lo, h, *tr)
    for loc in hts:
        assert np.array_equal(e.read_normals(*loc), o.read_normals(loc[0], loc[1], tw, th)), loc
    eye = topo.geometry_transform(float(hts[(0, 0)].max()) + 900.0, 0.02, 0.03)
    u = topo.camera_uniforms(eye, 0.4, 0.5, math.radians(90), W, H, 0.0, 0.0, 0)
    e.update(W, H, u, topo.post_uniforms(W, H))
    o.update(W, H, u, topo.post_uniforms(W, H))
    assert_same_frame(e.render(), o.render(), f"{tw}x{th}")


# ---- lane bodies of k_raster / k_raster_big under a bounds-checking sink -------------------------------------------
def _tri_cases(rng, n, W, H):
    """Snapped triangles (1/256 px) of every size class: sub-pixel, a few pixels, ~64 px (the int32/int64 border),
    giants, slivers, partly and wholly off-target, degenerate, and ones hugging the target's last row/column."""
    out = []
    for i in range(n):
        kind = i % 8
        cx, cy = rng.uniform(-0.2 * W, 1.2 * W), rng.uniform(-0.2 * H, 1.2 * H)
        size = [0.7, 3.0, 12.0, 40.0, 63.9, 200.0, 3000.0, 30.0][kind]
        pts = np.stack([rng.uniform(-size, size, 3) + cx, rng.uniform(-size, size, 3) + cy], axis=1)
        if kind == 7:                       # sliver: third vertex almost on the edge of the first two
            t = rng.uniform(0, 1)
            pts[2] = pts[0] + t * (pts[1] - pts[0]) + rng.uniform(-0.3, 0.3, 2)
        if i % 11 == 0:                     # touch the far corner of the target
            pts += np.array([W - 1.0, H - 1.0]) - pts[0]
        if i % 13 == 0:
            pts[1] = pts[0]                 # degenerate
        X = np.rint(pts[:, 0] * 256).astype(np.int32)
        Y = np.rint(pts[:, 1] * 256).astype(np.int32)
        z = rng.uniform(0.0, 1.05, 3).astype(np.float32)
        if i % 7 == 0:
            z = rng.uniform(-0.2, 1.3, 3).astype(np.float32)       # crosses both depth clamps
        if rng.integers(0, 2):              # both windings (back faces must emit nothing)
            X[[1, 2]] = X[[2, 1]]; Y[[1, 2]] = Y[[2, 1]]; z[[1, 2]] = z[[2, 1]]
        out.append((X, Y, z))
    return out


@pytest.mark.parametrize("W,H", [(200, 130), (64, 64), (257, 65)])
def test_big_item_lanes_stay_in_bounds_and_match_triangle_pixel(W, H):
    """Every BigItem a triangle can produce (one per overlapped 64x64 px region, as enqueue_big cuts it), through the
    product's 64 lane bodies with a checking sink: no fragment outside the target or the region, none emitted twice, and
    the keys are exactly those of triangle_setup/triangle_pixel restricted to the region."""
    import ctypes as C
    L = emul.lib()
    L.emul_big_item.restype = C.c_int
    clear = np.uint64(0x3F800000FFFFFFFF)
    rng = np.random.default_rng(W * 1000 + H)
    n_medium = n_giant = 0
    for i, (X, Y, z) in enumerate(_tri_cases(rng, 260, W, H)):
        for ry in range((H + 63) // 64):
            for rx in range((W + 63) // 64):
                got = np.full(W * H, clear, np.uint64)
                ref = np.full(W * H, clear, np.uint64)
                ng, nr, med = C.c_uint32(), C.c_uint32(), C.c_int()
                v = L.emul_big_item(emul._p(X), emul._p(Y), emul._p(z), 2 * i, W, H, rx, ry, emul._p(got), C.byref(ng), C.byref(med))
                L.emul_reference_triangle(emul._p(X), emul._p(Y), emul._p(z), 2 * i, W, H, rx, ry, emul._p(ref), C.byref(nr))
                assert v == 0, f"triangle {i} region ({rx},{ry}): {v} bounds/duplicate violations"
                assert ng.value == nr.value and np.array_equal(got, ref), f"triangle {i} region ({rx},{ry})"
                n_medium += med.value; n_giant += 1 - med.value
    assert n_medium > 100 and n_giant > 50


def test_raster_rows_stays_in_bounds_and_matches_triangle_pixel():
    import ctypes as C
    L = emul.lib()
    L.emul_raster_rows.restype = C.c_int
    clear = np.uint64(0x3F800000FFFFFFFF)
    W, H = 96, 70
    rng = np.random.default_rng(11)
    walked = 0
    for i, (X, Y, z) in enumerate(_tri_cases(rng, 1200, W, H)):
        got = np.full(W * H, clear, np.uint64)
        ref = np.full(W * H, clear, np.uint64)
        ng, nr = C.c_uint32(), C.c_uint32()
        v = L.emul_raster_rows(emul._p(X), emul._p(Y), emul._p(z), 2 * i, W, H, emul._p(got), C.byref(ng))
        if v < 0:
            continue                        # spans >= 64 px: not a k_raster triangle
        L.emul_reference_triangle(emul._p(X), emul._p(Y), emul._p(z), 2 * i, W, H, -1, -1, emul._p(ref), C.byref(nr))
        assert v == 0, f"triangle {i}: {v} violations"
        assert ng.value == nr.value and np.array_equal(got, ref), f"triangle {i}"
        walked += ng.value > 0
    assert walked > 100


def test_split_resolve_equals_resolve_varyings(topo, orc):
    """resolve_setup + resolve_pixel (per-triangle record, then per pixel: the route k_resolve takes for waves whose pixels
    share few winners) gives the frame resolve_varyings gives, and both equal the oracle's -- uncut triangles of both
    barycentric widths, primitives cut by the near plane, every view mode."""
    L = emul.lib()
    try:
        for cfg in CASES + [(12, 2, 2, 200, 150, 10, 35, 110, 0, 60), (12, 2, 2, 160, 120, 200, 80, 110, 2, 60), (24, 1, 1, 96, 96, 33, 89, 140, 0, 3)]:
            tile, n_lat, n_lon, W, H, yaw, pitch, fov, mode, dh = cfg
            sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
            e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
            sc.load(e)
            sc.load(o)
            u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
            e.update(W, H, u, pu)
            o.update(W, H, u, pu)
            ref = o.render()
            L.emul_set_split(1)
            assert_same_frame(e.render(), ref, f"split route {cfg}")
            L.emul_set_split(0)
            assert_same_frame(e.render(), ref, f"one-step route {cfg}")
    finally:
        L.emul_set_split(0)


def test_normal_texel_fast_path():
    """normal_texel_fast (reciprocal square root estimate + guard band, the route k_normals_interior takes for 998 texels
    in 1000) never claims a texel that differs from normal_texel's: random stencils at the benchmark's metric steps, steep
    and flat ones, heights that drive a component onto a code boundary (where the guard must hand over), and non-finite
    heights (which must always be handed over)."""
    import ctypes as C
    L = emul.lib()
    L.emul_normal_fast_check.restype = C.c_uint64
    L.emul_normal_fast_check.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(21)
    total_fast = 0
    for rep in range(7):
        n = 4_000_000
        xs = rng.uniform(20.0, 120.0, n).astype(np.float32)          # half-steps in metres (COP90: ~46 x 33 m at 45 degrees)
        ys = rng.uniform(5.0, 120.0, n).astype(np.float32)
        scale = np.float32([0.5, 30.0, 300.0, 3000.0, 3.0e5, 1.0e-3, 1.0][rep])
        h = (rng.standard_normal((n, 4)) * scale + rng.uniform(0, 4000, (n, 1))).astype(np.float32)
        if rep == 6:      # degenerate scales: the squared length underflows (the fast route must hand over below 1e-30)
            xs = (10.0 ** rng.uniform(-14.0, -2.0, n)).astype(np.float32)
            ys = (10.0 ** rng.uniform(-14.0, -2.0, n)).astype(np.float32)
            h = (rng.standard_normal((n, 4)) * (10.0 ** rng.uniform(-16.0, -3.0, (n, 1)))).astype(np.float32)
        if rep == 1:      # adversarial: choose hR so that component x lands (in f64) right on a code boundary
            k = rng.integers(1, 255, n)
            nxt = (k - 127.5) / 127.5                                  # n_x with 127.5 n + 128 = k + 0.5... a boundary of floor(t)
            nxt = (k + rng.choice([-1e-6, 0.0, 1e-6], n) - 128.0) / 127.5
            # n_x = -2y dz / sqrt((2y dz)^2 + (2x dzy)^2 + (4xy)^2); with dzy = 0: dz = -n_x * 2x / sqrt(1 - n_x^2)
            dz = -nxt * (2.0 * xs.astype(np.float64)) / np.sqrt(np.maximum(1e-12, 1.0 - nxt * nxt))
            h[:, 0] = h[:, 3]                                          # hT = hB
            h[:, 2] = (h[:, 1].astype(np.float64) + dz).astype(np.float32)
        if rep == 5:
            h[::7, 0] = np.inf
            h[3::11, 2] = np.nan
            h[5::13, 1] = -np.inf
            h[::17] = 3.0e38
        nf = C.c_uint64()
        bad = L.emul_normal_fast_check(emul._p(xs), emul._p(ys), emul._p(np.ascontiguousarray(h)), n, C.byref(nf))
        assert bad == 0, f"rep {rep}: {bad} texels differ"
        total_fast += nf.value
        if rep == 0:
            assert nf.value > 0.99 * n          # ordinary terrain: the fast route carries almost everything
        if rep == 1:
            assert nf.value < 0.8 * n           # on the boundaries the guard hands over
    assert total_fast > 10_000_000


@pytest.mark.parametrize("fmt", [1, 4])
def test_overlay_pass_header_code_matches_oracle(orc, fmt):
    """The overlay pass (line_shader.wgsl over the post pass) through the product's header functions --
    keys raised by max, then coloured -- against the oracle's in-order z-buffer with Greater."""
    import ctypes as C
    from scenes import overlay_geometry
    L = emul.lib()
    L.emul_overlay_lines.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    for W, H, seed, width in ((160, 96, 1, 0.5), (97, 61, 2, 1.5), (64, 64, 3, 0.0)):
        v, ix = overlay_geometry(W, H, seed=seed)
        base = np.random.default_rng(seed).integers(0, 255, (H, W, 4), dtype=np.uint8)
        a, b = base.copy(), base.copy()
        orc.OracleRenderer(W, H, color_format=fmt).overlay_lines(v, ix, a, width)
        L.emul_overlay_lines(emul._p(v), len(v), emul._p(ix), ix.size, width, W, H, 1 if fmt in (1, 2) else 0, 1 if fmt in (2, 4) else 0, emul._p(b), W * 4)
        assert np.array_equal(a, b), f"{np.argwhere((a != b).any(axis=-1))[:5]}"
        assert (a != base).any(axis=-1).mean() > 0.05


@pytest.mark.parametrize("fmt", [1, 2, 3, 4])
def test_text_overlay_header_code_matches_oracle(orc, fmt):
    """The text overlay (glyphon's glyph quads: text_renderer.rs:198-204, :259-291) through the product's header function
    glyph_blend with the kernels' ownership rule, against the oracle's in-order pass with Greater, in every surface format."""
    import ctypes as C
    from scenes import glyph_scene
    L = emul.lib()
    L.emul_overlay_glyphs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    for W, H, seed in ((200, 120, 1), (97, 61, 2)):
        glyphs, atlas = glyph_scene(W, H, seed=seed)
        base = np.random.default_rng(seed).integers(0, 256, (H, W, 4), dtype=np.uint8)
        a, b = base.copy(), base.copy()
        orc.OracleRenderer(W, H, color_format=fmt).overlay_glyphs(glyphs, atlas, a)
        L.emul_overlay_glyphs(emul._p(glyphs), len(glyphs), emul._p(atlas), atlas.shape[1], atlas.shape[0], W, H, 1 if fmt in (1, 2) else 0, 1 if fmt in (2, 4) else 0,
                              emul._p(b), W * 4)
        assert np.array_equal(a, b), f"{np.argwhere((a != b).any(axis=-1))[:5]}"
        assert (a != base).any(axis=-1).mean() > 0.02


def test_pixelise_branch_header_code_matches_oracle(topo, orc):
    """The pixelise branch of the post pass (postprocessing_shader.wgsl:70-74) through the product's header functions
    (sample_pixelized, post_mix) against the oracle, and its sampler's two regimes on a picture where they can be told apart."""
    for (W, H, n) in ((96, 64, 12.0), (75, 41, 50.0), (64, 48, 3.0)):
        sc = Scene(24, 2, 2, eye_dh=300.0)
        e, o = emul.EmulRenderer(W, H, topo.terrain_uniforms), orc.OracleRenderer(W, H)
        sc.load(e)
        sc.load(o)
        u, pu = sc.uniforms(W, H, 25, 20, 70, 0), topo.post_uniforms(W, H, pixelize_n=n)
        e.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(e.render(), o.render(), f"pixelize_n {n}")


def test_pixelise_sampler_known_answers(orc):
    """The sampler choice of the pixelise branch on a frame whose render target is known: all sky.  Every sample -- Linear
    or Nearest -- of a constant image is that constant, so the frame equals the unpixelised one; and pixelize_n just under
    100 still takes the branch while 100 does not (the shader's 99.99999 threshold)."""
    from scenes import Scene
    import topo_renderer_amd as T
    sc = Scene(16, 1, 1, eye_dh=300.0)
    o = orc.OracleRenderer(48, 32)
    sc.load(o)
    up = sc.uniforms(48, 32, 0.0, -80.0, 40.0, 1)
    o.update(48, 32, up, T.post_uniforms(48, 32))
    ref = o.render()
    for n in (1.0, 10.0, 99.9999):
        o.update(48, 32, up, T.post_uniforms(48, 32, pixelize_n=n))
        assert_same_frame(o.render(), ref, f"all sky, pixelize_n {n}")
    with pytest.raises(RuntimeError):
        o.update(48, 32, up, T.post_uniforms(48, 32, pixelize_n=0.25))
