"""GPU parity: the HIP path through the C ABI (libtopo_hip.so) against the CPU oracle, bit for bit.

Depth is compared as raw f32 bits (north_star tolerance: 1 ULP -- we hold 0), colour as RGBA8 bytes
(tolerance: 1 LSB -- we hold 0)."""
import math
import os

import numpy as np
import pytest

from scenes import Scene, assert_same_frame

pytestmark = pytest.mark.gpu


def both(topo, orc, W, H):
    return topo.TerrainRenderer(W, H), orc.OracleRenderer(W, H)


def test_device_sincos_matches_spec(topo, orc):
    r = topo.TerrainRenderer(8, 8)
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-3.2, 3.2, 200000), np.linspace(-7, 7, 50001), [0.0, -0.0, math.pi / 4, -math.pi / 4]]).astype(np.float32)
    s, c = r.probe_sincos(x)
    so, co = orc.sincos(x)
    assert np.array_equal(s.view(np.uint32), so.view(np.uint32))
    assert np.array_equal(c.view(np.uint32), co.view(np.uint32))
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7


def test_division_probe(topo):
    """The device's division forms (topo_math.h: div_f, div_const) against the IEEE quotient numpy computes."""
    r = topo.TerrainRenderer(8, 8)
    rng = np.random.default_rng(7)
    n = 2_000_000
    mant = lambda: rng.uniform(1.0, 2.0, n).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)
    # general form: operands and quotients well inside 2^-96 .. 2^96 (the range this path uses is far narrower)
    x = mant() * np.exp2(rng.integers(-40, 41, n)).astype(np.float32)
    y = mant() * np.exp2(rng.integers(-40, 41, n)).astype(np.float32)
    x[:1000] = 1.0                                        # reciprocals
    x[1000:2000] = 0.0
    y[2000:3000] = np.arange(1, 1001, dtype=np.float32)   # small integers (doubled areas)
    q = r.probe_div(0, x, y)
    with np.errstate(all="ignore"):
        ref = x / y
    assert np.array_equal(q, ref)                         # == : the sign of a zero quotient is not preserved
    assert np.array_equal(q[q != 0].view(np.uint32), ref[ref != 0].view(np.uint32))
    # same divisor shared by several numerators, the perspective and area reciprocals
    w = rng.uniform(50.0, 1.0e6, n).astype(np.float32)
    assert np.array_equal(r.probe_div(0, np.ones_like(w), w).view(np.uint32), (np.float32(1.0) / w).view(np.uint32))
    # square root (normalize): the hardware estimate corrected to the IEEE result
    sq = np.concatenate([rng.uniform(0.0, 4.0, n), np.exp2(rng.uniform(-20, 50, n)), [0.0, 1.0, 4.0, 2.0]]).astype(np.float32)
    assert np.array_equal(r.probe_div(3, sq, sq).view(np.uint32), np.sqrt(sq).view(np.uint32))
    # constants: x / 255 (dither) and x / 0.1 (contour), any magnitude the path can produce
    xs = np.concatenate([rng.uniform(-2.0, 2.0, n), rng.uniform(-1e-6, 1e-6, 1000), [0.0, 1.0, -1.0, 2.0 ** -24]]).astype(np.float32)
    for kind, c in ((1, np.float32(255.0)), (2, np.float32(0.15) - np.float32(0.05))):
        got = r.probe_div(kind, xs, xs)
        assert np.array_equal(got, xs / c)
        nz = got != 0
        assert np.array_equal(got[nz].view(np.uint32), (xs / c)[nz].view(np.uint32))


def _hash12n_f32(sx, sy):
    """render_shader.wgsl:75-79 in numpy binary32, one rounding per operation (fract(e) = e - floor(e))."""
    f = np.float32
    fract = lambda v: v - np.floor(v)
    px, py = fract(sx * f(5.3987)), fract(sy * f(5.4421))
    d = py * (px + f(21.5351)) + px * (py + f(14.3137))
    px, py = px + d, py + d
    return fract(px * py * f(95.4307))


def test_fract_probe(topo):
    """The dither's fractions on the device: v_fract_f32 is x - floor(x) except for the negatives above -2^-24, and the
    wave-level window test of shade_fragment (topo_math.h) keeps those on the two-instruction form."""
    r = topo.TerrainRenderer(8, 8)
    rng = np.random.default_rng(11)
    f = np.float32
    n = 1 << 20
    x = np.concatenate([rng.uniform(-4.0e6, 4.0e6, n), rng.uniform(-2.0, 2.0, n), -np.exp2(rng.uniform(-60.0, -20.0, n)),
                        np.exp2(rng.uniform(-60.0, 30.0, n)), [0.0, -0.0, -1.0, 1.0, -2.0 ** -24, -2.0 ** -25, np.inf, -np.inf]]).astype(f)
    with np.errstate(all="ignore"):
        ref = x - np.floor(x)
    same = lambda a, b: (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same(r.probe_div(5, x, x), ref).all()
    instr = r.probe_div(4, x, x)
    differ = ~same(instr, ref)
    assert differ.any() and ((x[differ] < 0) & (x[differ] > -f(2.0) ** -24)).all()
    assert (instr[differ] == np.nextafter(f(1.0), f(0.0))).all() and (ref[differ] == 1.0).all()
    # the dither itself, channel by channel, against the numpy restatement: arguments all over the range the path produces and
    # clustered on the twelve first-level zeros p = -c (a few ulps either side), alone and mixed into waves of ordinary pixels
    offs = [f(0.0), f(0.07), f(0.11), f(0.13), f(0.13) + f(0.07), f(0.13) + f(0.11)]
    m = 1 << 18
    px, py = rng.uniform(-5.0e5, 5.0e5, m).astype(f), rng.uniform(-5.0e5, 5.0e5, m).astype(f)
    near = np.array([-c + f(k) * np.spacing(c) for c in offs[1:] for k in range(-8, 9)] + [-(2.0 ** -k) for k in range(20, 60)], dtype=f)
    chains = [near, near + f(0.07), near + f(0.11), near + f(0.13), (near + f(0.13)) + f(0.07), (near + f(0.13)) + f(0.11)]
    risky = [((v * f(5.3987) < 0) & (v * f(5.3987) > -f(2.0) ** -24)).sum() for v in chains]
    assert all(risky[k] > 0 for k in (0, 1, 2, 4)), risky      # (the sample reaches the instruction's exceptional range through every chain that can:
    #                                                               p + 0.13 near zero is a multiple of 1.5e-8, times 5.4 beyond 2^-24)
    near = np.concatenate([near, -near, near * f(0.5)])
    sel = rng.integers(0, m, near.size * 2)
    px[sel[:near.size]] = near                     # one risky pixel inside otherwise ordinary waves
    py[sel[near.size:]] = near
    px = np.concatenate([px, np.tile(near, 8)])    # and whole waves of them
    py = np.concatenate([py, rng.uniform(-3.0, 3.0, near.size * 8).astype(f)])
    qx, qy = px + f(0.13), py + f(0.13)
    with np.errstate(all="ignore"):
        for k, off in enumerate((f(0.0), f(0.07), f(0.11))):
            h1 = _hash12n_f32(px + off, py + off) if k else _hash12n_f32(px, py)
            h2 = _hash12n_f32(qx + off, qy + off) if k else _hash12n_f32(qx, qy)
            want = f(0.01) + f(0.7) * (f(0.25) / f(0.7)) + f(1.0) * (h1 + h2 - f(1.0)) / f(255.0)
            got = r.probe_div(6 + k, px, py)
            bad = ~same(got, want.astype(f))
            assert not bad.any(), (k, int(bad.sum()), px[bad][:4], py[bad][:4], got[bad][:4], want[bad][:4])


@pytest.mark.parametrize("tile,n_lat,n_lon", [(64, 1, 1), (48, 2, 2), (33, 3, 3), (150, 1, 2)])
def test_normals_byte_exact(topo, orc, tile, n_lat, n_lon):
    sc = Scene(tile, n_lat, n_lon)
    g, o = both(topo, orc, 16, 16)
    sc.load(g)
    sc.load(o)
    for loc in sc.locs:
        a, b = g.read_normals(*loc), o.read_normals(loc[0], loc[1], tile, tile)
        assert np.array_equal(a, b), f"tile {loc}: {np.argwhere((a != b).any(axis=-1))[:4]}"


def test_normals_insertion_order_and_unload(topo, orc):
    sc = Scene(40, 2, 2)
    order = [sc.locs[3], sc.locs[0], sc.locs[2], sc.locs[1]]
    g, o = both(topo, orc, 16, 16)
    sc.load(g, order)
    sc.load(o, order)
    g.unload_terrain(*order[1])
    o.unload_terrain(*order[1])
    g.add_terrain(order[1][0], order[1][1], sc.heights[order[1]], *sc.transform(order[1]))
    o.add_terrain(order[1][0], order[1][1], sc.heights[order[1]], *sc.transform(order[1]))
    for loc in sc.locs:
        assert np.array_equal(g.read_normals(*loc), o.read_normals(loc[0], loc[1], 40, 40))
    g.recompute_normals()   # replay of the load phase must not change the state when nothing was unloaded since
    g2, _ = both(topo, orc, 16, 16)
    sc.load(g2, order)
    g2.recompute_normals()
    o2 = orc.OracleRenderer(16, 16)
    sc.load(o2, order)
    for loc in sc.locs:
        assert np.array_equal(g2.read_normals(*loc), o2.read_normals(loc[0], loc[1], 40, 40))


FRAMES = [
    # tile, n_lat, n_lon, W, H, yaw, pitch, fov, mode, eye_dh
    (64, 1, 1, 128, 64, 0, 0, 60, 0, 50),
    (64, 1, 1, 128, 64, 0, 0, 60, 1, 50),
    (64, 1, 1, 128, 64, 0, 0, 60, 2, 50),
    (64, 2, 2, 128, 64, 30, 10, 60, 0, 50),
    (48, 3, 3, 160, 96, 200, 30, 79.28, 0, 50),
    (128, 2, 2, 256, 128, 123, -5, 45, 0, 50),
    (64, 2, 2, 128, 128, 77, 60, 100, 0, 100),     # big near-field triangles, near-plane clipping
    (64, 2, 2, 128, 128, 77, 85, 100, 1, 400),
    (256, 1, 1, 300, 200, 10, 45, 90, 0, 20),
    (200, 1, 2, 257, 131, 300, 20, 120, 2, 80),     # odd sizes
]


@pytest.mark.parametrize("cfg", FRAMES, ids=[f"f{i}" for i in range(len(FRAMES))])
def test_frame_bit_exact(topo, orc, cfg):
    tile, n_lat, n_lon, W, H, yaw, pitch, fov, mode, dh = cfg
    sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    assert_same_frame(g.render(), o.render(), f"frame {cfg}")
    assert (g.counters()["status"] & 1) == 0


def test_depth_pitch_pad_256(topo, orc):
    sc = Scene(64, 1, 1)
    W, H = 100, 40
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(W, H, 45, 20, 70), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    rgba, depth = g.render(padded_depth=True)
    assert depth.shape[1] * 4 == topo.pad_256(4 * W) == 512
    ro, do = o.render()
    assert np.array_equal(depth[:, :W].view(np.uint32), do.view(np.uint32))
    assert np.array_equal(rgba, ro)


def test_empty_scene_is_sky(topo, orc):
    g, o = both(topo, orc, 64, 32)
    sc = Scene(16, 1, 1)
    u, pu = sc.uniforms(64, 32), topo.post_uniforms(64, 32)
    g.update(64, 32, u, pu)
    o.update(64, 32, u, pu)
    assert_same_frame(g.render(), o.render(), "empty scene")
    g.add_terrain(45, 15, sc.heights[(45, 15)], *sc.transform((45, 15)))
    g.unload_terrain(45, 15)
    assert_same_frame(g.render(), o.render(), "after unload")


def test_errors_mirror_reference_limits(topo):
    g = topo.TerrainRenderer(32, 32)
    sc = Scene(16, 1, 1)
    g.add_terrain(45, 15, sc.heights[(45, 15)], *sc.transform((45, 15)))
    with pytest.raises(topo.TopoError) as e:
        g.add_terrain(45, 16, np.zeros((20, 20), np.float32), *sc.transform((45, 16)))
    assert e.value.code == topo.TOPO_ERR_INVALID
    with pytest.raises(topo.TopoError) as e:
        g.update(32, 32, sc.uniforms(32, 32), topo.post_uniforms(32, 32, pixelize_n=0.5))
    assert e.value.code == topo.TOPO_ERR_INVALID


@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_pixelise_branch_of_the_post_pass(topo, orc, fmt):
    """postprocessing_shader.wgsl:70-74 (never taken by the reference, which pins pixelize_n to 100): the colour sampled at
    floor(uv n) / n through the render target's sampler -- Linear inside a block of equal uv, Nearest where the pixel's quad
    straddles two blocks -- then the same contour mix; depth unchanged.  Several block counts, an odd-sized target, a target
    that is not the viewport's size, with and without a depth output, and back to 100."""
    import torch
    sc = Scene(64, 2, 2, eye_dh=120.0)
    for (W, H, n) in ((200, 136, 50.0), (333, 129, 7.5), (64, 64, 99.0), (128, 96, 1.0)):
        g, o = topo.TerrainRenderer(W, H, color_format=fmt), orc.OracleRenderer(W, H, color_format=fmt)
        sc.load(g)
        sc.load(o)
        u = sc.uniforms(W, H, 30, 12, 75, 0)
        plain = topo.post_uniforms(W, H)
        g.update(W, H, u, plain)
        ref_plain = g.render()
        pu = topo.post_uniforms(W, H, pixelize_n=n)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        got, want = g.render(), o.render()
        assert_same_frame(got, want, f"pixelize_n {n} {W}x{H} format {fmt}")
        assert np.array_equal(got[1], ref_plain[1]) and (got[0] != ref_plain[0]).any(axis=-1).mean() > 0.05      # same depth, another picture
        rgba_only, _ = g.render(want_depth=False)
        assert np.array_equal(rgba_only, want[0])
        img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        g.render_device(img.data_ptr(), W * 4)
        torch.cuda.synchronize()
        assert np.array_equal(img.cpu().numpy(), want[0])
        g.set_stream(0)
        g.update(W, H, u, plain)
        assert_same_frame(g.render(), ref_plain, "back to pixelize_n 100")


def test_panorama_views_match_per_sector_frames(topo, orc):
    import torch
    sc = Scene(96, 2, 2)
    sw, sh = 64, 128
    g, o = both(topo, orc, sw, sh)
    sc.load(g)
    sc.load(o)
    us = sc.panorama(sw, sh, yaw0_deg=11.0)
    strip = torch.zeros((sh, 8 * sw, 4), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((sh, 8 * sw), dtype=torch.float32, device="cuda")
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.render_views_device(us, sw, sh, strip.data_ptr(), 4 * sw, 4 * 8 * sw, depth.data_ptr(), 4 * sw, 4 * 8 * sw)
    torch.cuda.synchronize()
    strip, depth = strip.cpu().numpy(), depth.cpu().numpy()
    o.update(sw, sh, us[0], topo.post_uniforms(sw, sh))
    ro, do = o.render_views(us, threads=4)
    for k in range(8):
        assert_same_frame((strip[:, k * sw:(k + 1) * sw], depth[:, k * sw:(k + 1) * sw]), (ro[k], do[k]), f"sector {k}")


@pytest.mark.parametrize("name", ["single_64", "block2x2_24", "nearfield_16"])
def test_gpu_reproduces_golden(topo, name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    tile, n_lat, n_lon, W, H, yaw, pitch, fov, dh = g["params"]
    tile, W, H = int(tile), int(W), int(H)
    sc = Scene(tile, int(n_lat), int(n_lon), eye_dh=float(dh))
    r = topo.TerrainRenderer(W, H)
    sc.load(r)
    for i, loc in enumerate(sc.locs):
        assert np.array_equal(r.read_normals(*loc), g[f"normals_{i}"])
    for mode in (0, 1, 2):
        r.update(W, H, g[f"uniforms_{mode}"], topo.post_uniforms(W, H))      # the committed uniform block, not recomputed
        rgba, depth = r.render()
        assert np.array_equal(rgba, g[f"rgba_{mode}"])
        if mode == 0:
            assert np.array_equal(depth.view(np.uint32), g["depth_bits"])


def _strip(topo, r, views, sw, sh):
    import torch
    n = len(views)
    rgba = torch.zeros((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((n, sh, sw), dtype=torch.float32, device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.render_views_device(views, sw, sh, rgba.data_ptr(), sh * sw * 4, sw * 4, depth.data_ptr(), sh * sw * 4, sw * 4)
    torch.cuda.synchronize()
    return rgba.cpu().numpy(), depth.cpu().numpy()


def test_full_size_config2_against_oracle_and_properties(topo, orc):
    # BASELINE config 2: one 1200x1200 COP90-shaped tile, 4096x1024 panorama (8 sectors of 512x1024), f32
    sc = Scene(1200, 1, 1, lat0=40, lon0=10, vfrac=(0.623, 0.717))
    sw, sh = 512, 1024
    r = topo.TerrainRenderer(sw, sh)
    sc.load(r)
    views = sc.panorama(sw, sh)
    rgba, depth = _strip(topo, r, views, sw, sh)
    # idempotence + independence from how the sectors are batched (the multi-GPU sharding property)
    rgba2, depth2 = _strip(topo, r, views, sw, sh)
    assert np.array_equal(rgba, rgba2) and np.array_equal(depth.view(np.uint32), depth2.view(np.uint32))
    for lo, hi in ((0, 4), (4, 8), (2, 3)):
        ra, da = _strip(topo, r, views[lo:hi], sw, sh)
        assert np.array_equal(ra, rgba[lo:hi]) and np.array_equal(da.view(np.uint32), depth[lo:hi].view(np.uint32))
    # every sector is a complete reference frame: compare all 8 with the oracle
    o = orc.OracleRenderer(sw, sh)
    sc.load(o)
    o.update(sw, sh, views[0], topo.post_uniforms(sw, sh))
    ro, do = o.render_views(views, threads=8)
    for k in range(8):
        assert_same_frame((rgba[k], depth[k]), (ro[k], do[k]), f"config-2 sector {k}")
    assert np.array_equal(r.read_normals(40, 10), o.read_normals(40, 10, 1200, 1200))
    # sanity of the picture itself: sky on top, terrain below, depth in [0,1]
    assert (depth[:, 0, :] == 1.0).all() and (depth[:, -1, :] < 1.0).mean() > 0.9
    assert depth.min() >= 0.0 and depth.max() <= 1.0
    assert (r.counters()["status"] & 1) == 0


def test_big_triangle_queue_and_clipping_paths_are_exercised(topo, orc):
    # a coarse mesh seen from close up: every triangle is large; many cross the near plane
    sc = Scene(12, 2, 2, eye_dh=60.0)
    W, H = 640, 480
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    big = []
    for yaw, pitch in ((10, 35), (200, 80), (100, 5)):
        u, pu = sc.uniforms(W, H, yaw, pitch, 110, 0), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(g.render(), o.render(), f"coarse mesh yaw {yaw} pitch {pitch}")
        big.append(g.counters()["big_items"])
    assert max(big) > 50, big


def test_peak_visibility_against_depth(topo, orc):
    # get_visible_labels over the device-resident depth (render_engine.rs:338-396)
    from scenes import random_peaks
    sc = Scene(96, 2, 2, eye_dh=300.0)
    W, H = 333, 200
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    peaks = random_peaks(sc, 2000)
    total = 0
    for yaw, pitch in ((0, 20), (120, 35), (250, 10), (40, 0)):
        u, pu = sc.uniforms(W, H, yaw, pitch, 90, 0), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(g.render(), o.render(), "peaks frame")
        vg, xg = g.visible_peaks(peaks)
        vo, xo = o.visible_peaks(peaks)
        assert np.array_equal(vg, vo) and np.array_equal(xg, xo)
        total += int(vo.sum())
    assert 20 < total < 4 * len(peaks) - 20
    g.render(want_depth=False)
    with pytest.raises(topo.TopoError):
        g.visible_peaks(peaks)                        # no depth was produced by the last render


def test_queue_overflow_paths(topo, orc):
    # big-triangle queue full -> the producers rasterise in place: slower, still exact, status bit 0 set;
    # rare-triangle queue full -> triangles would be dropped: topo_render must not hand that frame out.
    sc = Scene(12, 2, 2, eye_dh=60.0)
    W, H = 320, 240
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(W, H, 10, 35, 110, 0), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    ref = o.render()
    g.debug_set_queue_caps(16, 0)
    assert_same_frame(g.render(), ref, "big queue overflow")
    assert g.counters()["status"] & 1
    g.debug_set_queue_caps(0, 2)                      # an explicit capacity is not grown: the call fails loudly
    with pytest.raises(topo.TopoError) as e:
        g.render()
    assert e.value.code == topo.TOPO_ERR_CAPACITY
    g.debug_set_queue_caps(0, 2 | 0x80000000)         # a growable queue (the default, from a tiny start): re-rendered, exact
    assert_same_frame(g.render(), ref, "rare queue grown on demand")
    assert g.counters()["status"] == 0 and g.counters()["rare_items"] > 2
    g.debug_set_queue_caps(0, 0)
    assert_same_frame(g.render(), ref, "defaults restored")
    assert g.counters()["status"] == 0
    # a larger target, where triangles span 24 regions and more: their region items are written by the whole wave (k_raster_rare),
    # and a full queue sends such a triangle back to its lane -- all of it, or the part of a reservation that ran over the end
    W2, H2 = 640, 480
    g2, o2 = both(topo, orc, W2, H2)
    sc.load(g2)
    sc.load(o2)
    u2, pu2 = sc.uniforms(W2, H2, 10, 35, 110, 0), topo.post_uniforms(W2, H2)
    g2.update(W2, H2, u2, pu2)
    o2.update(W2, H2, u2, pu2)
    ref2 = o2.render()
    assert_same_frame(g2.render(), ref2, "wave-written region items")
    n_items = g2.counters()["big_items"]
    assert n_items > 50, n_items
    for cap in (16, 40, n_items // 2, n_items - 3):
        g2.debug_set_queue_caps(cap, 0)
        assert_same_frame(g2.render(), ref2, f"big queue of {cap} with wave-written items")
        assert g2.counters()["status"] & 1


def test_overflow_status_is_per_frame_on_the_async_paths(topo, orc):
    """topo_render_device / topo_render_views_device return before the frame exists: an overflowed (incomplete) frame is
    reported by the call that waits for it -- once -- and a complete frame after it is reported as complete."""
    import torch
    sc = Scene(12, 2, 2, eye_dh=60.0)
    W, H = 320, 240
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    down = sc.uniforms(W, H, 10, 35, 110, 0)                                             # near-field giants: hundreds of rare triangles
    high = topo.geometry_transform(sc.ground + 250000.0, sc.vlon, sc.vlat)               # from 250 km up: nothing near, nothing clipped
    up = topo.camera_uniforms(high, 0.3, np.radians(80.0), np.radians(60.0), W, H, sc.vlon, sc.vlat, 0)
    rgba = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    pu = topo.post_uniforms(W, H)

    def frame(u):
        g.update(W, H, u, pu)
        g.render_device(rgba.data_ptr(), W * 4, depth.data_ptr(), W * 4)

    g.debug_set_queue_caps(0, 2)
    frame(down)
    with pytest.raises(topo.TopoError) as e:
        g.join()
    assert e.value.code == topo.TOPO_ERR_CAPACITY
    g.join()                                           # reported once
    assert g.frame_status()["rare_overflow"]
    frame(up)                                          # no rare triangles from up there: complete although the cap is still 2
    g.synchronize()
    st = g.frame_status()
    assert not st["rare_overflow"] and st["status"] == 0
    o.update(W, H, up, pu)
    ro, do = o.render()
    assert np.array_equal(rgba.cpu().numpy(), ro) and np.array_equal(depth.cpu().numpy().view(np.uint32), do.view(np.uint32))
    for depth_frames in (1, 2):                        # the same through the frames-in-flight path
        g.set_pipeline_depth(depth_frames)
        frame(down)
        frame(up)
        with pytest.raises(topo.TopoError):
            g.join()
        frame(up)
        g.join()
    g.set_pipeline_depth(1)
    g.debug_set_queue_caps(0, 0)
    frame(down)
    g.synchronize()
    o.update(W, H, down, pu)
    ro, do = o.render()
    assert np.array_equal(rgba.cpu().numpy(), ro) and np.array_equal(depth.cpu().numpy().view(np.uint32), do.view(np.uint32))


@pytest.mark.parametrize("tw,th", [(40, 30), (30, 40), (3, 3), (61, 16), (62, 17)])
def test_non_square_and_tiny_tiles(topo, orc, tw, th):
    # the seam shaders guard with dimensions.x while indexing rows (edge_shader.wgsl:33); block edges at 60/15 cells
    import math
    W, H = 96, 64
    g, o = both(topo, orc, W, H)
    hts = {}
    for (la, lo) in topo.synth.mosaic_locations(45, 15, 2, 2):
        h = topo.synth_tile(la, lo, max(tw, th), max(tw, th))[:th, :tw].copy()
        tr = (np.float32([0, 0]), np.float32([lo, la + 1]), np.float32([1.0 / tw, 1.0 / th]))
        hts[(la, lo)] = h
        g.add_terrain(la, lo, h, *tr)
        o.add_terrain(la, lo, h, *tr)
    for loc in hts:
        assert np.array_equal(g.read_normals(*loc), o.read_normals(loc[0], loc[1], tw, th)), loc
    eye = topo.geometry_transform(float(hts[(46, 16)].max()) + 900.0, 16.02, 46.03)
    for yaw, pitch in ((20, 25), (200, 60)):
        u = topo.camera_uniforms(eye, math.radians(yaw), math.radians(pitch), math.radians(90), W, H, 16.0, 46.0, 0)
        g.update(W, H, u, topo.post_uniforms(W, H))
        o.update(W, H, u, topo.post_uniforms(W, H))
        assert_same_frame(g.render(), o.render(), f"{tw}x{th} tiles yaw {yaw}")


@pytest.mark.parametrize("tw,th", [(240, 47), (480, 31), (240, 46), (240, 16), (720, 15)])
def test_one_pass_load_path_equals_the_separate_kernels(topo, orc, tw, th):
    """Tiles 240 k columns wide take the load path that reads the DEM once (k_trig_tables -> k_normals_rolling<4, 4, true>:
    normals + block minima / maxima -> k_block_bounds).  Its normals are the oracle's, and every table it leaves -- block
    min/max, sin/cos tables, f64 cull bounds -- is bit for bit what k_block_tables + the LDS-tile normals kernel leave."""
    import math
    W, H = 96, 64
    locs = topo.synth.mosaic_locations(45, 15, 2, 2)
    hts = {}
    for (la, lo) in locs:
        h = topo.synth_tile(la, lo, max(tw, th), max(tw, th))[:th, :tw].copy()
        h[th // 3, tw // 5] = -12.5                     # a negative height, zeros of both signs and a spike on a block edge
        h[0, 0], h[th - 1, tw - 1] = 0.0, -0.0
        h[min(15, th - 1), 60] = 8000.0
        hts[(la, lo)] = h
    tr = lambda la, lo: (np.float32([0, 0]), np.float32([lo, la + 1]), np.float32([1.0 / tw, 1.0 / th]))
    one, sep, o = topo.TerrainRenderer(W, H), topo.TerrainRenderer(W, H), orc.OracleRenderer(W, H)
    sep.set_normals_lds_rows(8)                         # the LDS-tile kernel + k_block_tables
    for (la, lo) in locs:
        for r in (one, sep, o):
            r.add_terrain(la, lo, hts[(la, lo)], *tr(la, lo))
    for loc in locs:
        n1 = one.read_normals(*loc)
        assert np.array_equal(n1, o.read_normals(loc[0], loc[1], tw, th)), loc
        assert np.array_equal(n1, sep.read_normals(*loc)), loc
        a, b = one.read_tile_tables(*loc), sep.read_tile_tables(*loc)
        assert a["minmax"].shape[0] == ((tw - 1 + 59) // 60) * ((th - 1 + 14) // 15)
        for k in ("minmax", "trig", "bounds"):
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (loc, k, np.argwhere(a[k] != b[k])[:4])
        # the minima / maxima against numpy on the blocks' vertex ranges
        bxc = (tw - 1 + 59) // 60
        for blk, (lo_, hi_) in enumerate(a["minmax"]):
            by, bx = divmod(blk, bxc)
            v = hts[loc][15 * by:min(15 * by + 16, th), 60 * bx:min(60 * bx + 61, tw)]
            assert lo_ == v.min() and hi_ == v.max(), (loc, blk)
    one.recompute_normals()                             # the batched form (all tiles in one launch of each kernel)
    for loc in locs:
        assert np.array_equal(one.read_normals(*loc), sep.read_normals(*loc)), loc
        a, b = one.read_tile_tables(*loc), sep.read_tile_tables(*loc)
        for k in ("minmax", "trig", "bounds"):
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (loc, k)
    eye = topo.geometry_transform(float(hts[(46, 16)].max()) + 900.0, 16.02, 46.03)
    u = topo.camera_uniforms(eye, math.radians(20), math.radians(25), math.radians(90), W, H, 16.0, 46.0, 0)
    for r in (one, sep, o):
        r.update(W, H, u, topo.post_uniforms(W, H))
    f1 = one.render()
    assert_same_frame(f1, o.render(), f"{tw}x{th} tiles")
    assert_same_frame(f1, sep.render(), f"{tw}x{th} tiles, separate kernels")
    assert one.counters() == sep.counters()


def test_hemispheres_replacement_and_draw_order(topo, orc):
    # tiles around (0, 0): GeoLocation::from_coord maps 0 to S / W, BTreeMap order is (|lat|, dir, |lon|, dir)
    import math
    tile, W, H = 20, 128, 96
    g, o = both(topo, orc, W, H)
    locs = [(0, 0), (-1, 0), (0, -1), (-1, -1), (1, 1), (1, -1)]
    for (la, lo) in locs:
        h = topo.synth_tile(la + 45, lo + 15, tile, tile) + np.float32(100.0 * (la + 2))
        tr = topo.synth.tile_transform(la, lo, tile, tile)
        g.add_terrain(la, lo, h, *tr)
        o.add_terrain(la, lo, h, *tr)
    # replace one tile (BTreeMap::insert on an existing key) and unload another
    h2 = topo.synth_tile(50, 20, tile, tile)
    g.add_terrain(0, 0, h2, *topo.synth.tile_transform(0, 0, tile, tile))
    o.add_terrain(0, 0, h2, *topo.synth.tile_transform(0, 0, tile, tile))
    g.unload_terrain(1, -1)
    o.unload_terrain(1, -1)
    g.unload_terrain(7, 7)          # removing a key that is not there is a no-op
    for (la, lo) in locs:
        if (la, lo) != (1, -1):
            assert np.array_equal(g.read_normals(la, lo), o.read_normals(la, lo, tile, tile)), (la, lo)
    eye = topo.geometry_transform(40000.0, 0.2, 0.1)
    for yaw, pitch in ((0, 60), (140, 75), (270, 50)):
        u = topo.camera_uniforms(eye, math.radians(yaw), math.radians(pitch), math.radians(100), W, H, 0.0, 0.0, 0)
        g.update(W, H, u, topo.post_uniforms(W, H))
        o.update(W, H, u, topo.post_uniforms(W, H))
        fr = g.render()
        assert_same_frame(fr, o.render(), f"hemispheres yaw {yaw}")
        assert (fr[1] < 1).mean() > 0.2


def test_eye_below_the_surface_and_far_above(topo, orc):
    sc = Scene(64, 2, 2)
    W, H = 160, 100
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    for dh, pitch in ((-300.0, 10.0), (-300.0, -40.0), (250000.0, 89.0), (1.0e6, 89.9)):
        eye = topo.geometry_transform(sc.ground + dh, sc.vlon, sc.vlat)
        u = topo.camera_uniforms(eye, 0.7, np.radians(pitch), np.radians(60), W, H, sc.vlon, sc.vlat, 0)
        g.update(W, H, u, topo.post_uniforms(W, H))
        o.update(W, H, u, topo.post_uniforms(W, H))
        assert_same_frame(g.render(), o.render(), f"eye dh {dh}")


def test_full_size_config4_sectors_against_oracle(topo, orc):
    # BASELINE config 4 inputs (10x10 degree mosaic = 100 tiles of 1200x1200, 2048x4096 sectors): two of the eight
    # sectors at full size against the oracle, plus invariance of the result to the occlusion filter
    import math
    deg, sw, sh, tile = 10, 2048, 4096, 1200
    locs = topo.synth.mosaic_locations(40, 10, deg, deg)
    g, o = topo.TerrainRenderer(sw, sh), orc.OracleRenderer(sw, sh)
    vlat, vlon = 40 + deg / 2 + 0.123, 10 + deg / 2 + 0.217
    ground = None
    for (la, lo) in locs:
        h = topo.synth_tile(la, lo, tile, tile)
        if (la, lo) == (int(math.floor(vlat)), int(math.floor(vlon))):
            ground = topo.synth.height_at(h, la, lo, vlon, vlat)
        tr = topo.synth.tile_transform(la, lo, tile, tile)
        g.add_terrain(la, lo, h, *tr)
        o.add_terrain(la, lo, h, *tr)
    eye = topo.geometry_transform(ground + 50.0, vlon, vlat)
    views = topo.panorama_uniforms(eye, 0.0, sw, sh, vlon, vlat, 0)
    pick = [views[1], views[6]]
    rg, dg = _strip(topo, g, pick, sw, sh)
    c_on = g.counters()
    g.set_occlusion_split(0.0)
    rg0, dg0 = _strip(topo, g, pick, sw, sh)
    c_off = g.counters()
    assert np.array_equal(rg, rg0) and np.array_equal(dg.view(np.uint32), dg0.view(np.uint32))
    assert c_on["far_tested"] > 10 * c_on["far_survived"] and c_off["far_tested"] == 0
    assert c_on["blocks_rastered"] < c_off["blocks_rastered"] // 2      # (with the filter on the count is in strips of 2 cell rows: up to 8 per block)
    o.update(sw, sh, pick[0], topo.post_uniforms(sw, sh))
    ro, do = o.render_views(pick, threads=2)
    for k in range(2):
        assert_same_frame((rg[k], dg[k]), (ro[k], do[k]), f"config-4 sector {k}")
    # ... and the WHOLE panorama, all eight sectors, against the oracle's bytes: their SHA-256 as committed by
    # tests/golden/make_c4_hashes.py (the oracle's 16 s per sector paid once, in the build container)
    import hashlib
    import json
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_sector_sha256.json")))
    assert want["sector_w"] == sw and want["sector_h"] == sh and want["n_tiles"] == len(locs)
    g.set_occlusion_split(90000.0)
    ra, da = _strip(topo, g, views, sw, sh)
    for k in range(8):
        assert hashlib.sha256(np.ascontiguousarray(ra[k]).tobytes()).hexdigest() == want["rgba"][k], f"sector {k}: colour bytes differ from the oracle's"
        assert hashlib.sha256(np.ascontiguousarray(da[k]).tobytes()).hexdigest() == want["depth"][k], f"sector {k}: depth bits differ from the oracle's"
    assert hashlib.sha256(np.ascontiguousarray(ro[0]).tobytes()).hexdigest() == want["rgba"][1]      # (the file is the oracle's: sector 1, rendered here)


def test_frame_sequence_on_one_renderer(topo, orc):
    """State carried between frames (the touched-segment marks of the visibility buffer, grow-only buffers, the
    queues): one renderer draws a sequence of frames of changing size, direction and scene -- terrain turning into
    sky and back, a smaller frame after a larger one, a width that is not a multiple of 64 px, a tile unloaded and
    re-added -- and every frame must equal the oracle's (same call history; its z-buffer starts from scratch each frame)."""
    import torch
    sc = Scene(64, 2, 2, eye_dh=60)
    g, o = both(topo, orc, 64, 64)
    sc.load(g)
    sc.load(o)
    seq = [  # (W, H, yaw, pitch, fov, mode, action)
        (256, 192, 20, 10, 70, 0, None),
        (256, 192, 20, -80, 70, 0, None),            # all sky: every mark of the previous frame must have been undone
        (256, 192, 200, 35, 70, 1, None),
        (96, 80, 200, 35, 70, 0, None),              # smaller frame inside the same buffers
        (333, 117, 75, 5, 100, 2, None),             # row length not a multiple of the 64-key segment
        (333, 117, 75, 5, 100, 0, ("unload", sc.locs[0])),
        (320, 256, 310, 15, 60, 0, ("add", sc.locs[0])),
        (320, 256, 310, -89, 60, 0, None),
        (320, 256, 130, 60, 120, 0, None),
    ]
    for i, (W, H, yaw, pitch, fov, mode, action) in enumerate(seq):
        if action:
            kind, loc = action
            for r in (g, o):
                if kind == "unload":
                    r.unload_terrain(*loc)
                else:
                    r.add_terrain(loc[0], loc[1], sc.heights[loc], *sc.transform(loc))
        u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(g.render(), o.render(), f"sequence frame {i}")
    # several views per submission after single-view frames, and back
    views = sc.panorama(96, 64, yaw0_deg=10)
    strip = torch.empty((8, 64, 96, 4), dtype=torch.uint8, device="cuda")
    depth = torch.empty((8, 64, 96), dtype=torch.float32, device="cuda")
    for _ in range(2):
        g.render_views_device(views, 96, 64, strip.data_ptr(), 64 * 96 * 4, 96 * 4, depth.data_ptr(), 64 * 96 * 4, 96 * 4)
        g.synchronize()
        for k in (0, 3, 7):
            o.update(96, 64, views[k], topo.post_uniforms(96, 64))
            assert_same_frame((strip[k].cpu().numpy(), depth[k].cpu().numpy()), o.render(), f"sequence panorama sector {k}")
        W, H = 128, 128
        u, pu = sc.uniforms(W, H, 45, -70, 60, 0), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(g.render(), o.render(), "sequence: single frame after a panorama")


def test_pipelined_frames_equal_serial_frames(topo, orc):
    """topo_set_pipeline_depth: frames kept in flight on their own streams give the bytes the one-at-a-time path gives
    (and the oracle's), also across tile changes and a switch back to depth 1."""
    import torch
    sc = Scene(64, 2, 2, eye_dh=60)
    g, o = both(topo, orc, 96, 64)
    sc.load(g)
    sc.load(o)
    W, H = 96, 64
    pans = [sc.panorama(W, H, yaw0_deg=y) for y in (0, 33, 170, 285, 90, 201)]
    bufs = [(torch.empty((8, H, W, 4), dtype=torch.uint8, device="cuda"), torch.empty((8, H, W), dtype=torch.float32, device="cuda"))
            for _ in pans]

    def check(k_list=(0, 5)):
        for views, (s, d) in zip(pans, bufs):
            for k in k_list:
                o.update(W, H, views[k], topo.post_uniforms(W, H))
                assert_same_frame((s[k].cpu().numpy(), d[k].cpu().numpy()), o.render(), "pipelined panorama")

    for depth in (3, 2, 1):
        g.set_pipeline_depth(depth)
        for b in bufs:
            b[0].zero_()
            b[1].zero_()
        torch.cuda.synchronize()
        for views, (s, d) in zip(pans, bufs):
            g.render_views_device(views, W, H, s.data_ptr(), H * W * 4, W * 4, d.data_ptr(), H * W * 4, W * 4)
        g.join()
        torch.cuda.synchronize()
        check()
    # tiles change while frames are in flight: add/unload join the pipeline first
    g.set_pipeline_depth(2)
    for views, (s, d) in zip(pans[:3], bufs[:3]):
        g.render_views_device(views, W, H, s.data_ptr(), H * W * 4, W * 4, d.data_ptr(), H * W * 4, W * 4)
    g.unload_terrain(*sc.locs[1])
    o.unload_terrain(*sc.locs[1])
    for views, (s, d) in zip(pans[3:], bufs[3:]):
        g.render_views_device(views, W, H, s.data_ptr(), H * W * 4, W * 4, d.data_ptr(), H * W * 4, W * 4)
    g.join()
    torch.cuda.synchronize()
    for views, (s, d) in zip(pans[3:], bufs[3:]):
        o.update(W, H, views[2], topo.post_uniforms(W, H))
        assert_same_frame((s[2].cpu().numpy(), d[2].cpu().numpy()), o.render(), "pipelined panorama after unload")
    # the host path still returns a finished frame
    u, pu = sc.uniforms(W, H, 20, 10, 70, 0), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    assert_same_frame(g.render(), o.render(), "topo_render with a pipeline depth of 2")
    with pytest.raises(topo.TopoError):
        g.set_pipeline_depth(9)


GEOTIFF_CASES = [
    dict(),                                                     # COP90 style: Deflate + floating-point predictor, one strip
    dict(byteorder=">"),
    dict(tile=(16, 16)),                                        # tiles, padded at the right / bottom edge
    dict(tile=(64, 32), byteorder=">", predictor=3),
    dict(rows_per_strip=7, predictor=3, compression="adobe_deflate_old"),
    dict(rows_per_strip=1, predictor=2),
    dict(predictor=2, byteorder=">", tile=(32, 16)),
    dict(compression="none", predictor=1),
    dict(compression="none", predictor=1, byteorder=">", rows_per_strip=3),
    dict(compression="packbits", predictor=1, rows_per_strip=5),
    dict(compression="packbits", predictor=3),
]


@pytest.mark.parametrize("kw", GEOTIFF_CASES, ids=[f"t{i}" for i in range(len(GEOTIFF_CASES))])
def test_geotiff_decode_bit_exact(topo, kw):
    """topo_geotiff_decode returns the float bits the encoder was given (tests/tiff_writer.py, independent of the decoder):
    specials and denormals included."""
    from tiff_writer import write_geotiff
    rng = np.random.default_rng(11)
    a = rng.normal(1200.0, 900.0, (45, 77)).astype(np.float32)
    a[0, :6] = [0.0, -0.0, np.inf, -np.inf, 1e-42, -32768.0]
    a[3, 5] = np.nan
    r = topo.TerrainRenderer(8, 8)
    got = r.decode_geotiff(write_geotiff(a, **kw))
    assert got.shape == a.shape and np.array_equal(got.view(np.uint32), a.view(np.uint32))


def test_geotiff_decode_of_libtiff_files(topo):
    """Files written by Pillow/libtiff (an encoder this repository does not control), incl. LZW and the predictors it offers."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image, TiffImagePlugin
    rng = np.random.default_rng(12)
    a = rng.normal(800.0, 500.0, (120, 131)).astype(np.float32)
    r = topo.TerrainRenderer(8, 8)
    n_ok = 0
    for comp in (None, "tiff_adobe_deflate", "tiff_lzw", "packbits"):
        for pred in (1, 3):
            ifd = TiffImagePlugin.ImageFileDirectory_v2()
            ifd[33550] = (1 / 1200, 1 / 1200, 0.0)
            ifd.tagtype[33550] = 12
            ifd[33922] = (0.0, 0.0, 0.0, 11.0, 47.0, 0.0)
            ifd.tagtype[33922] = 12
            if pred != 1:
                if comp in (None, "packbits"):
                    continue                              # libtiff applies predictors only with LZW / Deflate
                ifd[317] = pred
            buf = io.BytesIO()
            Image.fromarray(a, mode="F").save(buf, format="TIFF", compression=comp, tiffinfo=ifd)
            back = np.array(Image.open(io.BytesIO(buf.getvalue())))          # libtiff's own decode: the file is sound
            assert np.array_equal(back.view(np.uint32), a.view(np.uint32))
            got = r.decode_geotiff(buf.getvalue())
            assert np.array_equal(got.view(np.uint32), a.view(np.uint32)), (comp, pred)
            n_ok += 1
    assert n_ok >= 6


def test_add_terrain_geotiff_equals_add_terrain(topo, orc):
    """Decode + add_terrain in one call renders the frame add_terrain of the same raster renders (and the oracle's)."""
    from tiff_writer import write_geotiff
    sc = Scene(64, 1, 2, eye_dh=80)
    g, o = both(topo, orc, 160, 96)
    for loc in sc.locs:
        rp, mp, ps = sc.transform(loc)
        data = write_geotiff(sc.heights[loc], tile=(32, 32), pixel_scale=(float(ps[0]), float(ps[1]), 0.0),
                             tie_points=(float(rp[0]), float(rp[1]), 0.0, float(mp[0]), float(mp[1]), 0.0))
        w, h, ct = topo.geotiff_info(data)
        assert (w, h) == (64, 64)
        assert np.array_equal(ct.model_point, np.asarray(mp, np.float32)) and np.array_equal(ct.pixel_scale, np.asarray(ps, np.float32))
        g.add_terrain_geotiff(loc[0], loc[1], data)
    sc.load(o)
    u, pu = sc.uniforms(160, 96, 40, 12, 70, 0), topo.post_uniforms(160, 96)
    g.update(160, 96, u, pu)
    o.update(160, 96, u, pu)
    assert_same_frame(g.render(), o.render(), "tiles loaded from GeoTIFF bytes")
    assert np.array_equal(g.read_normals(*sc.locs[0]), o.read_normals(sc.locs[0][0], sc.locs[0][1], 64, 64))


def test_sixty_four_views_in_one_submission(topo, orc):
    """The per-submission limit (8 viewpoints x 8 sectors, as bench.py's batch mode submits them): every view equals the
    oracle's frame for its uniforms; 65 views are rejected."""
    import torch
    sc = Scene(48, 2, 2, eye_dh=70)
    g, o = both(topo, orc, 64, 48)
    sc.load(g)
    sc.load(o)
    W, H = 64, 48
    views = []
    for y in range(8):
        views += sc.panorama(W, H, yaw0_deg=13.0 * y)
    packed = np.ascontiguousarray(np.stack([np.ascontiguousarray(u).view(np.uint8).reshape(160) for u in views]))
    s = torch.zeros((64, H, W, 4), dtype=torch.uint8, device="cuda")
    d = torch.zeros((64, H, W), dtype=torch.float32, device="cuda")
    g.render_views_device(packed, W, H, s.data_ptr(), H * W * 4, W * 4, d.data_ptr(), H * W * 4, W * 4)
    g.synchronize()
    sh, dh = s.cpu().numpy(), d.cpu().numpy()
    for k in (0, 7, 8, 21, 38, 63):
        o.update(W, H, views[k], topo.post_uniforms(W, H))
        assert_same_frame((sh[k], dh[k]), o.render(), f"view {k} of 64")
    # up to eight views travel as a kernel argument (k_put_views), more through the pinned staging ring: same frames either way
    for lo in (0, 24, 56):
        s8 = torch.zeros((8, H, W, 4), dtype=torch.uint8, device="cuda")
        d8 = torch.zeros((8, H, W), dtype=torch.float32, device="cuda")
        g.render_views_device(packed[lo:lo + 8], W, H, s8.data_ptr(), H * W * 4, W * 4, d8.data_ptr(), H * W * 4, W * 4)
        g.synchronize()
        assert np.array_equal(s8.cpu().numpy(), sh[lo:lo + 8]) and np.array_equal(d8.cpu().numpy().view(np.uint32), dh[lo:lo + 8].view(np.uint32)), lo
    with pytest.raises(topo.TopoError):
        g.render_views_device(views + views[:1], W, H, s.data_ptr(), H * W * 4, W * 4, d.data_ptr(), H * W * 4, W * 4)


def test_random_frames(topo, orc):
    """Seeded sweep over camera poses and targets the hand-picked cases do not name: odd target sizes, steep pitches
    (ground filling the frame, triangles cut by the near plane), wide and narrow fields of view, eyes a few metres to tens
    of kilometres above the surface, all three view modes -- every frame bit-identical to the oracle's."""
    # (TOPO_FUZZ_FRAMES / TOPO_FUZZ_SEED: a longer or different sweep, run by hand on a GPU box)
    rng = np.random.default_rng(int(os.environ.get("TOPO_FUZZ_SEED", "20261004")))
    rng_split = np.random.default_rng(7)      # the occlusion split: off, inside the scene (far phase runs), the default (the host proves it empty)
    scenes = {}
    for i in range(int(os.environ.get("TOPO_FUZZ_FRAMES", "600"))):
        tile = int(rng.choice([24, 40, 64, 96]))
        n_lat, n_lon = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        dh = float(rng.choice([3.0, 12.0, 50.0, 50.0, 400.0, 5000.0, 40000.0]))
        key = (tile, n_lat, n_lon, dh)
        if key not in scenes:
            sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
            g, o = both(topo, orc, 16, 16)
            sc.load(g)
            sc.load(o)
            scenes[key] = (sc, g, o)
        sc, g, o = scenes[key]
        W, H = int(rng.integers(17, 230)), int(rng.integers(9, 160))
        yaw, pitch = float(rng.uniform(0, 360)), float(rng.choice([rng.uniform(-20, 20), rng.uniform(20, 89), rng.uniform(-89, -20)]))
        fov, mode = float(rng.uniform(12, 150)), int(rng.integers(0, 3))
        u, pu = sc.uniforms(W, H, yaw, pitch, fov, mode), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        split = float(rng_split.choice([0.0, 3000.0, 15000.0, 90000.0]))
        g.set_occlusion_split(split)
        assert_same_frame(g.render(), o.render(), f"random frame {i}: tile {tile} {n_lat}x{n_lon} dh {dh} {W}x{H} yaw {yaw:.1f} pitch {pitch:.1f} fov {fov:.1f} mode {mode} split {split}")
        assert (g.counters()["status"] & 1) == 0


def test_strip_edges_of_the_resolve_pass(topo, orc):
    """k_resolve works in 64 x 4 px strips of 64 x 16 px blocks, each wave with its own 1 px halo ring: targets one to a few
    pixels high, heights around the strip and block sizes, widths just under / at / over one and two strip widths -- the
    clamped halo, the partial last strip and the lanes beyond the right edge all decide pixels here (contours included)."""
    sc = Scene(40, 1, 1, eye_dh=30.0)
    g, o = both(topo, orc, 16, 16)
    sc.load(g)
    sc.load(o)
    sizes = [(1, 1), (2, 1), (5, 3), (63, 4), (64, 4), (65, 5), (64, 15), (127, 16), (128, 17), (129, 7), (191, 33), (3, 70)]
    for k, (W, H) in enumerate(sizes):
        for mode in (0, 2):
            u, pu = sc.uniforms(W, H, 40.0 * k, -12.0 + 3.0 * k, 70.0, mode), topo.post_uniforms(W, H)
            g.update(W, H, u, pu)
            o.update(W, H, u, pu)
            assert_same_frame(g.render(), o.render(), f"strip edges: {W}x{H} mode {mode}")


def test_timing_slots_select_events(topo, orc):
    """topo_set_timing_slots: unselected per-kernel slots read 0, the total is measured unless TOPO_TIMING_NO_TOTAL is set, frames
    are unchanged (and so are the counters, which k_resolve stores into the pinned status ring itself)."""
    sc = Scene(64, 1, 1, eye_dh=60)
    g, o = both(topo, orc, 96, 64)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(96, 64, 10, 5, 60, 0), topo.post_uniforms(96, 64)
    o.update(96, 64, u, pu)
    want = o.render()
    g.update(96, 64, u, pu)
    for names in (None, ("resolve",), ("raster", "cull"), ()):
        g.set_timing_slots(names)
        assert_same_frame(g.render(), want, f"timing slots {names}")
        tm = g.timings()
        assert tm["total"] > 0.0
        for k in ("clear", "cull", "raster", "occlusion", "raster_big", "resolve"):
            if names is None or k in names:
                assert tm[k] > 0.0, (names, k)
            else:
                assert tm[k] == 0.0, (names, k)
    base = g.counters()
    for names in (("resolve",), ()):
        g.set_timing_slots(names, total=False)
        assert_same_frame(g.render(), want, f"timing slots {names}, no total")
        tm = g.timings()
        assert tm["total"] == 0.0 and (tm["resolve"] > 0.0) == ("resolve" in names) and tm["raster"] == 0.0
        assert g.counters() == base and base["near_blocks"] > 0
    g.set_timing_slots(None)
    g.render()
    assert g.timings()["total"] > 0.0


def test_fall_back_paths_render_the_same_frames():
    """The library's older ways of doing what stands around a frame's kernels stay selectable by environment (tools/README.md): the
    view constants through the pinned ring or through k_put_views, the status words copied behind the frame, timing events as
    markers, the far phase always launched, clear and cull as two launches.  Each is read once per process, so each gets a process
    of its own (tests/env_paths_worker.py): every frame, depth image and counter set must hash to what the default path gives."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_paths_worker.py")

    def run(extra):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (extra, out.stderr[-2000:])
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("SHA256 ")]
        assert len(lines) == 1, (extra, out.stdout[-500:])
        return lines[0]

    want = run({})
    for extra in ({"TOPO_VIEWS_IN_CULL": "0"}, {"TOPO_VIEWS_BY_COPY": "1"}, {"TOPO_STATUS_BY_COPY": "1"}, {"TOPO_EVENTS_BY_MARKER": "1"},
                  {"TOPO_FAR_SKIP": "0"}, {"TOPO_FUSE_CLEAR_CULL": "0"}, {"TOPO_LOAD_FUSED": "0"}):
        assert run(extra) == want, extra


def test_render_device_equals_render(topo, orc):
    """topo_render_device (device outputs, stream-ordered) gives the bytes topo_render copies to the host; a padded pitch works."""
    import torch
    sc = Scene(64, 1, 2, eye_dh=70)
    g, o = both(topo, orc, 100, 60)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(100, 60, 25, 8, 65, 0), topo.post_uniforms(100, 60)
    g.update(100, 60, u, pu)
    o.update(100, 60, u, pu)
    want = o.render()
    rgba = torch.zeros((60, 128, 4), dtype=torch.uint8, device="cuda")       # pitch 512 B > 400 B
    depth = torch.zeros((60, 128), dtype=torch.float32, device="cuda")
    g.render_device(rgba.data_ptr(), 128 * 4, depth.data_ptr(), 128 * 4)
    g.synchronize()
    assert_same_frame((rgba[:, :100].cpu().numpy(), depth[:, :100].cpu().numpy()), want, "topo_render_device")
    assert int(rgba[:, 100:].max()) == 0 and float(depth[:, 100:].abs().max()) == 0.0      # the padding is not touched


# ---- round 2: the knobs, configs and entry points round 1 left without a GPU parity test ------------------------------

@pytest.mark.parametrize("rows", [4, 8, 16, 32, 64])
def test_normals_lds_tile_sizes_byte_exact(topo, orc, rows):
    """BASELINE config 3's knob: every instantiation of k_normals_interior<ROWS> (topo_set_normals_lds_rows) against the
    oracle, on a 3x3 mosaic whose tile size is not a multiple of any ROWS (partial LDS tiles on both axes), both through
    add_terrain (one tile per launch) and through topo_recompute_normals (the batched launch the bench times)."""
    tile = 150
    sc = Scene(tile, 3, 3)
    g, o = both(topo, orc, 16, 16)
    g.set_normals_lds_rows(rows)
    sc.load(g)
    sc.load(o)
    ref = {loc: o.read_normals(loc[0], loc[1], tile, tile) for loc in sc.locs}
    for loc in sc.locs:
        a = g.read_normals(*loc)
        assert np.array_equal(a, ref[loc]), f"rows {rows} add_terrain tile {loc}: {np.argwhere((a != ref[loc]).any(axis=-1))[:4]}"
    g.recompute_normals()
    for loc in sc.locs:
        assert np.array_equal(g.read_normals(*loc), ref[loc]), f"rows {rows} recompute tile {loc}"
    with pytest.raises(topo.TopoError):
        g.set_normals_lds_rows(12)


@pytest.mark.parametrize("tile", [152, 260, 516])
def test_normals_without_lds_byte_exact(topo, orc, tile):
    """topo_set_normals_lds_rows(0): the interior normals without an LDS tile (k_normals_rolling: four texels per lane, the rows
    above and below kept in registers, neighbours by DPP wave shifts) against the oracle: tile widths that end inside a strip
    of 256 columns, exactly on one and beyond two, heights that do not fill the last chunk of rows; add_terrain and the
    batched recompute; a width that is not a multiple of four falls back to the LDS form."""
    sc = Scene(tile, 2, 2)
    g, o = both(topo, orc, 16, 16)
    g.set_normals_lds_rows(0)
    sc.load(g)
    sc.load(o)
    ref = {loc: o.read_normals(loc[0], loc[1], tile, tile) for loc in sc.locs}
    for loc in sc.locs:
        a = g.read_normals(*loc)
        assert np.array_equal(a, ref[loc]), f"add_terrain tile {loc}: {np.argwhere((a != ref[loc]).any(axis=-1))[:4]}"
    g.recompute_normals()
    for loc in sc.locs:
        assert np.array_equal(g.read_normals(*loc), ref[loc]), f"recompute tile {loc}"
    if tile == 152:
        sc2 = Scene(150, 1, 2)
        g2, o2 = both(topo, orc, 16, 16)
        g2.set_normals_lds_rows(0)
        sc2.load(g2)
        sc2.load(o2)
        for loc in sc2.locs:
            assert np.array_equal(g2.read_normals(*loc), o2.read_normals(loc[0], loc[1], 150, 150))


def test_config1_single_tile_small_panorama(topo, orc):
    # BASELINE config 1 (the reference's CPU-plumbing case): one 1200x1200 tile, 1024x256 panorama = 8 sectors of 128x256
    sc = Scene(1200, 1, 1, lat0=40, lon0=10, vfrac=(0.623, 0.717))
    sw, sh = 128, 256
    g, o = both(topo, orc, sw, sh)
    sc.load(g)
    sc.load(o)
    for mode in (0, 1, 2):
        views = sc.panorama(sw, sh, mode=mode)
        rgba, depth = _strip(topo, g, views, sw, sh)
        o.update(sw, sh, views[0], topo.post_uniforms(sw, sh))
        ro, do = o.render_views(views, threads=8)
        for k in range(8):
            assert_same_frame((rgba[k], depth[k]), (ro[k], do[k]), f"config-1 mode {mode} sector {k}")
    assert (depth < 1.0).mean() > 0.2
    # a lone tile around the viewpoint: the host proves that (next to) no block lies beyond the occlusion split, raises the frame's
    # split above the farthest block and leaves the far phase's four launches out
    assert not g.debug_far_phase_launched() and g.counters()["far_tested"] == 0
    g.set_occlusion_split(30000.0)          # ... which a split well inside the tile does not allow
    rgba2, depth2 = _strip(topo, g, views, sw, sh)
    assert g.debug_far_phase_launched() and g.counters()["far_tested"] > 0
    assert np.array_equal(rgba2, rgba) and np.array_equal(depth2.view(np.uint32), depth.view(np.uint32))


def test_config3_mosaic25_sector_and_batching(topo, orc):
    # BASELINE config 3: 5x5 degree mosaic (25 tiles of 1200x1200), 8192x2048 panorama = 8 sectors of 1024x2048
    deg, sw, sh, tile = 5, 1024, 2048, 1200
    sc = Scene(tile, deg, deg, lat0=40, lon0=10, vfrac=(0.5 + 0.123 / deg, 0.5 + 0.217 / deg))
    g, o = both(topo, orc, sw, sh)
    sc.load(g)
    sc.load(o)
    views = sc.panorama(sw, sh)
    rgba, depth = _strip(topo, g, views, sw, sh)
    # whole-panorama batching invariance: one submission = two halves = eight single-view submissions
    for lo, hi in ((0, 4), (4, 8)) + tuple((k, k + 1) for k in (0, 3, 7)):
        ra, da = _strip(topo, g, views[lo:hi], sw, sh)
        assert np.array_equal(ra, rgba[lo:hi]) and np.array_equal(da.view(np.uint32), depth[lo:hi].view(np.uint32)), (lo, hi)
    # one full-size sector against the oracle (seams between the 25 tiles are in view), + its normals
    k = 5
    o.update(sw, sh, views[k], topo.post_uniforms(sw, sh))
    assert_same_frame((rgba[k], depth[k]), o.render(), f"config-3 sector {k}")
    for loc in (sc.locs[0], sc.locs[12], sc.locs[24]):
        assert np.array_equal(g.read_normals(*loc), o.read_normals(loc[0], loc[1], tile, tile))
    assert (g.counters()["status"] & 2) == 0
    assert g.debug_far_phase_launched() and g.counters()["far_tested"] > 0


@pytest.fixture(scope="module")
def mosaic100(topo, orc):
    """The 10x10 degree COP90-shaped mosaic of BASELINE configs 4 and 5, resident once on the GPU and in the oracle."""
    import math
    deg, tile = 10, 1200
    locs = topo.synth.mosaic_locations(40, 10, deg, deg)
    g, o = topo.TerrainRenderer(512, 1024), orc.OracleRenderer(512, 1024)
    heights = {}
    for (la, lo) in locs:
        h = topo.synth_tile(la, lo, tile, tile)
        tr = topo.synth.tile_transform(la, lo, tile, tile)
        g.add_terrain(la, lo, h, *tr)
        o.add_terrain(la, lo, h, *tr)
        heights[(la, lo)] = h
    return {"g": g, "o": o, "locs": locs, "heights": heights, "deg": deg, "tile": tile}


def test_config5_batch_submission_over_the_mosaic(topo, orc, mosaic100):
    """BASELINE config 5's shape at full input size: 8 viewpoints x 8 sectors = 64 views of 512x1024 in ONE submission over
    the 100-tile mosaic, two such submissions in flight (pipeline depth 2); three views of each against the oracle, and
    the pipelined outputs against the serial ones."""
    import math
    import torch
    g, o, sw, sh = mosaic100["g"], mosaic100["o"], 512, 1024
    rng = np.random.default_rng(55)
    sets = []
    for s in range(2):
        vs = []
        for _ in range(8):
            lat, lon, yaw = 41.0 + 8.0 * rng.random(), 11.0 + 8.0 * rng.random(), 2 * math.pi * rng.random()
            key = (int(math.floor(lat)), int(math.floor(lon)))
            ground = topo.synth.height_at(mosaic100["heights"][key], key[0], key[1], lon, lat)
            eye = topo.geometry_transform(ground + 50.0, lon, lat)
            vs += topo.panorama_uniforms(eye, yaw, sw, sh, lon, lat, 0)
        sets.append(vs)
    serial = [_strip(topo, g, vs, sw, sh) for vs in sets]
    g.set_pipeline_depth(2)
    outs = [(torch.zeros((64, sh, sw, 4), dtype=torch.uint8, device="cuda"), torch.zeros((64, sh, sw), dtype=torch.float32, device="cuda")) for _ in range(2)]
    for rep in range(2):
        for i, vs in enumerate(sets):
            g.render_views_device(vs, sw, sh, outs[i][0].data_ptr(), sh * sw * 4, sw * 4, outs[i][1].data_ptr(), sh * sw * 4, sw * 4)
    g.join()
    g.set_pipeline_depth(1)
    for i in range(2):
        assert np.array_equal(outs[i][0].cpu().numpy(), serial[i][0]), f"pipelined submission {i}: colour"
        assert np.array_equal(outs[i][1].cpu().numpy().view(np.uint32), serial[i][1].view(np.uint32)), f"pipelined submission {i}: depth"
    assert g.frame_status()["status"] & 6 == 0
    for i, pick in ((0, (0, 21, 63)), (1, (7, 40, 58))):
        o.update(sw, sh, sets[i][pick[0]], topo.post_uniforms(sw, sh))
        ro, do = o.render_views([sets[i][k] for k in pick], threads=3)
        for j, k in enumerate(pick):
            assert_same_frame((serial[i][0][k], serial[i][1][k]), (ro[j], do[j]), f"config-5 submission {i} view {k}")


def test_visible_peaks_device_entry_point(topo, orc):
    """topo_visible_peaks_device: peaks, depth and results stay in device memory (any view of a submission)."""
    import torch
    from scenes import random_peaks
    sc = Scene(96, 2, 2, eye_dh=300.0)
    sw, sh = 200, 120
    g, o = both(topo, orc, sw, sh)
    sc.load(g)
    sc.load(o)
    views = sc.panorama(sw, sh, yaw0_deg=33.0)
    n = len(views)
    rgba = torch.zeros((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
    pitch = topo.pad_256(4 * sw)                                    # the reference's depth row pitch
    depth = torch.zeros((n, sh, pitch // 4), dtype=torch.float32, device="cuda")
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.render_views_device(views, sw, sh, rgba.data_ptr(), sh * sw * 4, sw * 4, depth.data_ptr(), sh * pitch, pitch)
    peaks = random_peaks(sc, 1500)
    d_peaks = torch.from_numpy(peaks).cuda()
    d_vis = torch.zeros(len(peaks), dtype=torch.uint8, device="cuda")
    d_xy = torch.zeros((len(peaks), 2), dtype=torch.int32, device="cuda")
    seen = 0
    for k in (0, 3, 6):
        g.visible_peaks_device(views[k], sw, sh, depth[k].data_ptr(), pitch, len(peaks), d_peaks.data_ptr(), d_vis.data_ptr(), d_xy.data_ptr())
        torch.cuda.synchronize()
        o.update(sw, sh, views[k], topo.post_uniforms(sw, sh))
        o.render()
        vo, xo = o.visible_peaks(peaks)
        assert np.array_equal(d_vis.cpu().numpy().astype(bool), vo)
        assert np.array_equal(d_xy.cpu().numpy().view(np.uint32), xo)
        seen += int(vo.sum())
    assert seen > 10


def test_add_terrain_device_entry_point(topo, orc):
    """topo_add_terrain_device (heights already in HBM) leaves the renderer in the state topo_add_terrain does."""
    import torch
    sc = Scene(80, 2, 2)
    W, H = 160, 96
    g, o = both(topo, orc, W, H)
    keep = []
    for loc in sc.locs:
        d = torch.from_numpy(np.ascontiguousarray(sc.heights[loc])).cuda()
        keep.append(d)
        g.add_terrain_device(loc[0], loc[1], d.data_ptr(), 80, 80, *sc.transform(loc))
        d.zero_()                                     # the call copied: the caller's buffer is free again
    torch.cuda.synchronize()
    sc.load(o)
    for loc in sc.locs:
        assert np.array_equal(g.read_normals(*loc), o.read_normals(loc[0], loc[1], 80, 80))
    u, pu = sc.uniforms(W, H, 140, 15, 70, 0), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    assert_same_frame(g.render(), o.render(), "after add_terrain_device")


def test_bounds_checked_build_is_clean_and_identical(topo):
    """libtopo_hip_check.so = the product with every device-side index tested before use (TOPO_BOUNDS_CHECK; the GPU
    address sanitizer this pool does not offer).  The scenes of tests/bounds_scenes.py -- near-field giants, clipping, queue
    overflows, a 64-view submission, odd sizes, a full-size config-2 panorama, frames in flight -- run through it in a
    child process: no index may be out of range, and every frame must hash to what the product build renders."""
    import json
    import os
    import subprocess
    import sys
    import bounds_scenes
    check = os.path.join(os.path.dirname(topo.LIB_PATH), "libtopo_hip_check.so")
    assert os.path.exists(check), "run __graft_entry__.build()"
    env = dict(os.environ, TOPO_HIP_LIB=check)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "bounds_scenes.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().split("\n")[-1])
    assert got["lib"].endswith("libtopo_hip_check.so")
    want = bounds_scenes.run(topo)
    assert set(got["cases"]) == set(want["cases"]) and len(want["cases"]) >= 12
    for name, c in got["cases"].items():
        assert not c["bounds_violation"], f"{name}: out-of-range index at site {c['bounds_site']}, value {c['bounds_value']}"
        assert c["sha"] == want["cases"][name]["sha"], f"{name}: the checked build renders a different frame"
        assert c["status"] == want["cases"][name]["status"]


@pytest.mark.parametrize("tile,n,split", [(24, 4, 15000.0), (96, 3, 30000.0), (300, 2, 20000.0), (720, 2, 20000.0)])
def test_occlusion_filter_is_conservative_on_coarse_tiles(topo, orc, tile, n, split):
    """The far-block slab of the occlusion filter is flat-faced; the patch of a coarse tile's block bulges out of it by its
    sagitta (hundreds of metres when a 60 x 15 cell block spans a degree).  k_block_minmax measures that: blocks beyond the
    1 m allowance are never filtered, finer ones get the sagitta added to the slab.  Filter on == filter off == oracle."""
    sc = Scene(tile, n, n, vfrac=(0.08, 0.07), eye_dh=900.0)
    W, H = 256, 128
    g, o = both(topo, orc, W, H)
    sc.load(g)
    sc.load(o)
    pu = topo.post_uniforms(W, H)
    tested = 0
    for yaw, pitch in ((45, 4), (20, 1), (70, 8)):
        u = sc.uniforms(W, H, yaw, pitch, 50, 0)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        ref = o.render()
        g.set_occlusion_split(split)
        assert_same_frame(g.render(), ref, f"tile {tile} split {split} yaw {yaw}")
        tested += g.counters()["far_tested"]
        g.set_occlusion_split(0.0)
        assert_same_frame(g.render(), ref, f"tile {tile} filter off yaw {yaw}")
    if tile <= 300:
        assert tested == 0          # sagitta 206 m / 53 m / 5.5 m: beyond the allowance, never filtered
    else:
        assert tested > 0           # 720-px tiles: 0.95 m -- filtered, with the sagitta added to the slab


def test_two_rank_panorama_over_rccl(topo, tmp_path):
    """The N > 1 half of the C-ABI panorama path on real hardware: two processes, two GPUs, RCCL bound by libtopo_hip.so itself
    (topo_comm_unique_id / topo_comm_init), the frame resolved slot by slot with each slot's exchange under the next slot's
    resolve.  Every rank must end up with the strip a world of one renders.  Skipped on a box with fewer than two GPUs."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(rk), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rk in (0, 1)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    sc = Scene(96, 2, 2, eye_dh=120.0)
    sw, sh = 96, 160
    g = topo.TerrainRenderer(sw, sh)
    sc.load(g)
    ra, _ = _strip(topo, g, sc.panorama(sw, sh, yaw0_deg=25.0), sw, sh)
    for rk in (0, 1):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"strip{rk}.npy")), ra), f"rank {rk}'s strip differs from the world-of-one strip"


def test_panorama_and_batch_entry_points(topo, orc):
    """topo_render_panorama (world of one: all 8 sectors, no collective) and topo_render_batch -- the multi-GPU entry points of
    the C ABI -- produce what topo_render_views_device produces from the same cameras; one sector and one batch viewpoint
    against the oracle."""
    import torch
    sc = Scene(96, 2, 2, eye_dh=120.0)
    sw, sh = 96, 160
    g, o = both(topo, orc, sw, sh)
    sc.load(g)
    sc.load(o)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    comm = topo.Comm(0, 1)
    assert list(topo.panorama_sector_range(0, 1)) == list(range(8)) and list(topo.panorama_sector_range(1, 4)) == [2, 3]
    strip = torch.zeros((8, sh, sw, 4), dtype=torch.uint8, device="cuda")
    depth = torch.zeros((8, sh, sw), dtype=torch.float32, device="cuda")
    yaw0 = math.radians(25.0)
    for use_comm in (comm, None):
        strip.zero_(); depth.zero_()
        g.render_panorama(use_comm, sc.eye, yaw0, sw, sh, sc.vlon, sc.vlat, strip.data_ptr(), depth.data_ptr())
        g.synchronize()
        views = sc.panorama(sw, sh, yaw0_deg=25.0)
        ra, da = _strip(topo, g, views, sw, sh)
        assert np.array_equal(strip.cpu().numpy(), ra) and np.array_equal(depth.cpu().numpy().view(np.uint32), da.view(np.uint32))
    o.update(sw, sh, views[3], topo.post_uniforms(sw, sh))
    assert_same_frame((ra[3], da[3]), o.render(), "panorama sector 3")
    # the slot-by-slot resolve of the N > 1 path (k_resolve over block ranges, one launch per (sector, band of rows)) without
    # its exchange -- what a one-GPU box can run of it: the same strip, for bands that cut the sectors into several slots
    os.environ["TOPO_PANORAMA_FORCE_SLOTS"] = "1"
    os.environ["TOPO_PANORAMA_BAND_BYTES"] = str(sw * 4 * 50)
    try:
        assert len(topo.panorama_slots(2, sw, sh)) == 4 * 3       # 160 rows = five block rows of 32 in bands of two
        strip.zero_(); depth.zero_()
        g.render_panorama(comm, sc.eye, yaw0, sw, sh, sc.vlon, sc.vlat, strip.data_ptr(), depth.data_ptr())
        g.synchronize()
        assert np.array_equal(strip.cpu().numpy(), ra) and np.array_equal(depth.cpu().numpy().view(np.uint32), da.view(np.uint32))
    finally:
        del os.environ["TOPO_PANORAMA_FORCE_SLOTS"], os.environ["TOPO_PANORAMA_BAND_BYTES"]
    # batch: 11 viewpoints (one full group of 8 + a partial one), two submissions in flight
    rng = np.random.default_rng(9)
    eyes, yaws, suns = [], [], []
    for _ in range(11):
        lat, lon = 45.1 + 1.8 * rng.random(), 15.1 + 1.8 * rng.random()
        key = (int(math.floor(lat)), int(math.floor(lon)))
        ground = topo.synth.height_at(sc.heights[key], key[0], key[1], lon, lat)
        eyes.append(topo.geometry_transform(ground + 80.0, lon, lat)); yaws.append(2 * math.pi * rng.random()); suns.append((lon, lat))
    out = torch.zeros((11, 8, sh, sw, 4), dtype=torch.uint8, device="cuda")
    dout = torch.zeros((11, 8, sh, sw), dtype=torch.float32, device="cuda")
    g.set_pipeline_depth(2)
    g.render_batch(eyes, yaws, suns, sw, sh, out.data_ptr(), dout.data_ptr())
    g.join()
    g.set_pipeline_depth(1)
    for v in (0, 7, 8, 10):
        views = topo.panorama_uniforms(eyes[v], yaws[v], sw, sh, suns[v][0], suns[v][1], 0)
        ra, da = _strip(topo, g, views, sw, sh)
        assert np.array_equal(out[v].cpu().numpy(), ra) and np.array_equal(dout[v].cpu().numpy().view(np.uint32), da.view(np.uint32)), v
    o.update(sw, sh, views[5], topo.post_uniforms(sw, sh))
    assert_same_frame((out[10, 5].cpu().numpy(), dout[10, 5].cpu().numpy()), o.render(), "batch viewpoint 10 sector 5")
    with pytest.raises(topo.TopoError):
        topo.Comm(0, 3)                   # 8 sectors do not divide among 3 ranks


def test_change_location_on_the_renderer(topo, orc):
    """topo_change_location: the renderer's own tile set is the `loaded_locations`; tiles that left the 100 km range are
    unloaded (their seams stay as they are, like unload_terrain), the missing ones are reported in get_locations_range's order."""
    sc = Scene(32, 3, 3, lat0=44, lon0=14)          # tiles 44..46 x 14..16
    g, o = both(topo, orc, 64, 48)
    sc.load(g)
    sc.load(o)
    req, n_un = g.change_location(45.623, 15.717)    # range = exactly these nine tiles
    assert req == [] and n_un == 0
    req, n_un = g.change_location(46.7, 16.9)        # moved north-east: rows 46..47, columns 15..18 -> five of ours leave
    want = topo.locations_range(46.7, 16.9)
    kept = [l for l in sc.locs if l in want]
    assert n_un == 9 - len(kept) and req == [l for l in want if l not in sc.locs]
    for l in sc.locs:
        if l not in kept:
            o.unload_terrain(*l)
    u, pu = sc.uniforms(64, 48, 40, 20, 70, 0), topo.post_uniforms(64, 48)
    g.update(64, 48, u, pu)
    o.update(64, 48, u, pu)
    assert_same_frame(g.render(), o.render(), "after change_location")


@pytest.mark.parametrize("fmt", [1, 2, 3, 4], ids=["rgba_srgb", "bgra_srgb", "rgba_unorm", "bgra_unorm"])
def test_surface_formats(topo, orc, fmt):
    """The four surface formats the reference can end up with (render_engine.rs:77-84), each bit-exact against the oracle:
    terrain, contour transitions (the decode -> mix -> encode path of the post pass) and sky blocks."""
    sc = Scene(64, 2, 2, eye_dh=90.0)
    W, H = 200, 136
    g, o = topo.TerrainRenderer(W, H, color_format=fmt), orc.OracleRenderer(W, H, color_format=fmt)
    sc.load(g)
    sc.load(o)
    for yaw, pitch, mode in ((30, 12, 0), (200, 40, 1), (110, -5, 2)):
        u, pu = sc.uniforms(W, H, yaw, pitch, 75, mode), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        assert_same_frame(g.render(), o.render(), f"format {fmt} yaw {yaw}")
    with pytest.raises(topo.TopoError):
        topo.TerrainRenderer(W, H, color_format=7)


@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_overlay_lines_over_a_rendered_frame(topo, orc, fmt):
    """the line overlay (leader lines, label boxes; line_renderer.rs + line_shader.wgsl) drawn into the post
    pass's image with the reference's layering -- host entry over the frame topo_render returned, device entry over the
    frame topo_render_device left in HBM -- bit-exact against the oracle; a second overlay on the same renderer starts from
    a clean layer buffer."""
    import torch
    from scenes import overlay_geometry
    sc = Scene(64, 2, 2, eye_dh=120.0)
    W, H = 200, 136
    g, o = topo.TerrainRenderer(W, H, color_format=fmt), orc.OracleRenderer(W, H, color_format=fmt)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(W, H, 30, 12, 75, 0), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    (rg, dg), (ro, do) = g.render(), o.render()
    assert_same_frame((rg, dg), (ro, do), "base frame")
    for seed, width in ((5, 0.5), (6, 2.0)):
        v, ix = overlay_geometry(W, H, seed=seed)
        a, b = np.ascontiguousarray(rg.copy()), np.ascontiguousarray(ro.copy())
        g.overlay_lines(v, ix, a, width)
        o.overlay_lines(v, ix, b, width)
        assert np.array_equal(a, b), f"format {fmt} seed {seed}: {np.argwhere((a != b).any(axis=-1))[:5]}"
        assert (a != rg).any(axis=-1).mean() > 0.05
    # device entry: render + overlay without leaving HBM
    img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.render_device(img.data_ptr(), W * 4)
    g.overlay_lines_device(v, ix, img.data_ptr(), W * 4, width)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), b)
    g.overlay_lines_device(v[:0], ix[:0], img.data_ptr(), W * 4, width)      # an empty overlay changes nothing
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), b)
    with pytest.raises(topo.TopoError):
        g.overlay_lines(v, ix[:4], a, width)


@pytest.mark.parametrize("fmt", [1, 2, 3])
def test_text_overlay_over_a_rendered_frame(topo, orc, fmt):
    """SURVEY 8f rank 4, the text half: glyphon's glyph quads (text_renderer.rs:198-204, :259-291) alpha-blended into the post
    pass's image after the lines, the first quad over a pixel keeping it -- host entry and device entry, bit-exact against
    the oracle; lines and text in the reference's order; colour glyphs and a depth at or below the post quad's are refused."""
    import torch
    from scenes import glyph_scene, overlay_geometry
    sc = Scene(64, 2, 2, eye_dh=120.0)
    W, H = 200, 136
    g, o = topo.TerrainRenderer(W, H, color_format=fmt), orc.OracleRenderer(W, H, color_format=fmt)
    sc.load(g)
    sc.load(o)
    u, pu = sc.uniforms(W, H, 30, 12, 75, 0), topo.post_uniforms(W, H)
    g.update(W, H, u, pu)
    o.update(W, H, u, pu)
    (rg, _), (ro, _) = g.render(want_depth=False), o.render()
    assert np.array_equal(rg, ro)
    v, ix = overlay_geometry(W, H, seed=5)
    for seed in (3, 4):
        glyphs, atlas = glyph_scene(W, H, seed=seed)
        a, b = np.ascontiguousarray(rg.copy()), np.ascontiguousarray(ro.copy())
        g.overlay_lines(v, ix, a)                       # render_engine.rs:215-216: lines, then text, into the same pass
        o.overlay_lines(v, ix, b)
        lines = a.copy()
        g.overlay_glyphs(glyphs, atlas, a)
        o.overlay_glyphs(glyphs, atlas, b)
        assert np.array_equal(a, b), f"format {fmt} seed {seed}: {np.argwhere((a != b).any(axis=-1))[:5]}"
        assert (a != lines).any(axis=-1).mean() > 0.01
    img = torch.from_numpy(lines).cuda()
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.overlay_glyphs_device(glyphs, atlas, img.data_ptr(), W * 4)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), b)
    g.overlay_glyphs_device(glyphs[:0], atlas, img.data_ptr(), W * 4)      # no glyphs: nothing changes
    g.overlay_lines_device(v[:0], ix[:0], img.data_ptr(), W * 4)            # and the shared key image was left clean
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), b)
    bad = glyphs[:3].copy()
    bad["content_type_with_srgb"][1] = (0, 1)
    with pytest.raises(topo.TopoError) as e:
        g.overlay_glyphs(bad, atlas, a)
    assert e.value.code == topo.TOPO_ERR_UNSUPPORTED
    with pytest.raises(topo.TopoError):
        g.overlay_glyphs(glyphs, atlas, a, depth=1.0 / 4096.0)


def test_host_outputs_staged_and_pinned(topo, orc):
    """topo_render's two ways out to host memory -- through the context's pinned staging image (slices moved on by host
    threads; odd sizes, a padded depth pitch, a row pitch larger than the row) and straight into buffers the caller pinned
    (topo_pin_host_buffer) -- hand back the same bytes, the oracle's; unpinning a buffer that was never pinned is an error."""
    sc = Scene(48, 2, 2, eye_dh=80.0)
    for (W, H) in ((333, 129), (640, 512)):
        g, o = both(topo, orc, W, H)
        sc.load(g)
        sc.load(o)
        u, pu = sc.uniforms(W, H, 20.0, 12.0, 70.0, 0), topo.post_uniforms(W, H)
        g.update(W, H, u, pu)
        o.update(W, H, u, pu)
        ref = o.render()
        rgba = np.zeros((H, W + 3, 4), np.uint8)[:, :W]                      # rows 12 bytes further apart than they are long
        depth = np.full((H, topo.pad_256(4 * W) // 4), -1.0, np.float32)
        g.render_into(rgba, depth)
        assert_same_frame((rgba, depth[:, :W]), ref, f"staged {W}x{H}")
        assert (depth[:, W:] == -1.0).all()                                  # nothing written beyond a row
        big = np.zeros((H, W, 4), np.uint8)
        dbig = np.zeros((H, W), np.float32)
        g.pin_host_buffer(big)
        g.pin_host_buffer(dbig)
        g.pin_host_buffer(big)                                               # (again: fine)
        g.render_into(big, dbig)
        assert_same_frame((big, dbig), ref, f"pinned {W}x{H}")
        big[:] = 0
        g.render_into(big, None)                                             # depth only when asked for
        assert np.array_equal(big, ref[0])
        g.unpin_host_buffer(big)
        g.unpin_host_buffer(dbig)
        with pytest.raises(topo.TopoError) as e:
            g.unpin_host_buffer(dbig)
        assert e.value.code == topo.TOPO_ERR_NOT_FOUND
        g.render_into(big, dbig)                                             # and the staged route again for the same arrays
        assert_same_frame((big, dbig), ref, f"staged after unpin {W}x{H}")
