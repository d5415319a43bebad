"""The C-ABI library loads, exports every symbol include/topo_hip.h declares, refuses to work without a GPU,
and its host-side helpers (the reference's CPU math) agree with the oracle's independent restatement."""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest


def declared_symbols(path):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(topo_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(topo):
    boundary, hooks = declared_symbols(topo.HEADER_PATH), declared_symbols(topo.TEST_HEADER_PATH)
    # the test hooks live in their own header: none of them is declared by the boundary header the Rust bindings come from
    assert not set(boundary) & set(hooks) and {"topo_probe_sincos", "topo_probe_div", "topo_debug_set_queue_caps", "topo_synth_tile"} <= set(hooks)
    decl = sorted(set(boundary) | set(hooks))
    assert len(boundary) >= 20
    nm = subprocess.run(["nm", "-D", "--defined-only", topo.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (topo_[a-z0-9_]+)", nm))
    missing = [s for s in decl if s not in exported]
    assert not missing, missing
    L = topo.lib()
    for s in decl:
        assert hasattr(L, s)
    assert set(L._topo_symbols) == set(decl), set(L._topo_symbols) ^ set(decl)


def test_library_contains_gfx950_code_object(topo):
    out = subprocess.run(["strings", "-n", "6", topo.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for k in ("k_raster", "k_resolve", "k_normals_interior", "k_cull", "k_raster_big", "k_clear"):
        assert k in out, k


def test_no_cpu_fallback_without_gpu(topo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(topo.TopoError) as e:
        topo.TerrainRenderer(64, 64)
    assert e.value.code == topo.TOPO_ERR_HIP and "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle(topo):
    import os
    root = os.path.dirname(topo.HEADER_PATH)
    pkg = os.path.dirname(topo.LIB_PATH)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in src.replace("oracle/ ", "") or f in ("topo_math.h",), f
                assert "liboracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_struct_layouts_and_small_helpers(topo, orc):
    assert topo.pad_256(1) == 256 and topo.pad_256(256) == 256 and topo.pad_256(257) == 512
    assert topo.pad_256(4 * 2048) == 8192 and topo.pad_256(4 * 128) == 512
    for n in (1, 255, 256, 257, 4 * 800, 4 * 2048):
        assert topo.pad_256(n) == orc.pad_256(n)
    for d in (0.0, 0.5, 0.9, 0.999, 1.0):
        assert topo.dist_from_depth(d) == float(orc.lib().oracle_dist_from_depth(d))
    assert topo.dist_from_depth(0.0) == 50.0 and topo.dist_from_depth(1.0) == 500000.0


def test_host_camera_math_matches_oracle_restatement(topo, orc):
    rng = np.random.default_rng(3)
    for _ in range(200):
        lat, lon = rng.uniform(-80, 80), rng.uniform(-179, 179)
        h = rng.uniform(0, 4000)
        a, b = topo.geometry_transform(h, lon, lat), orc.geometry_transform(h, lon, lat)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        args = (a, rng.uniform(-4, 4), rng.uniform(-1.5, 1.5), math.radians(rng.uniform(10, 160)), 2048.0, 4096.0, lon, lat, int(rng.integers(0, 3)))
        u, v = topo.camera_uniforms(*args), orc.camera_uniforms(*args)
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
        tu = (np.float32([0, 0]), np.float32([math.floor(lon), math.floor(lat) + 1]), np.float32([1 / 1200, 1 / 1200]), 1200, 1200)
        assert np.array_equal(topo.terrain_uniforms(*tu).view(np.uint32), orc.terrain_uniforms(*tu).view(np.uint32))
    # degenerate arcs of Quat::from_rotation_arc: eye on the -Y / +Y axis
    for eye in ((0.0, -6.4e6, 0.0), (0.0, 6.4e6, 0.0)):
        args = (np.float32(eye), 0.3, 0.1, 1.0, 800.0, 600.0, 0.0, 0.0, 0)
        assert np.array_equal(topo.camera_uniforms(*args).view(np.uint32), orc.camera_uniforms(*args).view(np.uint32))


def test_sun_at_zenith_and_rotation_semantics(topo):
    # LightAngle{theta: lon, phi: lat}.to_vec3() is the local zenith (camera.rs:44-53, :91-94)
    lon, lat = 15.217, 45.123
    eye = topo.geometry_transform(1000.0, lon, lat)
    u = topo.camera_uniforms(eye, 0.0, 0.0, 1.0, 100.0, 100.0, lon, lat, 0)
    zen = eye / np.linalg.norm(eye)
    assert np.abs(u[36:39] - zen).max() < 1e-6
    tu = topo.terrain_uniforms(np.float32([0, 0]), np.float32([lon, lat]), np.float32([1 / 1200, 1 / 1200]), 1200, 1200)
    rot = tu[8:24].reshape(4, 4).T[:3, :3]
    assert np.abs(rot @ np.array([0, 0, 1.0]) - zen).max() < 1e-6      # normal-space +z -> zenith


def test_synth_numpy_and_cpp_agree(topo):
    for (la, lo, n) in ((45, 15, 64), (-3, -70, 33), (0, 0, 128), (49, 19, 1200)):
        a = topo.synth_tile(la, lo, n, n)
        b = topo.synth.synth_tile(la, lo, n, n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert a.min() >= 0 and a.max() <= 3000
    left, right = topo.synth_tile(45, 15, 64, 64), topo.synth_tile(45, 16, 64, 64)
    assert np.abs(left[:, -1] - right[:, 0]).max() < 60       # continuous across the tile border


def test_sector_fov(topo):
    for sw, sh in ((128, 256), (512, 1024), (1024, 2048), (2048, 4096)):
        assert abs(math.degrees(topo.sector_fov_y(sw, sh)) - 79.2785) < 1e-3


def test_locations_range_matches_oracle_and_known_answer(topo, orc):
    # UiController::get_locations_range (ui_controller.rs:61-83)
    got = topo.locations_range(45.623, 15.717)
    assert sorted(got) == [(la, lo) for la in (44, 45, 46) for lo in (14, 15, 16)]          # 0.64 deg x 0.91 deg half-range
    # sort centre latitude is pinned to 89 by `.min(-90).max(89)`: northern rows first, columns by distance from 15
    assert got == [(46, 15), (46, 14), (46, 16), (45, 15), (45, 14), (45, 16), (44, 15), (44, 14), (44, 16)]
    rng = np.random.default_rng(9)
    for _ in range(300):
        la, lo = float(rng.uniform(-85, 85)), float(rng.uniform(-179.9, 179.9))
        rd = float(rng.choice([1.0e4, 1.0e5, 2.5e5]))
        assert topo.locations_range(la, lo, rd) == orc.locations_range(la, lo, rd), (la, lo, rd)
    assert topo.locations_range(0.2, 179.8)[0][1] in range(-180, 180)                         # wraps across the antimeridian


def test_change_location_plan_matches_the_reference_set_arithmetic(topo, orc):
    """UiController::change_location (ui_controller.rs:23-59), restated here with Python sets over the ORACLE's
    get_locations_range: new = set(range(location)); for each loaded tile: in new -> drop it from new, else -> unload;
    what is left of new is requested."""
    rng = np.random.default_rng(31)
    for it in range(200):
        la, lo = float(rng.uniform(-80, 80)), float(rng.uniform(-179.9, 179.9))
        # a loaded set from an earlier position nearby (overlapping ranges), plus a few strays
        la0, lo0 = la + float(rng.uniform(-1.5, 1.5)), lo + float(rng.uniform(-1.5, 1.5))
        loaded = list(dict.fromkeys(orc.locations_range(min(max(la0, -85.0), 85.0), min(max(lo0, -179.9), 179.9), 1.0e5)))
        loaded = [l for l in loaded if rng.random() < 0.8] + [(int(rng.integers(-60, 60)), int(rng.integers(-170, 170))) for _ in range(2)]
        loaded = list(dict.fromkeys(loaded))
        new = set(orc.locations_range(la, lo, 1.0e5))
        want_unload = [l for l in loaded if l not in new]
        want_request = new - set(loaded)
        unload, request = topo.change_location_plan(la, lo, loaded)
        assert unload == want_unload, (la, lo)
        assert set(request) == want_request and len(request) == len(set(request)), (la, lo)
        # the defined order: get_locations_range's
        order = {l: i for i, l in reversed(list(enumerate(orc.locations_range(la, lo, 1.0e5))))}
        assert request == sorted(request, key=lambda l: order[l])
    assert topo.change_location_plan(45.6, 15.7, []) == ([], topo.locations_range(45.6, 15.7))


def test_coordinate_transform_from_geo_tags(topo):
    """CoordinateTransform::from_geo_tag_data (coordinate_transform.rs:23-57): values, f64 -> f32 narrowing, both errors."""
    ps = [1.0 / 1200.0, 1.0 / 1200.0, 0.0]
    tp = [0.0, 0.0, 0.0, 10.0 - 0.5 / 1200.0, 47.0 + 0.5 / 1200.0, 0.0]
    ct = topo.CoordinateTransform.from_geo_tag_data(ps, tp)
    assert ct.raster_point.tolist() == [0.0, 0.0]
    assert ct.model_point.tolist() == [float(np.float32(tp[3])), float(np.float32(tp[4]))]
    assert ct.pixel_scale.tolist() == [float(np.float32(ps[0]))] * 2
    with pytest.raises(topo.TopoError) as e:          # ModelTransformationTag present -> IncorrectGeoTags
        topo.CoordinateTransform.from_geo_tag_data(ps, tp, [1.0] * 16)
    assert e.value.code == -2
    for a, b in ((None, tp), (ps, None)):             # a required tag absent -> IncorrectGeoTags
        with pytest.raises(topo.TopoError) as e:
            topo.CoordinateTransform.from_geo_tag_data(a, b)
        assert e.value.code == -2
    for a, b in ((ps[:2], tp), (ps, tp[:5]), (ps + [0.0], tp)):   # wrong counts -> IncorrectGeoTagData
        with pytest.raises(topo.TopoError) as e:
            topo.CoordinateTransform.from_geo_tag_data(a, b)
        assert e.value.code == -1


def test_to_model_to_raster_and_height_lookup(topo):
    """to_model / to_raster / get_height_value_at (coordinate_transform.rs:59-90) against the same f32 expressions in numpy."""
    f = np.float32
    ct = topo.CoordinateTransform([0.5, 0.5], [10.0, 48.0], [1.0 / 1200.0, 1.0 / 1200.0])
    rp, mp, sc = ct.raster_point, ct.model_point, ct.pixel_scale
    for x, y in ((0.0, 0.0), (600.0, 37.5), (1199.0, 1199.0)):
        mx, my = ct.to_model(x, y)
        assert f(mx) == (f(x) - rp[0]) * sc[0] + mp[0] and f(my) == (f(y) - rp[1]) * -sc[1] + mp[1]
    w, h = 40, 30
    hts = np.arange(w * h, dtype=np.float32).reshape(h, w)
    ct = topo.CoordinateTransform([0.0, 0.0], [10.0, 48.0], [1.0 / w, 1.0 / h])
    rng = np.random.default_rng(5)
    for lon, lat in zip(rng.uniform(9.9, 11.1, 300), rng.uniform(46.9, 48.1, 300)):
        rx = (f(lon) - ct.model_point[0]) / ct.pixel_scale[0] + ct.raster_point[0]
        ry = (f(lat) - ct.model_point[1]) / -ct.pixel_scale[1] + ct.raster_point[1]
        assert ct.to_raster(lon, lat) == (float(rx), float(ry))
        ix, iy = (int(rx) if rx > 0 else 0), (int(ry) if ry > 0 else 0)     # Rust `as usize` saturates below at 0
        idx = iy * w + ix                                                     # no per-axis bounds check in the reference
        want = float(hts.reshape(-1)[idx]) if idx < w * h else None
        assert ct.height_value_at(hts, lon, lat) == want
    assert ct.height_value_at(hts, float("nan"), 47.5) == float(hts[int((f(47.5) - f(48.0)) / -ct.pixel_scale[1]), 0])


def test_panorama_uniforms_are_the_sector_cameras(topo):
    """topo_panorama_uniforms = camera_uniforms of each sector: yaw0 - k * 360/n degrees (f32 inputs, f64 sector arithmetic),
    vertical FOV 2 atan(tan(180/n deg) * h / w)."""
    eye = topo.geometry_transform(1500.0, 15.3, 45.2)
    for n, (w, h) in ((8, (2048, 4096)), (8, (128, 256)), (4, (300, 200))):
        got = topo.panorama_uniforms(eye, 0.3, w, h, 15.3, 45.2, 1, n_sectors=n, pitch=0.1)
        fov = 2.0 * math.atan(math.tan(math.pi / n) * h / w)
        assert abs(float(topo.lib().topo_sector_fov_y(w, h, n)) - fov) < 1e-6
        yaw0 = float(np.float32(0.3))
        for k in range(n):
            want = topo.camera_uniforms(eye, yaw0 - k * (2.0 * math.pi / n), float(np.float32(0.1)), fov, w, h, 15.3, 45.2, 1)
            assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), (n, k)


def test_header_is_plain_c(topo, tmp_path):
    """include/topo_hip.h is C (not only C++): a C99 harness compiles against it, links libtopo_hip.so and runs its
    host-side entry points; topo_create fails cleanly on a box without a HIP device."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_harness")
    libdir = os.path.dirname(topo.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_harness.c"),
                           "-o", exe, "-L", libdir, "-ltopo_hip", "-Wl,-rpath," + libdir, "-lm"])
    env = dict(os.environ)
    torch_lib = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")
    env["LD_LIBRARY_PATH"] = torch_lib + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi harness ok" in out.stdout


def test_rust_bindings_cover_the_header():
    """rust/topo-hip-sys/src/lib.rs (source only: no Rust toolchain in the image) is generated from include/topo_hip.h: it
    must be what tools/gen_rust_bindings.py generates today, and declare every function the header declares."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(root, "tools", "gen_rust_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(root, "rust", "topo-hip-sys", "src", "lib.rs")).read()
    assert committed == gen.generate(), "run tools/gen_rust_bindings.py"
    hdr = open(os.path.join(root, "include", "topo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(topo_[a-z0-9_]+)\s*\(", hdr))
    bound = set(re.findall(r"pub fn (topo_[a-z0-9_]+)\(", committed))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    wrapper = open(os.path.join(root, "rust", "topo-hip", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(topo_[a-z0-9_]+)\(", wrapper))
    assert used <= bound and {"topo_create", "topo_update", "topo_add_terrain", "topo_unload_terrain", "topo_render", "topo_render_panorama"} <= used


def test_missing_rccl_is_an_error_not_a_crash(topo):
    """A host without librccl.so: topo_comm_unique_id and topo_comm_init(world > 1) return TOPO_ERR_HIP with a message
    (round 2 called dlerror() twice -- the second call returns NULL -- and crashed in std::string); a world of one needs
    no RCCL at all.  Own process: the library is bound once per process."""
    code = ("import numpy as np, topo_renderer_amd as T\n"
            "for f in (lambda: T.comm_unique_id(), lambda: T.Comm(0, 2, np.zeros(128, np.uint8))):\n"
            "    try:\n"
            "        f(); print('NOERR')\n"
            "    except T.TopoError as e:\n"
            "        print('ERR', e.code, str(e))\n"
            "c = T.Comm(0, 1); print('WORLD1', c.world)\n")
    env = dict(os.environ, TOPO_RCCL_LIB="/nonexistent/librccl.so.1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")
    assert len(lines) == 3 and lines[2] == "WORLD1 1", lines
    for l in lines[:2]:
        assert l.startswith(f"ERR {topo.TOPO_ERR_HIP} ") and "librccl.so not found" in l and len(l) > 40, l
