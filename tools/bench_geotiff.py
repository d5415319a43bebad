#!/usr/bin/env python3
"""Decode rate of a COP90-shaped GeoTIFF (1200x1200 f32, Deflate + floating-point predictor, 1 strip / 256x256 tiles):
python tools/bench_geotiff.py   (GPU box; prints one JSON line)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import topo_renderer_amd as T
from tiff_writer import write_geotiff
h = T.synth_tile(45, 15, 1200, 1200)
out = {}
r = T.TerrainRenderer(8, 8)
for name, kw in (("deflate_pred3_strip", dict()), ("deflate_pred3_tiles256", dict(tile=(256, 256))), ("none_pred1", dict(compression="none", predictor=1))):
    data = write_geotiff(h, **kw)
    got = r.decode_geotiff(data)
    assert np.array_equal(got.view(np.uint32), h.view(np.uint32))
    t0 = time.perf_counter()
    for _ in range(10):
        r.add_terrain_geotiff(45, 15, data)
    dt = (time.perf_counter() - t0) / 10
    out[name] = {"file_MB": round(len(data) / 1e6, 2), "ms_decode_plus_add_terrain": round(dt * 1e3, 2), "raster_MBps": round(5.76 / dt, 1)}
print(json.dumps(out))
