#!/usr/bin/env python3
"""What ONE rank of an N-GPU panorama renders, timed on one GPU: the c4 frame restricted to the rank's 8/N sectors (one submission,
no exchange).  Feeds the "render" column of DESIGN.md section 6's table; the exchange column stays a prediction.
python tools/exp_rank_share.py"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import topo_renderer_amd as T
deg, PW, PH, TILE = 10, 16384, 4096, 1200
SW = PW // 8
locs = T.synth.mosaic_locations(40, 10, deg, deg)
vlat, vlon = 40 + deg / 2 + 0.123, 10 + deg / 2 + 0.217
ground = None
r = T.TerrainRenderer(SW, PH)
for (la, lo) in locs:
    t = T.synth_tile(la, lo, TILE, TILE)
    if la == int(math.floor(vlat)) and lo == int(math.floor(vlon)):
        ground = T.synth.height_at(t, la, lo, vlon, vlat)
    r.add_terrain(la, lo, t, *T.synth.tile_transform(la, lo, TILE, TILE))
eye = T.geometry_transform(ground + 50.0, vlon, vlat)
views = T.panorama_uniforms(eye, 0.0, SW, PH, vlon, vlat, 0)
rgba = torch.empty((8, PH, SW, 4), dtype=torch.uint8, device="cuda")
depth = torch.empty((8, PH, SW), dtype=torch.float32, device="cuda")
r.set_timing_slots((), total=False)
VS = PH * SW * 4
for n_ranks in (1, 2, 4, 8):
    per = 8 // n_ranks
    worst = 0.0
    for rank in range(n_ranks):
        vs = views[rank * per:(rank + 1) * per]
        for _ in range(3):
            r.render_views_device(vs, SW, PH, rgba.data_ptr() + rank * per * VS, VS, SW * 4, depth.data_ptr() + rank * per * VS, VS, SW * 4)
        r.synchronize(); torch.cuda.synchronize()
        K = 20
        t0 = time.perf_counter()
        for _ in range(K):
            r.render_views_device(vs, SW, PH, rgba.data_ptr() + rank * per * VS, VS, SW * 4, depth.data_ptr() + rank * per * VS, VS, SW * 4)
        r.synchronize(); torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / K * 1e3)
    r.set_timing_slots(None)      # (every timing event on: the kernels' own durations, ~6 us of idle GPU per event on top of `worst`)
    for _ in range(3):
        r.render_views_device(views[:per], SW, PH, rgba.data_ptr(), VS, SW * 4, depth.data_ptr(), VS, SW * 4)
        r.synchronize()
    tm = r.timings()
    r.set_timing_slots(())
    print(f"N = {n_ranks}: {per} sectors per rank, slowest rank's share {worst:.4f} ms; kernels of rank 0's share: "
          + ", ".join(f"{k} {tm[k]:.3f}" for k in ("clear", "cull", "raster", "raster_big", "occlusion", "resolve", "total")), flush=True)
