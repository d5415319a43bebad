#!/usr/bin/env python3
"""Experiment: consecutive c4 panoramas alternating between two renderers on two streams (frame N+1's cull/raster under
frame N's resolve).  python tools/exp_pipeline.py [n_renderers]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import topo_renderer_amd as T
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 2
deg, PW, PH, TILE = 10, 16384, 4096, 1200
SW = PW // 8
locs = T.synth.mosaic_locations(40, 10, deg, deg)
vlat, vlon = 40 + deg / 2 + 0.123, 10 + deg / 2 + 0.217
streams = [torch.cuda.Stream() for _ in range(NR)]
rs = []
ground = None
tiles = {}
for (la, lo) in locs:
    tiles[(la, lo)] = T.synth_tile(la, lo, TILE, TILE)
    if la == int(math.floor(vlat)) and lo == int(math.floor(vlon)):
        ground = T.synth.height_at(tiles[(la, lo)], la, lo, vlon, vlat)
for i in range(NR):
    r = T.TerrainRenderer(SW, PH)
    r.set_stream(streams[i].cuda_stream)
    for (la, lo) in locs:
        r.add_terrain(la, lo, tiles[(la, lo)], *T.synth.tile_transform(la, lo, TILE, TILE))
    rs.append(r)
eye = T.geometry_transform(ground + 50.0, vlon, vlat)
views = T.panorama_uniforms(eye, 0.0, SW, PH, vlon, vlat, 0)
outs = [(torch.empty((8, PH, SW, 4), dtype=torch.uint8, device="cuda"), torch.empty((8, PH, SW), dtype=torch.float32, device="cuda")) for _ in range(NR)]
def step(i):
    r, (s, d) = rs[i % NR], outs[i % NR]
    r.render_views_device(views, SW, PH, s.data_ptr(), PH * SW * 4, SW * 4, d.data_ptr(), PH * SW * 4, SW * 4)
for i in range(6):
    step(i)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"renderers {NR}: {dt * 1e3:.4f} ms per panorama, {PW * PH / dt / 1e9:.2f} Gpix/s")
