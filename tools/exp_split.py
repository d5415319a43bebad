#!/usr/bin/env python3
"""Experiment: ONE c4 panorama per step, its 8 sectors submitted as G groups rotated through the renderer's frame contexts
(topo_set_pipeline_depth), joined after every panorama: group k+1 is culled and rasterised under group k's resolve INSIDE one step.
python tools/exp_split.py          -> a table over (groups, frames in flight)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import topo_renderer_amd as T
deg, PW, PH, TILE = 10, 16384, 4096, 1200
SW = PW // 8
locs = T.synth.mosaic_locations(40, 10, deg, deg)
vlat, vlon = 40 + deg / 2 + 0.123, 10 + deg / 2 + 0.217
ground, tiles = None, {}
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
r.set_timing_slots(())
VS, DS = PH * SW * 4, PH * SW * 4


def panorama(groups):
    per = 8 // groups
    for g in range(groups):
        r.render_views_device(views[g * per:(g + 1) * per], SW, PH, rgba.data_ptr() + g * per * VS, VS, SW * 4, depth.data_ptr() + g * per * DS, DS, SW * 4)


for groups, depth_ in ((1, 1), (2, 2), (4, 2), (4, 4), (8, 2), (8, 4), (2, 1)):
    r.set_pipeline_depth(depth_)
    for _ in range(4):
        panorama(groups)
        r.join()
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        panorama(groups)
        r.join()                  # the step ends with its panorama complete
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"groups {groups} in flight {depth_}: {dt * 1e3:.4f} ms per panorama (joined per panorama), {PW * PH / dt / 1e9:.2f} Gpix/s", flush=True)
