#!/usr/bin/env python3
"""Experiment: where the ~20 us between a frame's GPU span and its wall time go.  One panorama per step, K steps back to back,
wall time per step and the host's share of it (time until the last submission returned).
python tools/exp_gap.py c1|c2|c4 [K]     (EXP_SLOTS=resolve,...: per-kernel timing events; EXP_TOTAL=0: no events around the frame;
TOPO_VIEWS_BY_COPY=1, TOPO_STATUS_BY_COPY=1, TOPO_FAR_SKIP=0: the renderer's older ways)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import topo_renderer_amd as T
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
deg, PW, PH = {"c1": (1, 1024, 256), "c2": (1, 4096, 1024), "c3": (5, 8192, 2048), "c4": (10, 16384, 4096)}[cfg]
TILE = 1200
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
VS = PH * SW * 4
slots = os.environ.get("EXP_SLOTS", "")
TOTAL = os.environ.get("EXP_TOTAL", "1") != "0"
r.set_timing_slots(tuple(x for x in slots.split(",") if x), total=TOTAL)


def frame():
    r.render_views_device(views, SW, PH, rgba.data_ptr(), VS, SW * 4, depth.data_ptr(), VS, SW * 4)


for _ in range(10):
    frame()
torch.cuda.synchronize()
best = None
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(K):
        frame()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cur = ((t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3)
    best = cur if best is None or cur[0] < best[0] else best
hist = r.timing_history(16)
span = float(np.median([h["total"] for h in hist])) if hist else 0.0
print(f"{cfg}: k_resolve {float(np.median([h['resolve'] for h in hist])) if hist else 0.0:.4f} ms (median of 16 frames' events)")
# the host's own cost of a submission: short bursts that fit the queue, the GPU idle at their start
burst = []
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        frame()
    burst.append((time.perf_counter() - t0) / 4 * 1e3)
torch.cuda.synchronize()
print(f"{cfg}: GPU span of a frame {span:.4f} ms (first to last event, median of 16); host cost of a submission {np.median(burst):.4f} ms (bursts of 4 on an idle queue)")
print(f"{cfg} slots=({slots}) total={int(TOTAL)}: {best[0]:.4f} ms per frame wall, host submission {best[1]:.4f} ms per frame", flush=True)
