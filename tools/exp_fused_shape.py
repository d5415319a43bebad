#!/usr/bin/env python3
"""The c4 load phase (100 tiles) for the build's TOPO_FUSED_SHAPE / TOPO_LOAD_FUSED / TOPO_ROLL_SHAPE: python tools/exp_fused_shape.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo_renderer_amd as T
deg, tile = int(os.environ.get("TOPO_SWEEP_DEG", "10")), 1200
r = T.TerrainRenderer(64, 64)
for (la, lo) in T.synth.mosaic_locations(40, 10, deg, deg):
    r.add_terrain(la, lo, T.synth_tile(la, lo, tile, tile), *T.synth.tile_transform(la, lo, tile, tile))
ms = []
for _ in range(10):
    r.recompute_normals()
    tm = r.timings()
    ms.append((tm["load"], tm["load_tables"]))
best = min(ms[2:])
print(f"shape {os.environ.get('TOPO_FUSED_SHAPE', '-')} fused {os.environ.get('TOPO_LOAD_FUSED', '1')}: load {best[0]:.4f} ms (in front of the normals {best[1]:.4f}), "
      f"{8.0 * deg * deg * tile * tile / best[0] / 1e6:.0f} GB/s")
