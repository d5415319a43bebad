#!/usr/bin/env python3
"""Shapes of the LDS-less normals kernel (TOPO_ROLL_SHAPE = rows per wave * 10 + waves per workgroup), one process per shape."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, os
sys.path.insert(0, %r)
import topo_renderer_amd as T
deg, tile = 10, 1200
r = T.TerrainRenderer(64, 64)
for (la, lo) in T.synth.mosaic_locations(40, 10, deg, deg):
    r.add_terrain(la, lo, T.synth_tile(la, lo, tile, tile), *T.synth.tile_transform(la, lo, tile, tile))
r.set_normals_lds_rows(int(os.environ.get("ROWS", "0")))
ms = []
for _ in range(8):
    r.recompute_normals()
    tm = r.timings()
    ms.append(tm["load"] - tm["load_tables"])
print(round(min(ms[2:]), 4))
""" % R
for shape in sys.argv[1:]:
    env = dict(os.environ, TOPO_ROLL_SHAPE=shape, ROWS="32" if shape == "lds32" else "0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    ms = float(out.stdout.strip().split("\n")[-1]) if out.returncode == 0 else None
    print(shape, ms, "ms", round(1152e6 / (ms * 1e-3) / 1e12, 3) if ms else out.stderr[-300:], "TB/s")
