#!/usr/bin/env python3
"""LDS tile-size sweep of the interior-normals kernel (BASELINE config 3: 5x5 degree mosaic, 25 tiles).

For each ROWS in {4, 8, 16, 32, 64} (output rows per 256-thread workgroup of 128 columns; the LDS tile is (ROWS+2) x 132 floats)
re-runs the load phase over the resident heights and prints the HIP-event time and the achieved GB/s against
the 8 B/texel algorithmic traffic (4 B height read + 4 B normal write).  Run it under
`rocprofv3 --kernel-trace --stats` to get the per-instantiation kernel durations.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo_renderer_amd as T  # noqa: E402

deg = int(os.environ.get("TOPO_SWEEP_DEG", "5"))
tile = 1200
r = T.TerrainRenderer(64, 64)
for (la, lo) in T.synth.mosaic_locations(40, 10, deg, deg):
    r.add_terrain(la, lo, T.synth_tile(la, lo, tile, tile), *T.synth.tile_transform(la, lo, tile, tile))
bytes_ = 8.0 * deg * deg * tile * tile
out = []
for rows in (0, 4, 8, 16, 32, 64):
    r.set_normals_lds_rows(rows)
    ms = []
    for _ in range(8):
        r.recompute_normals()
        tm = r.timings()
        ms.append(tm["load"] - tm["load_tables"])      # the normals K1-K3 (the load phase also holds the tables kernel, which has no LDS knob)
    best = min(ms[2:])
    out.append({"lds_rows": rows, "lds_bytes": (rows + 2) * 132 * 4 + rows * 4 if rows else 128, "load_ms": round(best, 4),
                "what": "k_normals_rolling<4,4,true> (no LDS tile; collects the block min/max too) + k_block_bounds + seams" if rows == 0 and os.environ.get("TOPO_LOAD_FUSED", "1") != "0"
                else ("k_normals_rolling<4,1,false> + seams" if rows == 0 else "k_normals_interior<ROWS> + seams"),
                "GBps": round(bytes_ / (best / 1e3) / 1e9, 1), "frac_of_8TBps": round(bytes_ / (best / 1e3) / 8e12, 4)})
print(json.dumps({"workload": f"{deg}x{deg} deg mosaic, {deg*deg} tiles of 1200x1200", "algorithmic_bytes": bytes_, "sweep": out}))
