#!/usr/bin/env python3
"""tests/golden/c4_sector_sha256.json: SHA-256 of every sector of the headline panorama (BASELINE config 4: 10 x 10 degree
mosaic, 16384 x 4096 = 8 sectors of 2048 x 4096, view mode 0) as the CPU ORACLE renders it -- RGBA bytes and depth bits --
so that the GPU suite can hold the whole panorama to the oracle without paying the oracle's minutes per run.

Run in the build container (no GPU needed; ~1.3 GB of host memory, a few minutes on 8 cores):
    python tests/golden/make_c4_hashes.py
The scene is bench.py's: same tiles (topo_synth_tile), same viewpoint, same cameras (topo_panorama_uniforms)."""
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def c4_scene(T):
    """(locs, views, eye, SW, PH) of bench.py's default workload."""
    import bench
    deg, PW, PH = bench.WORKLOADS["c4"]
    SW = PW // bench.N_SECTORS
    locs = T.synth.mosaic_locations(bench.LAT0, bench.LON0, deg, deg)
    vlat, vlon = bench.LAT0 + deg / 2 + 0.123, bench.LON0 + deg / 2 + 0.217
    la, lo = int(math.floor(vlat)), int(math.floor(vlon))
    ground = T.synth.height_at(T.synth_tile(la, lo, bench.TILE, bench.TILE), la, lo, vlon, vlat)
    eye = T.geometry_transform(ground + 50.0, vlon, vlat)
    views = T.panorama_uniforms(eye, 0.0, SW, PH, vlon, vlat, 0, pitch=0.0)
    return locs, views, eye, SW, PH, bench.TILE


def main():
    import numpy as np
    import topo_renderer_amd as T
    from oracle import oracle as O
    locs, views, eye, SW, PH, TILE = c4_scene(T)
    o = O.OracleRenderer(SW, PH)
    for (la, lo) in locs:
        o.add_terrain(la, lo, T.synth_tile(la, lo, TILE, TILE), *T.synth.tile_transform(la, lo, TILE, TILE))
    o.update(SW, PH, views[0], T.post_uniforms(SW, PH))
    t0 = time.time()
    cores = os.cpu_count() or 1
    rgba, depth = o.render_views_tiled(views, threads=cores, groups=max(1, cores // 8))
    out = {"what": "SHA-256 of each sector of bench.py's default panorama (c4, view mode 0) as oracle/topo_oracle.cpp renders it",
           "generator": "tests/golden/make_c4_hashes.py", "sector_w": SW, "sector_h": PH, "n_tiles": len(locs),
           "terrain_fraction": round(float((depth < 1).mean()), 4),
           "rgba": [hashlib.sha256(np.ascontiguousarray(rgba[k]).tobytes()).hexdigest() for k in range(len(views))],
           "depth": [hashlib.sha256(np.ascontiguousarray(depth[k]).tobytes()).hexdigest() for k in range(len(views))]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c4_sector_sha256.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(f"oracle: {time.time() - t0:.0f} s; terrain fraction {out['terrain_fraction']}")


if __name__ == "__main__":
    main()
