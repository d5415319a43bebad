"""Scenes that drive every kernel path of the frame (and load) phase, hashed -- run once with the product library and once,
in a child process with TOPO_HIP_LIB pointing at it, with the bounds-checked build (tests/test_gpu_parity.py)."""
from __future__ import annotations

import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


def run(T):
    import torch
    from scenes import Scene
    cases = {}

    def record(name, r, *arrays):
        st = r.frame_status()
        cases[name] = {"sha": _sha(*arrays), "status": st["status"] & 3, "bounds_violation": st["bounds_violation"],
                       "bounds_site": st["bounds_site"], "bounds_value": st["bounds_value"]}

    def host_frame(name, sc, W, H, yaw, pitch, fov, mode=0, caps=None):
        r = T.TerrainRenderer(W, H)
        sc.load(r)
        if caps:
            r.debug_set_queue_caps(*caps)
        r.update(W, H, sc.uniforms(W, H, yaw, pitch, fov, mode), T.post_uniforms(W, H))
        try:
            rgba, depth = r.render()
        except T.TopoError:
            rgba, depth = np.zeros(1), np.zeros(1)          # (rare-queue overflow with an explicit cap: no frame is handed out)
        record(name, r, rgba, depth, *[r.read_normals(*loc) for loc in sc.locs])

    coarse = Scene(12, 2, 2, eye_dh=60.0)                       # every triangle large; many cross the near plane
    host_frame("coarse_down", coarse, 640, 480, 10, 35, 110)
    host_frame("coarse_steep", coarse, 640, 480, 200, 80, 110)
    host_frame("coarse_last_rows", coarse, 333, 257, 100, 5, 110)    # odd target: boxes clamp at the last row / column
    host_frame("big_queue_overflow", coarse, 320, 240, 10, 35, 110, caps=(16, 0))
    host_frame("rare_queue_overflow", coarse, 320, 240, 10, 35, 110, caps=(0, 2))
    host_frame("rare_queue_grown", coarse, 320, 240, 10, 35, 110, caps=(0, 2 | 0x80000000))
    host_frame("fine_far", Scene(256, 1, 1, eye_dh=20), 300, 200, 10, 45, 90)
    host_frame("block_edges", Scene(62, 1, 2), 257, 131, 300, 20, 120, mode=2)
    host_frame("mosaic3x3", Scene(48, 3, 3), 160, 96, 200, 30, 79.28)
    host_frame("below_surface", Scene(64, 2, 2, eye_dh=-300.0), 128, 128, 40, 10, 60)
    host_frame("far_above", Scene(64, 2, 2, eye_dh=250000.0), 128, 128, 40, 89, 60)

    # device entry points: a 64-view submission, then frames in flight, then a full-size config-2 panorama
    def strip(r, views, sw, sh):
        n = len(views)
        rgba = torch.zeros((n, sh, sw, 4), dtype=torch.uint8, device="cuda")
        depth = torch.zeros((n, sh, sw), dtype=torch.float32, device="cuda")
        r.render_views_device(views, sw, sh, rgba.data_ptr(), sh * sw * 4, sw * 4, depth.data_ptr(), sh * sw * 4, sw * 4)
        r.synchronize()
        return rgba.cpu().numpy(), depth.cpu().numpy()

    sc = Scene(48, 2, 2, eye_dh=80)
    r = T.TerrainRenderer(96, 64)
    sc.load(r)
    views = []
    for k in range(8):
        views += sc.panorama(96, 64, yaw0_deg=7.0 * k)
    record("views64", r, *strip(r, views, 96, 64))
    r.set_pipeline_depth(2)
    outs = []
    for k in range(4):
        rgba = torch.zeros((8, 64, 96, 4), dtype=torch.uint8, device="cuda")
        depth = torch.zeros((8, 64, 96), dtype=torch.float32, device="cuda")
        r.render_views_device(views[8 * k:8 * k + 8], 96, 64, rgba.data_ptr(), 64 * 96 * 4, 96 * 4, depth.data_ptr(), 64 * 96 * 4, 96 * 4)
        outs += [rgba, depth]
    r.join()
    record("in_flight", r, *[t.cpu().numpy() for t in outs])
    r.set_pipeline_depth(1)

    c2 = Scene(1200, 1, 1, lat0=40, lon0=10, vfrac=(0.623, 0.717))
    r = T.TerrainRenderer(512, 1024)
    c2.load(r)
    record("config2_panorama", r, *strip(r, c2.panorama(512, 1024), 512, 1024))
    r.set_occlusion_split(0.0)
    record("config2_no_occlusion_filter", r, *strip(r, c2.panorama(512, 1024)[2:5], 512, 1024))
    return {"lib": T.LIB_PATH, "cases": cases}


if __name__ == "__main__":
    import topo_renderer_amd as T
    print(json.dumps(run(T)))
