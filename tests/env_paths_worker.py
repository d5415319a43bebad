"""One process of tests/test_gpu_parity.py::test_fall_back_paths_render_the_same_frames: renders a fixed set of frames through the C ABI
with whatever TOPO_* switches its environment carries (they are read once per process) and prints a SHA-256 over every frame's RGBA
bytes, depth bits and counters.  python env_paths_worker.py"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import topo_renderer_amd as T  # noqa: E402
from scenes import Scene  # noqa: E402

h = hashlib.sha256()
# a lone tile around the viewpoint (the far phase is provably empty) and a mosaic seen from its corner with the split inside it (it is not)
for tile, n, vfrac, split in ((96, 1, (0.52, 0.48), None), (64, 3, (0.1, 0.12), 20000.0)):
    sc = Scene(tile, n, n, vfrac=vfrac, eye_dh=80.0)
    W, H = 160, 96
    g = T.TerrainRenderer(W, H)
    sc.load(g)
    if split is not None:
        g.set_occlusion_split(split)
    for slots in (None, ("resolve",), ()):
        for total in (True, False):
            g.set_timing_slots(slots, total=total)
            for yaw, pitch, mode in ((10.0, 3.0, 0), (200.0, 25.0, 1), (95.0, -4.0, 2)):
                g.update(W, H, sc.uniforms(W, H, yaw, pitch, 70.0, mode), T.post_uniforms(W, H))
                rgba, depth = g.render()
                c = g.counters()
                h.update(rgba.tobytes())
                h.update(depth.view(np.uint32).tobytes())
                h.update(repr(sorted((k, v) for k, v in c.items())).encode())
                g.timings()
print("SHA256", h.hexdigest())
