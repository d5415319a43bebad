#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot run here and holds no
fixtures for this path -- SURVEY.md 8c -- so these vectors freeze the oracle, i.e. the spec).

Each file holds the scene parameters, the raw inputs that matter (uniform blocks) and the expected outputs
(RGBA8, depth bits, normal textures).  Heights are regenerated from the integer-hash synthesiser (seeded).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import topo_renderer_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from scenes import Scene  # noqa: E402

CASES = {
    # name: (tile, n_lat, n_lon, W, H, yaw, pitch, fov, eye_dh)
    "single_64": (64, 1, 1, 128, 64, 0.0, 0.0, 60.0, 50.0),
    "block2x2_24": (24, 2, 2, 128, 64, 200.0, 25.0, 79.28, 120.0),
    "nearfield_16": (16, 2, 2, 64, 32, 77.0, 70.0, 100.0, 300.0),
}


def make(name):
    tile, n_lat, n_lon, W, H, yaw, pitch, fov, dh = CASES[name]
    sc = Scene(tile, n_lat, n_lon, eye_dh=dh)
    o = O.OracleRenderer(W, H)
    sc.load(o)
    out = {"params": np.array([tile, n_lat, n_lon, W, H, yaw, pitch, fov, dh], np.float64),
           "heights_sha": np.frombuffer(__import__("hashlib").sha256(b"".join(sc.heights[l].tobytes() for l in sc.locs)).digest(), np.uint8)}
    for i, loc in enumerate(sc.locs):
        out[f"normals_{i}"] = o.read_normals(loc[0], loc[1], tile, tile)
    for mode in (0, 1, 2):
        u = sc.uniforms(W, H, yaw, pitch, fov, mode)
        o.update(W, H, u, T.post_uniforms(W, H))
        rgba, depth = o.render()
        out[f"uniforms_{mode}"] = u
        out[f"rgba_{mode}"] = rgba
        if mode == 0:
            out["depth_bits"] = depth.view(np.uint32)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        np.savez_compressed(os.path.join(here, name + ".npz"), **make(name))
        print(name, os.path.getsize(os.path.join(here, name + ".npz")), "bytes")
