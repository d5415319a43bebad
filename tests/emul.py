"""ctypes wrapper over tests/host_emul.cpp (TEST-ONLY g++ build of the product's TOPO_HD pipeline headers)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhost_emul.so")
_LIB = None


class EmulTile(C.Structure):
    _fields_ = [("heights", C.c_void_p), ("normals", C.c_void_p), ("tu", C.c_float * 24)]


def lib():
    global _LIB
    if _LIB is None:
        src = os.path.join(_HERE, "host_emul.cpp")
        hdrs = [os.path.join(_HERE, "..", "topo-renderer_amd", "csrc", f) for f in ("topo_math.h", "topo_pipeline.h", "srgb_tables.h")]
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        if not os.path.exists(_SO) or any(os.path.getmtime(f) > os.path.getmtime(_SO) for f in [src] + hdrs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-shared",
                                   "-o", _SO, src])
        _LIB = C.CDLL(_SO)
        _LIB.emul_render.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class EmulRenderer:
    """Same method set as the product's TerrainRenderer, running the product's header code on the CPU."""

    def __init__(self, width, height, terrain_uniforms_fn):
        self.size = (width, height)
        self.tiles = {}          # geo key -> dict
        self.seq = 0
        self._tu = terrain_uniforms_fn
        self.u = None

    @staticmethod
    def _key(lat, lon):
        return (abs(lat), 1 if lat > 0 else 0, abs(lon), 1 if lon > 0 else 0)

    def _et(self, t):
        e = EmulTile()
        e.heights = t["h"].ctypes.data
        e.normals = t["n"].ctypes.data
        for i in range(24):
            e.tu[i] = float(t["tu"][i])
        return e

    def add_terrain(self, lat, lon, heights, rp, mp, ps):
        hts = np.ascontiguousarray(heights, dtype=np.float32)
        H, W = hts.shape
        t = {"h": hts, "n": np.zeros((H, W), np.uint32), "tu": self._tu(rp, mp, ps, W, H), "lat": lat, "lon": lon}
        L = lib()
        et = self._et(t)
        L.emul_normals_interior(C.byref(et), W, H)
        g = lambda la, lo: self.tiles.get(self._key(la, lo))
        left, right, top, bottom = g(lat, lon - 1), g(lat, lon + 1), g(lat + 1, lon), g(lat - 1, lon)
        tl, tr, bl, br = g(lat + 1, lon - 1), g(lat + 1, lon + 1), g(lat - 1, lon - 1), g(lat - 1, lon + 1)
        E = lambda x: C.byref(self._et(x))
        if left: L.emul_normals_edge(E(left), E(t), E(t), W, H, 0)
        if right: L.emul_normals_edge(E(t), E(right), E(t), W, H, 0)
        if top: L.emul_normals_edge(E(top), E(t), E(t), W, H, 1)
        if bottom: L.emul_normals_edge(E(t), E(bottom), E(t), W, H, 1)
        if tl and top and left: L.emul_normals_corner(E(tl), E(top), E(left), E(t), E(t), W, H)
        if top and tr and right: L.emul_normals_corner(E(top), E(tr), E(t), E(right), E(t), W, H)
        if left and bl and bottom: L.emul_normals_corner(E(left), E(t), E(bl), E(bottom), E(t), W, H)
        if right and bottom and br: L.emul_normals_corner(E(t), E(right), E(bottom), E(br), E(t), W, H)
        self.tiles[self._key(lat, lon)] = t

    def unload_terrain(self, lat, lon):
        self.tiles.pop(self._key(lat, lon), None)

    def update(self, width, height, uniforms, post):
        self.size = (width, height)
        self.u = np.ascontiguousarray(uniforms).view(np.float32).copy()
        self.post = np.ascontiguousarray(post, dtype=np.float32).copy()

    def render(self):
        W, H = self.size
        order = [self.tiles[k] for k in sorted(self.tiles)]
        arr = (EmulTile * max(1, len(order)))(*[self._et(t) for t in order])
        tw, th = (order[0]["h"].shape[1], order[0]["h"].shape[0]) if order else (3, 3)
        rgba = np.empty((H, W, 4), np.uint8)
        depth = np.empty((H, W), np.float32)
        lib().emul_set_post.argtypes = [C.c_float, C.c_float, C.c_float]
        lib().emul_set_post(float(self.post[0]), float(self.post[1]), float(self.post[2]))
        rc = lib().emul_render(arr, len(order), tw, th, _p(self.u), W, H, _p(rgba), _p(depth))
        lib().emul_set_post(0.0, 0.0, 100.0)
        if rc != 0:
            raise RuntimeError("emul_render: winner triangle could not be resolved")
        return rgba, depth

    def read_normals(self, lat, lon):
        t = self.tiles[self._key(lat, lon)]
        return t["n"].view(np.uint8).reshape(t["n"].shape + (4,))
