"""TEST-ONLY GeoTIFF encoder (classic TIFF 6.0, one f32 sample per pixel) used to build fixtures for the decoder tests.

Independent of the product's decoder: numpy + zlib only.  Options cover what COP90-style files use (Deflate + floating
point predictor, strips or tiles) and the other combinations the decoder accepts."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def _packbits(data: bytes) -> bytes:
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 127:
            j += 1
        run = j - i + 1
        if run >= 3:
            out += bytes([(257 - run) & 0xFF, data[i]])
            i += run
            continue
        k = i
        while k < n and k - i < 128:
            if k + 2 < n and data[k] == data[k + 1] == data[k + 2]:
                break
            k += 1
        out += bytes([k - i - 1]) + data[i:k]
        i = k
    return bytes(out)


def _predict(block: np.ndarray, predictor: int, big: bool) -> bytes:
    """block: (rows, cols) float32 -> the bytes a TIFF writer stores for it (before compression)."""
    rows, cols = block.shape
    if predictor == 3:
        be = block.astype(">f4").view(np.uint8).reshape(rows, cols, 4)      # byte 0 = most significant
        planes = np.ascontiguousarray(be.transpose(0, 2, 1)).reshape(rows, 4 * cols)
        d = planes.copy()
        d[:, 1:] = planes[:, 1:] - planes[:, :-1]                            # uint8 wrap-around
        return d.tobytes()
    if predictor == 2:
        u = block.view(np.uint32).copy()
        u[:, 1:] = block.view(np.uint32)[:, 1:] - block.view(np.uint32)[:, :-1]
        return u.astype(">u4" if big else "<u4").tobytes()
    return block.astype(">f4" if big else "<f4").tobytes()


def write_geotiff(arr: np.ndarray, byteorder="<", compression="deflate", predictor=3, tile=None, rows_per_strip=None,
                  pixel_scale=(1 / 1200, 1 / 1200, 0.0), tie_points=(0, 0, 0, 10.0, 48.0, 0), model_transformation=None,
                  bits=32, sample_format=3) -> bytes:
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    h, w = arr.shape
    big = byteorder == ">"
    comp_code = {"none": 1, "deflate": 8, "adobe_deflate_old": 32946, "packbits": 32773}[compression]
    segs = []
    if tile:
        tw, th = tile
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                blk = np.zeros((th, tw), np.float32)
                part = arr[y0:y0 + th, x0:x0 + tw]
                blk[:part.shape[0], :part.shape[1]] = part
                segs.append(_predict(blk, predictor, big))
    else:
        rps = rows_per_strip or h
        for y0 in range(0, h, rps):
            segs.append(_predict(arr[y0:y0 + rps], predictor, big))
    if comp_code in (8, 32946):
        segs = [zlib.compress(s, 6) for s in segs]
    elif comp_code == 32773:
        segs = [_packbits(s) for s in segs]
    E = byteorder
    tags = []            # (tag, type, values)

    def add(tag, typ, vals):
        tags.append((tag, typ, list(vals)))
    add(256, 4, [w]); add(257, 4, [h]); add(258, 3, [bits]); add(259, 3, [comp_code]); add(262, 3, [1]); add(277, 3, [1])
    add(284, 3, [1]); add(339, 3, [sample_format])
    if predictor != 1:
        add(317, 3, [predictor])
    data_off = 8
    blob = bytearray()
    offs = []
    for s in segs:
        offs.append(data_off + len(blob))
        blob += s
        if len(blob) % 2:
            blob += b"\0"
    if tile:
        add(322, 3, [tile[0]]); add(323, 3, [tile[1]]); add(324, 4, offs); add(325, 4, [len(s) for s in segs])
    else:
        add(278, 4, [rows_per_strip or h]); add(273, 4, offs); add(279, 4, [len(s) for s in segs])
    if pixel_scale is not None:
        add(33550, 12, pixel_scale)
    if tie_points is not None:
        add(33922, 12, tie_points)
    if model_transformation is not None:
        add(34264, 12, model_transformation)
    tags.sort()
    ifd_off = data_off + len(blob)
    fmt = {3: "H", 4: "I", 12: "d"}
    size = {3: 2, 4: 4, 12: 8}
    extra = bytearray()
    extra_off = ifd_off + 2 + 12 * len(tags) + 4
    ifd = bytearray(struct.pack(E + "H", len(tags)))
    for tag, typ, vals in tags:
        raw = struct.pack(E + fmt[typ] * len(vals), *vals)
        if len(raw) <= 4:
            field = raw + b"\0" * (4 - len(raw))
        else:
            field = struct.pack(E + "I", extra_off + len(extra))
            extra += raw
            if len(extra) % 2:
                extra += b"\0"
        ifd += struct.pack(E + "HHI", tag, typ, len(vals)) + field
    ifd += struct.pack(E + "I", 0)
    head = (b"II" if not big else b"MM") + struct.pack(E + "H", 42) + struct.pack(E + "I", ifd_off)
    return bytes(head + blob + ifd + extra)
