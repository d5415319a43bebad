#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE (KiB -> bytes) per launch of each kernel from the rocprofv3 --pmc passes.

gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports reads by an access-width dependent factor and
is only calibrated (x2) for 16-B-per-lane streaming reads.  k_raster reads the DEM as 61-lane dword rows, the same
pattern as k_block_minmax, whose unique bytes per launch are known exactly (tile_w*tile_h*4): the ratio
known/FETCH_SIZE of k_block_minmax is used as the correction for k_raster.  WRITE_SIZE is exact for these stores
(k_normals_interior: 5,760,000 B written, 5,760,000 B counted)."""
import collections
import csv
import glob
import json
import os
import sys

root, workload = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for k in ("k_raster_big", "k_raster_rare", "k_raster", "k_resolve", "k_clear", "k_cull", "k_block_minmax", "k_normals_interior"):
            if k + "(" in name or k + "<" in name:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)
                break
med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in agg.items()}
tile_bytes = 1200 * 1200 * 4
cal = tile_bytes / med["k_block_minmax"]["FETCH_SIZE"] if "k_block_minmax" in med else 1.0
kernels = {}
for k, d in med.items():
    f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    # only k_raster's reads (61-lane dword rows) share k_block_minmax's pattern; the other kernels keep the raw count
    fc = f * cal if k == "k_raster" else f
    kernels[k] = {"fetch_size_raw": round(f), "write_size": round(w), "bytes_per_launch": round(fc + w),
                  "fetch_calibrated": k == "k_raster"}
out = {"workload": workload, "n_gpus": 1, "fetch_calibration_dword_rows": round(cal, 4), "kernels": kernels,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), median per launch; k_raster's FETCH_SIZE "
                 "corrected by the k_block_minmax calibration (same 61-lane dword row reads, known bytes); other kernels raw "
                 "(FETCH_SIZE is uncalibrated for their access widths on gfx950)"}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.path.dirname(root), "hbm_traffic.json"), "w"), indent=1)
