import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import topo_renderer_amd as T
r = T.TerrainRenderer(8, 8)
bad = []
chunk = 1 << 26
for start in range(0, 1 << 32, chunk):
    u = np.arange(start, start + chunk, dtype=np.uint64).astype(np.uint32)
    x = u.view(np.float32)
    got = r.probe_div(4, x, x)
    with np.errstate(all="ignore"):
        ref = x - np.floor(x)
    ok = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    if not ok.all():
        i = np.nonzero(~ok)[0]
        bad.append((hex(start), len(i), float(x[i].min()), float(x[i].max()), [(float(x[k]), float(got[k]), float(ref[k])) for k in i[:3]]))
for b in bad: print(b)
print("chunks with mismatches:", len(bad))
