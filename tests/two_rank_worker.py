"""One rank of the two-GPU panorama test (tests/test_gpu_parity.py::test_two_rank_panorama_over_rccl): python two_rank_worker.py RANK DIR.
Rank 0 draws the RCCL unique id through the C ABI and leaves it in DIR; both ranks render their half of the panorama with
topo_render_panorama (the overlapped per-slot exchange) and save the whole strip they end up with."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import topo_renderer_amd as T  # noqa: E402
from scenes import Scene  # noqa: E402

rank, d = int(sys.argv[1]), sys.argv[2]
torch.cuda.set_device(rank)
uid_path = os.path.join(d, "uid.npy")
if rank == 0:
    uid = T.comm_unique_id()
    np.save(uid_path + ".tmp.npy", uid)
    os.replace(uid_path + ".tmp.npy", uid_path)
else:
    t0 = time.time()
    while not os.path.exists(uid_path):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.05)
    uid = np.load(uid_path)
sc = Scene(96, 2, 2, eye_dh=120.0)
sw, sh = 96, 160
g = T.TerrainRenderer(sw, sh, device=rank)
sc.load(g)
g.set_stream(torch.cuda.current_stream().cuda_stream)
comm = T.Comm(rank, 2, uid, device=rank)
strip = torch.zeros((8, sh, sw, 4), dtype=torch.uint8, device="cuda")
for _ in range(2):      # twice: the second frame reuses every buffer and event of the first
    strip.zero_()
    g.render_panorama(comm, sc.eye, float(np.radians(25.0)), sw, sh, sc.vlon, sc.vlat, strip.data_ptr(), 0)
    g.synchronize()
    torch.cuda.synchronize()
np.save(os.path.join(d, f"strip{rank}.npy"), strip.cpu().numpy())
comm.close()
print("rank", rank, "done", flush=True)
