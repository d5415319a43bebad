"""N > 1 path on CPU: two gloo ranks each produce their azimuth sectors, all-gather the sector-major strip with
the same helper bench.py uses, and must end up with the single-process panorama."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, expect_path, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import topo_renderer_amd as T
    from oracle import oracle as O
    from scenes import Scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = Scene(24, 2, 2)
        sw, sh = 16, 32
        views = sc.panorama(sw, sh, yaw0_deg=5.0)
        mine = T.panorama.sector_range(rank, world)
        o = O.OracleRenderer(sw, sh)          # stand-in producer: the CPU suite has no GPU to render with
        sc.load(o)
        o.update(sw, sh, views[0], T.post_uniforms(sw, sh))
        rgba, _ = o.render_views([views[k] for k in mine], threads=1)
        strip = torch.zeros(T.panorama.strip_shape(world, sh, sw), dtype=torch.uint8)
        works = []
        for c in range(len(mine)):                 # one slot at a time, as bench.py pipelines it
            strip[c, rank] = torch.from_numpy(rgba[c])
            works.append(T.panorama.gather_slot(dist, strip, c, rank, world, async_op=True))
        for w in works:
            w.wait()
        expect = torch.from_numpy(np.load(expect_path))          # [8, H, SW, 4] in sector order
        ok = all(bool(torch.equal(T.panorama.sector(strip, k), expect[k])) for k in range(8))
        ok = ok and bool(torch.equal(T.panorama.to_row_major(strip), expect.permute(1, 0, 2, 3).reshape(sh, 8 * sw, 4)))
        # the sector-major layout of the C ABI (topo_render_panorama) and of bench.py: one in-place all-gather per panorama
        strip2 = torch.zeros((8, sh, sw, 4), dtype=torch.uint8)
        for c, k in enumerate(mine):
            strip2[k] = torch.from_numpy(rgba[c])
        T.panorama.gather_sector_major(dist, strip2, rank, world)
        ok = ok and bool(torch.equal(strip2, expect))
        ok = ok and list(mine) == list(T.panorama_sector_range(rank, world))      # the C ABI's split is the same split
        # topo_render_panorama's overlapped exchange: the frame resolved and shipped slot by slot (the C ABI's own plan,
        # topo_panorama_slots, for a sector size that yields several bands per sector), every slot a point-to-point exchange
        # of one band with every other rank, straight into place, in the same order on every rank
        big_w, big_h = 2048, 4096                                # the plan of the c4 sector: bands of whole block rows, ~8 MiB each
        plan = T.panorama_slots(world, big_w, big_h)
        per = 8 // world
        ok = ok and len(plan) == per * 4 and all(r0 % 32 == 0 for _, r0, _ in plan) and sorted(set(s for s, _, _ in plan)) == list(range(per))
        ok = ok and all(sum(r for s, _, r in plan if s == k) == big_h for k in range(per)) and T.panorama_slots(1, big_w, big_h) == [(k, 0, big_h) for k in range(8)]
        os.environ["TOPO_PANORAMA_BAND_BYTES"] = str(sw * 4 * 40)     # the test's own (tiny) sectors, stretched, with a band size that cuts them
        plan = T.panorama_slots(world, sw, 4 * sh)
        del os.environ["TOPO_PANORAMA_BAND_BYTES"]
        ok = ok and len(plan) == per * 4 and [r for _, _, r in plan[:4]] == [32, 32, 32, 32]
        tall = torch.from_numpy(np.load(expect_path)).repeat(1, 4, 1, 1)          # [8, 4 sh, sw, 4]
        strip3 = torch.zeros_like(tall)
        reqs = []
        for slot in plan:                                        # "resolve" a slot, then ship it while the next one is produced
            k = mine[0] + slot[0]
            strip3[k, slot[1]:slot[1] + slot[2]] = tall[k, slot[1]:slot[1] + slot[2]]
            reqs += T.panorama.exchange_slot(dist, strip3, slot, rank, world)
        for r in reqs:
            r.wait()
        ok = ok and bool(torch.equal(strip3, tall))
        out_q.put((rank, ok, list(mine)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sector_sharding_allgather_gloo(tmp_path, world):
    sys.path.insert(0, ROOT)
    import topo_renderer_amd as T
    from oracle import oracle as O
    from scenes import Scene
    sc = Scene(24, 2, 2)
    sw, sh = 16, 32
    views = sc.panorama(sw, sh, yaw0_deg=5.0)
    o = O.OracleRenderer(sw, sh)
    sc.load(o)
    o.update(sw, sh, views[0], T.post_uniforms(sw, sh))
    full, _ = o.render_views(views, threads=2)
    path = str(tmp_path / "expect.npy")
    np.save(path, full)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sorted(s for _, _, ss in res for s in ss) == list(range(8))


def test_sector_range_and_row_major(topo):
    P = topo.panorama
    assert [list(P.sector_range(r, 4)) for r in range(4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert list(P.sector_range(0, 1)) == list(range(8))
    with pytest.raises(ValueError):
        P.sector_range(0, 3)
    for world in (1, 2, 4, 8):
        shape = P.strip_shape(world, 2, 3)
        strip = np.zeros(shape, np.int32)
        for k in range(8):
            c, r = P.slot_of(k, world)
            strip[c, r] = k + 1
            assert (P.sector(strip, k) == k + 1).all()
        rm = P.to_row_major(strip)
        assert rm.shape == (2, 24, 4)
        for k in range(8):
            assert (rm[:, 3 * k:3 * k + 3] == k + 1).all()
        assert torch.equal(P.to_row_major(torch.from_numpy(strip)), torch.from_numpy(rm))
