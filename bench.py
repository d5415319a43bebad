#!/usr/bin/env python3
"""bench.py -- panorama Mpix/s + achieved HBM GB/s of the MI355X terrain path (BASELINE.json metric).

One "step" = one complete 360-degree panorama: the 8 fixed 45-degree sectors (each a full reference frame:
clear, cull, raster, resolve + contour post) rendered from the resident DEM mosaic, plus -- for N > 1 -- the
RCCL all-gather of the per-rank strips.  Tiles and their normal textures are resident in HBM before the timed
region (the reference computes normals once per tile load, terrain_renderer.rs:192-202); the load phase is
timed separately and reported as `load_ms`.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N GPUs shard the panorama by azimuth sector (rank g renders sectors [8g/N, 8(g+1)/N)), DEM replicated, so
the total work is fixed: scaling = "strong".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (mosaic degrees per side, panorama width, panorama height)      BASELINE.json configs
    "c1": (1, 1024, 256),
    "c2": (1, 4096, 1024),
    "c3": (5, 8192, 2048),
    "c4": (10, 16384, 4096),
    # config 5 (throughput mode): a batch of 1024 random viewpoints over the c4 mosaic, one 4096x1024 panorama each,
    # sharded by VIEWPOINT across the ranks (no collective).  Not the default: a parity-test / throughput case.
    "c5": (10, 4096, 1024),
}
C5_VIEWPOINTS = 1024
LAT0, LON0 = 40, 10            # mosaic SW corner
TILE = 1200
N_SECTORS = 8
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--view-mode", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="verify sector 0 against the oracle (slow)")
    ap.add_argument("--occlusion-split", type=float, default=None, help="metres; 0 disables the two-phase occlusion filter")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="frames in flight (topo_set_pipeline_depth) in the timed region; default 1 = strictly one panorama "
                         "after the other (per-kernel durations are then those of the kernel alone); 2 overlaps consecutive "
                         "panoramas (profiles/: +18 %% throughput at c4)")
    ap.add_argument("--also-pipelined", action="store_true", help="(the default since round 3; kept for the scripts that pass it)")
    ap.add_argument("--no-pipelined-extra", action="store_true",
                    help="skip the extra pass after the timed region that times the same panoramas with 2 frames in flight (reported under \"pipelined\", never `value`)")
    ap.add_argument("--pitch", type=float, default=0.0, help="camera pitch in radians (reference: positive looks down)")
    ap.add_argument("--no-host-path", action="store_true", help="skip timing topo_render (host outputs, PCIe-inclusive; an extra key, never `value`)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE, SQ counters) behind roofline.traffic / roofline_valu")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # this process IS such a pass: a few frames, no JSON
    args = ap.parse_args()
    if args.pmc_child:
        args.no_cpu_baseline = args.no_pmc = args.no_host_path = args.no_pipelined_extra = True

    import numpy as np
    import torch
    import topo_renderer_amd as T

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if N_SECTORS % world != 0:
        raise SystemExit("the sector count (8) must be divisible by the GPU count")
    # TOPO_BENCH_BACKEND=gloo is a rehearsal switch only (several ranks sharing one GPU on a single-GPU box); the
    # real multi-GPU run is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("TOPO_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    deg, PW, PH = WORKLOADS[args.workload]
    SW = PW // N_SECTORS
    n_tiles = deg * deg
    locs = T.synth.mosaic_locations(LAT0, LON0, deg, deg)

    # ---- synthetic COP90-shaped mosaic, uploaded once (not timed: inputs are resident before the timed region)
    t0 = time.time()
    r = T.TerrainRenderer(SW, PH, device=local_rank)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.occlusion_split is not None:
        r.set_occlusion_split(args.occlusion_split)
    depth_frames = args.pipeline if args.pipeline is not None else 1
    vlat, vlon = LAT0 + deg / 2 + 0.123, LON0 + deg / 2 + 0.217      # mosaic centre + (0.123, 0.217) degrees
    ground = None
    upload_s = 0.0
    for (la, lo) in locs:
        h = T.synth_tile(la, lo, TILE, TILE)
        if la == int(math.floor(vlat)) and lo == int(math.floor(vlon)):
            ground = T.synth.height_at(h, la, lo, vlon, vlat)
        t1 = time.time()
        r.add_terrain(la, lo, h, *T.synth.tile_transform(la, lo, TILE, TILE))
        upload_s += time.time() - t1
    r.synchronize()
    setup_s = time.time() - t0
    eye = T.geometry_transform(ground + 50.0, vlon, vlat)            # render_engine.rs:327
    views = T.panorama_uniforms(eye, 0.0, SW, PH, vlon, vlat, args.view_mode, pitch=args.pitch)
    if args.workload == "c5":
        return bench_batch(args, T, np, torch, dist, r, locs, deg, SW, PH, rank, world, setup_s)

    # ---- load phase (normals K1-K3 over resident heights), timed on its own
    # (every load-time kernel of the resident tiles is inside the bracket: the per-tile tables of the frame phase -- block
    # min/max, cull bounds, sin/cos tables: k_block_tables -- and the normals K1-K3)
    loads = []
    for _ in range(3):
        r.recompute_normals()
        tm = r.timings()
        loads.append((tm["load"], tm["load_tables"]))
    load_ms, load_tables_ms = min(loads)
    load_normals_ms = load_ms - load_tables_ms

    # ---- outputs: the sector-major strip of the C ABI (topo_render_panorama): strip[8][PH][SW][4], rank g's sectors
    # [8g/N, 8(g+1)/N) one contiguous block.  A rank renders ALL its sectors in one submission (one set of kernel launches:
    # a single view fills the chip worse) and the strip is assembled by ONE in-place all-gather, ordered after the frame
    # by the stream: the renderer runs on torch's current stream (set_stream above) and the collective waits for that stream.
    per = N_SECTORS // world
    my = list(T.panorama.sector_range(rank, world))
    if world > 1:
        depth_frames = 1            # frames in flight run on the contexts' own streams, which the collective would not wait for
    # N > 1: the exchange behind the C ABI by default (topo_render_panorama: the frame resolved slot by slot, each slot shipped
    # over RCCL under the next one's resolve); TOPO_BENCH_COMM=torch = one in-place torch.distributed all-gather after the frame
    use_capi_comm = world > 1 and os.environ.get("TOPO_BENCH_COMM", "capi") == "capi"
    comm, comm_note = None, None
    if use_capi_comm:
        # RCCL is bound by libtopo_hip.so itself; the unique id travels over torch.distributed
        try:
            uid = torch.from_numpy(T.comm_unique_id() if rank == 0 else np.zeros(128, np.uint8)).cuda()
            ok = torch.ones(1, device="cuda")
        except T.TopoError as e:
            uid, ok, comm_note = torch.zeros(128, dtype=torch.uint8, device="cuda"), torch.zeros(1, device="cuda"), str(e)
        dist.broadcast(uid, 0)
        dist.broadcast(ok, 0)
        if ok.item() > 0:
            try:
                comm = T.Comm(rank, world, uid.cpu().numpy(), device=local_rank)
            except T.TopoError as e:
                ok.zero_()
                comm_note = f"topo_comm_init failed on rank {rank}: {e}"
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            use_capi_comm = False      # (every rank falls back together)
            comm = None
    # one output set per frame in flight
    r.set_pipeline_depth(depth_frames)
    outs = [(torch.empty((N_SECTORS, PH, SW, 4), dtype=torch.uint8, device="cuda"),
             torch.empty((N_SECTORS, PH, SW), dtype=torch.float32, device="cuda")) for _ in range(depth_frames)]
    strip, depth_all = outs[0]
    depth = depth_all[my[0]:my[0] + per]
    frame_no = [0]

    def step():
        nonlocal strip, depth, depth_all
        strip, depth_all = outs[frame_no[0] % depth_frames]
        depth = depth_all[my[0]:my[0] + per]
        frame_no[0] += 1
        if use_capi_comm:
            r.render_panorama(comm, eye, 0.0, SW, PH, vlon, vlat, strip.data_ptr(), 0, args.view_mode, args.pitch)
            return
        r.render_views_device([views[k] for k in my], SW, PH, strip[my[0]].data_ptr(), PH * SW * 4, SW * 4,
                              depth.data_ptr(), PH * SW * 4, SW * 4)
        if world > 1:
            T.panorama.gather_sector_major(dist, strip, rank, world)      # RGBA only: the depth stays with its rank

    def fence():
        r.join()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- N > 1: the C-ABI exchange has only ever run with a world of one (one-GPU boxes).  Its first frame is therefore a trial:
    # every rank renders one panorama through it; if any rank gets an error, or the ranks do not end up with byte-identical
    # strips, ALL ranks fall back together to the torch.distributed all-gather and the line says so (`exchange_note`).
    if use_capi_comm:
        ok = torch.ones(1, device="cuda")
        try:
            step()
            r.synchronize()
            torch.cuda.synchronize()
        except T.TopoError as e:
            ok.zero_()
            comm_note = f"topo_render_panorama failed on rank {rank}: {e}"
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() > 0:
            # (a cheap fingerprint of the whole strip: 64-bit sums of its words, position-weighted)
            w = strip.view(torch.int32).to(torch.int64).flatten()
            fp = torch.stack([w.sum(), (w * (torch.arange(w.numel(), device="cuda") % 8191 + 1)).sum()])
            lo, hi = fp.clone(), fp.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                ok.zero_()
                comm_note = "the ranks' strips differ after topo_render_panorama"
        if ok.item() == 0:
            use_capi_comm = False
            if comm_note is None:
                comm_note = "topo_render_panorama failed on another rank"
            try:
                r.synchronize()
            except T.TopoError:
                pass
            torch.cuda.synchronize()
            dist.barrier()

    # ---- per-kernel breakdown, from a few frames with every timing event on.  Each event between two kernels idles the GPU
    # for ~6 us, so the timed region below keeps only the two events around the dominant kernel: its duration there is what
    # the roofline uses.  (The frame's GPU span -- an event in front of its first kernel, one behind its last -- comes from a
    # pass of its own: the event in front costs the frame ~7 us.)
    KERNELS = ("clear", "cull", "raster", "occlusion", "raster_big", "resolve")
    r.set_timing_slots(None)
    for _ in range(max(1, args.warmup)):       # first launches, buffer growth
        step()
    fence()
    samples = []
    for _ in range(7 + depth_frames):
        step()
        tm = r.timings()
        if tm["total"] > 0.0:
            samples.append({k: tm[k] for k in KERNELS + ("total",)})
    fence()
    n_detail = len(samples)
    detail = {k: (sorted(smp[k] for smp in samples)[n_detail // 2] if n_detail else 0.0) for k in KERNELS + ("total",)}   # medians
    dom = max(("resolve", "raster", "raster_big"), key=lambda k: detail[k])
    r.set_timing_slots(())
    n_span = 8 * depth_frames
    for _ in range(n_span):
        step()
    fence()
    span = sorted(h["total"] for h in r.timing_history(n_span) if h["total"] > 0.0)
    r.set_timing_slots((dom,), total=False)
    for _ in range(args.warmup):
        step()
    fence()
    if args.pmc_child:              # a rocprofv3 --pmc pass of this command: the frames above and a few more are all it needs
        for _ in range(3):
            step()
        fence()
        return
    # The timed region: exactly `steps` submissions between two fences, nothing in it waits for a frame.  The HIP-event
    # durations of its frames' kernels (the events sit on the stream the kernels are launched on) are read AFTER the region
    # from the renderer's event ring (topo_get_timing_history: the last 32 frames per frame in flight).
    # Even two events cost a frame 8-10 us of GPU time (markers or the kernel's own start / end times alike), so only every
    # EV_STRIDE-th frame of the region carries them: the dominant kernel's duration is the mean over those frames.
    EV_STRIDE = 4
    t_start = time.perf_counter()
    for i in range(args.steps):
        if i % EV_STRIDE < 2:
            r.set_timing_slots((dom,) if i % EV_STRIDE == 0 else (), total=False)
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    hist = [h for h in r.timing_history(min(args.steps, 32 * depth_frames)) if h[dom] > 0.0]
    timed_frames = len(hist)
    timed_ms = {dom: (sum(h[dom] for h in hist) / timed_frames if timed_frames else 0.0)}       # mean: what the roofline uses
    timed_median = {dom: (sorted(h[dom] for h in hist)[timed_frames // 2] if timed_frames else 0.0)}
    kernel_ms = dict(detail)                 # the table: all events on (a separate pass) ...
    kernel_ms[dom] = timed_ms[dom]           # ... except the dominant kernel (the timed region's) and the total (the span pass's)
    kernel_ms["total"] = span[len(span) // 2] if span else 0.0
    counters = r.counters()

    mpix = PW * PH / 1e6
    value = mpix / (ms_per_step / 1e3)

    # ---- roofline of the dominant kernel = the frame kernel with the largest mean duration (HIP events on the launch
    # stream).  Algorithmic bytes per launch:
    #   k_resolve : 8 B visibility key read + 4 B RGBA8 + 4 B depth written per pixel of this rank's sectors;
    #   k_raster  : 4 B per DEM texel of the mosaic share this rank is responsible for ("DEM read" = 4*T*W*H / N) --
    #               since the occlusion filter most of those texels are ruled out from the 8-byte-per-block min/max
    #               table instead of being read, so this prices the stage, not the bytes the kernel touches.
    #
    # SURVEY.md 8(d) prices the path by what the REFERENCE's data flow needs, not by this design's intermediates:
    #   resolve stage  : "write 4*P RGBA + 4*P depth" = 8 B per pixel  (the 8 B visibility key this design also reads is its own)
    #   whole frame    : "read 4*T*W*H (DEM once) + write 8*P"
    # `roofline.achieved` / `frac` use those 8(d) bytes; the line keeps the design's own count (16 B per pixel incl. the key)
    # as `design_bytes_per_launch` / `frac_design_bytes`.
    my_pixels = per * SW * PH
    alg = {"resolve": ("k_resolve", 8.0 * my_pixels, 16.0 * my_pixels),
           "raster": ("k_raster", 4.0 * n_tiles * TILE * TILE / world, 4.0 * n_tiles * TILE * TILE / world),
           "raster_big": ("k_raster_rare+k_raster_big", 8.0 * my_pixels, 8.0 * my_pixels)}
    dom_name, dom_bytes, dom_design = alg[dom]
    dom_s = kernel_ms[dom] / 1e3
    achieved = dom_bytes / dom_s / 1e9 if dom_s > 0 else 0.0
    frame_bytes_8d = 4.0 * n_tiles * TILE * TILE / world + 8.0 * my_pixels
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                "algorithmic_bytes_8d": dom_bytes, "avg_launch_ms": round(kernel_ms[dom], 4),
                "algorithmic_bytes_what": "SURVEY 8(d): 4 B RGBA8 + 4 B depth written per pixel of this rank's sectors" if dom == "resolve" else "SURVEY 8(d)",
                "design_bytes_per_launch": dom_design,
                "frac_design_bytes": round(dom_design / dom_s / 1e9 / HBM_PEAK_GBPS, 5) if dom_s > 0 else None,
                "frame_algorithmic_bytes_8d": frame_bytes_8d,
                "frame_frac_8d": round(frame_bytes_8d / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBPS, 5)}
    # HBM bytes per launch and the SQ instruction counters cannot be read from inside this process: they come from separate
    # `rocprofv3 --kernel-trace --pmc <counter>` passes of THIS command, run as child processes right here (rank 0, N = 1),
    # one counter group per pass as MI355X_MICROARCH.md prescribes.  FETCH_SIZE is corrected by the factor calibrated for
    # this access width (profiles/r02_calibration.json: 2.0 for 4- and 8-byte-per-lane reads on gfx950; raw value kept).
    roofline_valu = None
    if rank == 0 and world == 1 and not args.no_pmc:
        pmc = collect_pmc(args, dom_name.split("+")[-1])
        if pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
            fetch_factor = 2.0
            roofline["traffic"] = round(pmc["FETCH_SIZE"] * fetch_factor + pmc["WRITE_SIZE"])
            roofline["traffic_detail"] = {"FETCH_SIZE_raw": round(pmc["FETCH_SIZE"]), "fetch_correction": fetch_factor,
                                          "WRITE_SIZE": round(pmc["WRITE_SIZE"]), "launches_averaged": pmc.get("launches"),
                                          "source": "rocprofv3 --kernel-trace --pmc passes of this command, run by this process as children"}
        else:
            roofline["traffic_note"] = pmc.get("error", "PMC passes unavailable")
        roofline_valu = valu_roofline(pmc, kernel_ms[dom])
    dem_bytes = 4.0 * n_tiles * TILE * TILE / world
    stage_ms = kernel_ms["cull"] + kernel_ms["raster"] + kernel_ms["occlusion"] + kernel_ms["raster_big"]
    per_kernel = {name: {"ms": round(kernel_ms[k], 4), "algorithmic_GBps": round(b / (kernel_ms[k] / 1e3) / 1e9, 1) if kernel_ms[k] > 0 else None}
                  for k, (name, b, _design) in alg.items()}
    per_kernel["dem_to_visibility_stage"] = {"ms": round(stage_ms, 4), "what": "cull + raster + occlusion + rare + big against the DEM read (4*T*W*H/N)",
                                             "algorithmic_GBps": round(dem_bytes / (stage_ms / 1e3) / 1e9, 1) if stage_ms > 0 else None}

    out = {
        "metric": "panorama Mpix/s",
        "value": round(value, 2),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {deg}x{deg} deg COP90-shaped mosaic ({n_tiles} tiles of 1200x1200 f32), "
                               f"{PW}x{PH} panorama = 8 sectors of {SW}x{PH}, view_mode {args.view_mode}",
                   "frames_in_flight": depth_frames,
                   "sharding": f"azimuth sectors, {per} per GPU in one submission, DEM replicated" +
                               (", one in-place RCCL all-gather of the sector-major RGBA strip per panorama" +
                                (" through the C ABI (topo_render_panorama)" if use_capi_comm else " (torch.distributed)") if world > 1 else "")
                               if not use_capi_comm or world == 1 else
                               f"azimuth sectors, {per} per GPU in one cull/raster submission, DEM replicated; resolved and exchanged in "
                               f"{len(T.panorama_slots(world, SW, PH))} slots through the C ABI (topo_render_panorama: grouped ncclSend/ncclRecv of each "
                               f"slot on a second stream under the next slot's resolve)",
                   **({"comm_note": comm_note} if comm_note else {})},
        "roofline": roofline,
        "roofline_valu": roofline_valu,
        "per_kernel": per_kernel,
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
        "kernel_ms_note": f"'{dom}': HIP events in the timed region (the kernel's own start and end time, hipExtLaunchKernel, on every {EV_STRIDE}th "
                          f"frame of the region = {timed_frames} frames; no other event is recorded there); 'total': first to last "
                          f"event of a frame, median of a pass of {len(span)} frames with only those two events; the other "
                          f"kernels: medians of a separate pass of {n_detail} frames with all timing events on (total then {round(detail['total'], 4)} ms); "
                          "'clear' = k_clear_cull, the visibility clear and the cull side by side in one launch ('cull' = the empty event gap behind it; "
                          "TOPO_FUSE_CLEAR_CULL=0 launches them one after the other)",
        "gpu_ms_per_step": {"median": round(kernel_ms["total"], 4), "mean": round(sum(span) / len(span), 4) if span else 0.0, "frames": len(span),
                            dom + "_median": round(timed_median[dom], 4), dom + "_frames": timed_frames,
                            "what": "first to last HIP event of a frame, from a pass in front of the timed region with only those two events "
                                    f"(topo_get_timing_history); {dom}: the timed region's frames"},
        "load_ms": round(load_ms, 4),
        "load_what": "every load-time kernel of the resident tiles, the DEM read ONCE: k_trig_tables (sin/cos tables) -> k_normals_rolling<4,4,true> (interior normals K1 "
                     "+ the raster blocks' min/max heights in the same pass) -> k_block_bounds (f64 cull bounds from them) -> k_normals_border (K2, K3); "
                     "with TOPO_LOAD_FUSED=0: k_block_tables (a second DEM read) + the normals kernels",
        "load_fused": os.environ.get("TOPO_LOAD_FUSED", "1") != "0",
        "load_tables_ms": round(load_tables_ms, 4),
        "load_normals_ms": round(load_normals_ms, 4),
        "load_normals_GBps": round(8.0 * n_tiles * TILE * TILE / (load_normals_ms / 1e3) / 1e9, 1) if load_normals_ms > 0 else None,
        "load_normals_frac_hbm_peak": round(8.0 * n_tiles * TILE * TILE / (load_normals_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4) if load_normals_ms > 0 else None,
        "load_GBps": round(8.0 * n_tiles * TILE * TILE / (load_ms / 1e3) / 1e9, 1) if load_ms > 0 else None,
        "load_tables_what": "the kernels in front of the normals pass (fused path: k_trig_tables alone; separate path: k_block_tables); load_normals_ms = the rest "
                            "(fused path: normals + block min/max + f64 bounds + seams)",
        "load_GBps_what": "SURVEY 8(d) load-phase bytes (4 B read + 4 B written per texel) over the WHOLE load phase, this design's own tables included",
        "add_terrain_ms_per_tile": round(1e3 * upload_s / n_tiles, 3),
        "add_terrain_what": "topo_add_terrain per tile from pageable host memory: one pooled allocation, the 5.76 MB upload, tables + normals + seam passes, one stream sync",
        "hbm_read_roofline_frac_frame": round((4.0 * n_tiles * TILE * TILE / (ms_per_step / 1e3) / 1e9) / HBM_PEAK_GBPS, 5),
        "counters": counters,
        "terrain_pixel_frac": round(float((depth < 1.0).float().mean().item()), 4),   # this rank's sectors
        "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
        "setup_s": round(setup_s, 1),
        "upload_s": round(upload_s, 2),
    }

    # ---- throughput mode (additive, never `value`): consecutive panoramas with two frames in flight, so that the
    # latency-bound cull/raster phases of one run under the ALU-bound resolve of the previous one
    if world == 1 and not args.no_pipelined_extra and depth_frames == 1:
        r.set_timing_slots((), total=False)
        r.set_pipeline_depth(2)
        outs2 = outs + [(torch.empty_like(outs[0][0]), torch.empty_like(outs[0][1]))]
        def step2(i):
            s2, d2 = outs2[i % 2]
            r.render_views_device(views, SW, PH, s2.data_ptr(), PH * SW * 4, SW * 4, d2.data_ptr(), PH * SW * 4, SW * 4)
        for i in range(max(2, args.warmup)):
            step2(i)
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        fence()
        dt = (time.perf_counter() - t1) / args.steps
        out["pipelined"] = {"frames_in_flight": 2, "ms_per_step": round(dt * 1e3, 4), "value": round(mpix / dt, 2), "unit": "Mpix/s",
                            "what": "the same panoramas back to back with topo_set_pipeline_depth(2); kernels of adjacent frames overlap"}
        r.set_pipeline_depth(1)
        del outs2

    # ---- the drop-in entry point hands back HOST buffers: its PCIe-inclusive rate (never `value`; measured before the CPU baseline,
    # whose OpenMP threads keep spinning on the cores for a while after their last region), one sector-sized frame
    # (RGBA8 + pad_256-pitched depth) per call: into fresh pageable arrays (through the context's pinned staging image) and
    # into arrays the caller pinned once (topo_pin_host_buffer: direct copies)
    if rank == 0 and world == 1 and not args.no_host_path:
        r.set_stream(0)
        r.set_pipeline_depth(1)
        r.update(SW, PH, views[0], T.post_uniforms(SW, PH))
        hrgba = np.empty((PH, SW, 4), np.uint8)
        hdepth = np.zeros((PH, T.pad_256(4 * SW) // 4), np.float32)
        nbytes = hrgba.nbytes + hdepth.nbytes

        def timed(n=5, batches=3):      # best of three batches of five calls (the host's copy threads share the box's cores with whatever else runs there)
            r.render_into(hrgba, hdepth)
            best = None
            for _ in range(batches):
                t1 = time.perf_counter()
                for _ in range(n):
                    r.render_into(hrgba, hdepth)
                dt = (time.perf_counter() - t1) / n
                best = dt if best is None or dt < best else best
            return best
        dt_staged = timed()
        r.pin_host_buffer(hrgba)
        r.pin_host_buffer(hdepth)
        dt_pinned = timed()
        r.unpin_host_buffer(hrgba)
        r.unpin_host_buffer(hdepth)
        out["topo_render_host_path"] = {
            "what": f"one {SW}x{PH} frame through topo_render: render + device-to-host copies of RGBA8 and pad_256-pitched depth ({round(nbytes / 1e6, 1)} MB)",
            "pageable": {"ms": round(dt_staged * 1e3, 3), "GBps": round(nbytes / dt_staged / 1e9, 1), "mpix_s": round(SW * PH / 1e6 / dt_staged, 1),
                         "how": "pinned staging image of the context, slices copied on by host threads"},
            "pinned_by_caller": {"ms": round(dt_pinned * 1e3, 3), "GBps": round(nbytes / dt_pinned / 1e9, 1), "mpix_s": round(SW * PH / 1e6 / dt_pinned, 1),
                                 "how": "topo_pin_host_buffer: direct copies"}}

    # ---- CPU baseline: the oracle (a C++ port of the reference's shaders + raster semantics; the reference's
    # own wgpu CPU-adapter path cannot be built here) on a bounded sample, rank 0 at N=1 only.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T, np, locs, vlat, vlon, views, SW, PH, args.cpu_threads)

    if args.check and rank == 0:
        mine = strip[my[0]:my[0] + per]
        out["check"] = check_against_oracle(T, np, locs, views, my, mine, depth, SW, PH)

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


# issue cost of the VALU instruction classes, ns per wave64 instruction per SIMD at full occupancy: measured by
# tools/calib.hip (profiles/r02_calibration.json, HIP-event wall clock): full-rate f32 fma/mul/add, integer add, moves,
# logic 1.0-1.2; conversions, comparisons/selects, min/max/med3, floor/fract, shifts, 24/32-bit integer multiplies, 64-bit
# adds 1.7-1.9; rcp/sqrt/rsq 3.4-3.5
VALU_NS = {"full": 1.12, "half": 1.78, "trans": 3.45}
# One SIMD starts at most one instruction of ANY kind (vector, scalar, LDS, memory) per ~1.05 ns: tools/calib.hip's mixed
# streams (8 waves per SIMD) take 2.16 ns per v_fma_f32 + s_add_u32 pair and 2.08 ns per v_max_f32 + s_add_u32 pair, against
# 1.13 / 1.73 / 1.76 ns for the three instructions alone -- scalar instructions are not free beside vector ones.
ISSUE_SLOT_NS = 1.05


def valu_roofline(pmc, kernel_ms):
    """VALU issue time of the dominant kernel's instruction mix (SQ counters of a --pmc pass) over its duration x 1024 SIMDs."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT",
            "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32")
    if any(pmc.get(k) is None for k in need) or kernel_ms <= 0:
        return {"note": pmc.get("error", "SQ counter pass unavailable")}
    total, trans = pmc["SQ_INSTS_VALU"], pmc["SQ_INSTS_VALU_TRANS_F32"]
    f32 = pmc["SQ_INSTS_VALU_FMA_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_ADD_F32"]
    half = pmc["SQ_INSTS_VALU_INT64"] + pmc["SQ_INSTS_VALU_CVT"] + 0.5 * pmc["SQ_INSTS_VALU_INT32"]      # (INT32 = adds and multiplies/shifts: priced half and half)
    other = max(0.0, total - trans - f32 - pmc["SQ_INSTS_VALU_INT32"] - pmc["SQ_INSTS_VALU_INT64"] - pmc["SQ_INSTS_VALU_CVT"])
    full = f32 + 0.5 * pmc["SQ_INSTS_VALU_INT32"] + 0.5 * other      # (other = moves/logic at full rate, compares/selects/min/max at half: split)
    half += 0.5 * other
    issue_ns = full * VALU_NS["full"] + half * VALU_NS["half"] + trans * VALU_NS["trans"]
    out = {"kernel_ms": round(kernel_ms, 4), "valu_insts_per_launch": round(total), "class_split": {"full_rate": round(full), "half_rate": round(half), "transcendental": round(trans)},
           "issue_ns_per_wave_inst": VALU_NS, "valu_issue_ms": round(issue_ns / 1024 / 1e6, 4),
           "frac": round(issue_ns / 1024 / 1e6 / kernel_ms, 4),
           "what": "sum over instruction classes of (wave-instructions x issue cost) / (1024 SIMDs x kernel duration); costs from tools/calib.hip"}
    others = ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
    if all(pmc.get(k) is not None for k in others):
        # every instruction takes an issue slot of its SIMD; the vector ones also their execution time
        n_all = total + sum(pmc[k] for k in others)
        slots_ns = n_all * ISSUE_SLOT_NS
        bound_ns = max(slots_ns, issue_ns)
        out["issue"] = {"salu_insts_per_launch": round(pmc["SQ_INSTS_SALU"]), "lds_insts_per_launch": round(pmc["SQ_INSTS_LDS"]),
                        "vmem_insts_per_launch": round(pmc["SQ_INSTS_VMEM_RD"] + pmc["SQ_INSTS_VMEM_WR"]), "slot_ns": ISSUE_SLOT_NS,
                        "issue_slots_ms": round(slots_ns / 1024 / 1e6, 4), "bound_ms": round(bound_ns / 1024 / 1e6, 4),
                        "frac": round(bound_ns / 1024 / 1e6 / kernel_ms, 4),
                        "what": "max(all wave-instructions x one issue slot, vector execution time) / (1024 SIMDs x kernel duration): branches and waits not counted"}
    return out


def collect_pmc(args, kernel):
    """Per-launch medians of the --pmc counters of `kernel`, from rocprofv3 child passes of this very command."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    groups = ["FETCH_SIZE", "WRITE_SIZE",
              "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32",
              "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="topo_pmc_")
    try:
        for gi, grp in enumerate(groups):
            d = os.path.join(tmp, str(gi))
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp.split(), "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--view-mode", str(args.view_mode),
                   "--warmup", "1", "--steps", "1"]
            if args.occlusion_split is not None:
                cmd += ["--occlusion-split", str(args.occlusion_split)]
            try:
                subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240, check=True)
            except (subprocess.SubprocessError, OSError) as e:
                tail = (getattr(e, "stderr", None) or b"").decode("utf-8", "replace").strip().split("\n")[-3:]
                out["error"] = f"rocprofv3 --pmc {grp.split()[0]}... failed: {type(e).__name__} {getattr(e, 'returncode', '')} {' | '.join(t[-160:] for t in tail)}"
                continue
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]      # "void topo::(anonymous namespace)::k_resolve<true, false>(topo::FrameParams, ...)"
                    if kernel + "(" in name or kernel + "<" in name or name.rstrip().endswith(kernel):
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c, v in vals.items():
                v.sort()
                out[c] = v[len(v) // 2] * (1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 1.0)      # the two sizes are reported in KiB
                out["launches"] = len(v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x, z ^ (z >> 31)


C5_GROUP = 8          # viewpoints per submission (x 8 sectors = 64 views, the per-submission limit)


def bench_batch(args, T, np, torch, dist, r, locs, deg, SW, PH, rank, world, setup_s):
    """BASELINE config 5: 1024 viewpoints (splitmix64 seed 0x5EED0005; lat/lon uniform in the inner 8x8 degrees, yaw uniform
    in [0, 2pi)), 4096x1024 each = 8 sectors of 512x1024; rank g renders viewpoints [g*1024/N, (g+1)*1024/N)."""
    state, vps = 0x5EED0005, []
    tiles = {}
    for _ in range(C5_VIEWPOINTS):
        u = []
        for _k in range(3):
            state, z = _splitmix64(state)
            u.append(z / 2.0 ** 64)
        lat, lon, yaw = LAT0 + 1.0 + 8.0 * u[0], LON0 + 1.0 + 8.0 * u[1], 2.0 * math.pi * u[2]
        key = (int(math.floor(lat)), int(math.floor(lon)))
        if key not in tiles:
            tiles[key] = T.synth_tile(key[0], key[1], TILE, TILE)
        ground = T.synth.height_at(tiles[key], key[0], key[1], lon, lat)
        vps.append((T.geometry_transform(ground + 50.0, lon, lat), yaw, lon, lat))
    per = C5_VIEWPOINTS // world
    mine = vps[rank * per:(rank + 1) * per]
    # throughput mode through the C ABI's batch entry (topo_render_batch): the viewpoints are independent, so 8 of them
    # (x 8 sectors = 64 views) go into one submission -- one set of kernel launches instead of eight -- and several
    # submissions are kept in flight (topo_set_pipeline_depth); outputs are written chunk after chunk into the same buffers
    in_flight = args.pipeline if args.pipeline is not None else 2
    r.set_pipeline_depth(in_flight)
    r.set_timing_slots((), total=False)  # no timing events at all (each one idles the GPU a few microseconds)
    CHUNK = 64                           # viewpoints per topo_render_batch call (1 GiB of RGBA + 1 GiB of depth)
    eyes = np.stack([v[0] for v in mine]).astype(np.float32)
    yaws = np.array([v[1] for v in mine], np.float32)
    suns = np.array([(v[2], v[3]) for v in mine], np.float32)
    rgba = torch.empty((min(CHUNK, len(mine)), N_SECTORS, PH, SW, 4), dtype=torch.uint8, device="cuda")
    depth = torch.empty((min(CHUNK, len(mine)), N_SECTORS, PH, SW), dtype=torch.float32, device="cuda")

    def step():      # one step = this rank's whole share of the batch
        for c0 in range(0, len(mine), CHUNK):
            r.render_batch(eyes[c0:c0 + CHUNK], yaws[c0:c0 + CHUNK], suns[c0:c0 + CHUNK], SW, PH, rgba.data_ptr(), depth.data_ptr(), args.view_mode)

    def fence():
        r.join()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(min(args.warmup, 1)):
        step()
    fence()
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = 1e3 * elapsed / steps
    out = {"metric": "panorama Mpix/s", "value": round(C5_VIEWPOINTS * N_SECTORS * SW * PH / 1e6 / (ms / 1e3), 2), "unit": "Mpix/s",
           "n_gpus": world, "steps": steps, "warmup": min(args.warmup, 1), "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"c5: batch of {C5_VIEWPOINTS} viewpoints over the {deg}x{deg} deg mosaic, {N_SECTORS * SW}x{PH} panorama each",
                      "frames_in_flight": in_flight, "viewpoints_per_submission": C5_GROUP, "entry_point": "topo_render_batch",
                      "sharding": f"viewpoints, {per} per GPU, DEM replicated, no collective"},
           "ms_per_viewpoint": round(ms / per, 4), "setup_s": round(setup_s, 1)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _nearest_tiles(locs, vlat, vlon, k):
    return sorted(locs, key=lambda l: (l[0] + 0.5 - vlat) ** 2 + (l[1] + 0.5 - vlon) ** 2)[:k]


def usable_cores():
    """The cores this process may actually run on: the affinity mask, capped by the cgroup's CPU quota (a GPU box shows all
    256 host cores but grants a share of them; more threads than that only slow the baseline down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(T, np, locs, vlat, vlon, views, SW, PH, threads):
    """The oracle (kind "port": the reference's own wgpu CPU-adapter path cannot be built here) over the FULL workload -- one
    whole panorama -- on ALL host cores: with more cores than sectors each sector's tiles are split into runs of the draw
    order, every (sector, run) job fills a z-buffer of its own and the runs are merged per pixel (oracle_render_views_tiled;
    same bytes).  Capped at 128 threads: 67 MB of z-buffer per job at c4."""
    from oracle import oracle as O
    host = os.cpu_count() or 1
    cores = threads or min(usable_cores(), 128)
    groups = max(1, min(len(locs), (cores + N_SECTORS - 1) // N_SECTORS))
    sample = _nearest_tiles(locs, vlat, vlon, len(locs) if os.environ.get('TOPO_CPU_SAMPLE_TILES') is None else int(os.environ['TOPO_CPU_SAMPLE_TILES']))
    o = O.OracleRenderer(SW, PH)
    for (la, lo) in [l for l in locs if l in sample]:          # keep the insertion order
        o.add_terrain(la, lo, T.synth_tile(la, lo, TILE, TILE), *T.synth.tile_transform(la, lo, TILE, TILE))
    o.update(SW, PH, views[0], T.post_uniforms(SW, PH))
    t0 = time.perf_counter()
    o.render_views_tiled(views, threads=cores, groups=groups)
    dt = time.perf_counter() - t0
    return {"value": round(N_SECTORS * SW * PH / 1e6 / dt, 3), "unit": "Mpix/s", "cores": cores, "host_cores": host, "usable_cores": usable_cores(), "kind": "port",
            "seconds": round(dt, 2),
            "sample": f"one full panorama: all 8 sectors at full size ({N_SECTORS * SW}x{PH}) over {len(sample)} of the {len(locs)} tiles; "
                      f"oracle/topo_oracle.cpp, {cores} OpenMP threads over {N_SECTORS} sectors x {groups} runs of tiles"}


def check_against_oracle(T, np, locs, views, my, mine, depth, SW, PH):
    """Every sector this rank rendered, at full size over the full mosaic, against the oracle (bit for bit)."""
    from oracle import oracle as O
    o = O.OracleRenderer(SW, PH)
    for (la, lo) in locs:
        o.add_terrain(la, lo, T.synth_tile(la, lo, TILE, TILE), *T.synth.tile_transform(la, lo, TILE, TILE))
    o.update(SW, PH, views[my[0]], T.post_uniforms(SW, PH))
    ro, do = o.render_views([views[k] for k in my], threads=min(len(my), os.cpu_count() or 1))
    rg, dg = mine.cpu().numpy(), depth.cpu().numpy()
    return {"sectors": list(my), "rgba_mismatch": int((ro != rg).any(axis=-1).sum()),
            "depth_mismatch": int((do.view(np.uint32) != dg.view(np.uint32)).sum()),
            "terrain_fraction": round(float((do < 1).mean()), 4)}


if __name__ == "__main__":
    main()
