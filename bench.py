"""Headline benchmark: frames/s of the forward render path on synthetic random-Gaussian scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one `render_gaussians(..., backend="hip")` forward (project -> bin/sort -> rasterise)
of the whole frame, inputs resident in HBM.  N=1 workload = BASELINE config 3's forward
(1M Gaussians, 1920x1080, the config the metric is quoted on).  N>1 = the SAME frame rendered
as tile-row bands, one band per rank, + an RCCL all-gather of the framebuffer (strong scaling:
total work per step is fixed).  Rank 0 prints ONE JSON line.

`roofline` prices the dominant kernel (the tile rasteriser) with SURVEY.md 8(d)'s algorithmic
bytes (40 B/intersection + 8 B/tile + 12 B/pixel) over its average duration measured with HIP
events on the launch stream inside the timed region (the two events that bracket the kernel, on
every 8th frame of a run of >= 64 steps, since each event between two kernels costs the GPU a bubble;
the project / bin split comes from a short untimed pass with all four stage events).  `cpu_baseline` times the scalar C oracle
(1 core) on ONE frame of the same workload on this box's host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (N, W, H, ell, fp16 colours)
    "cfg2": (100_000, 1920, 1080, -4.0, False),
    "cfg3": (1_000_000, 1920, 1080, -4.0, False),
    "cfg2-heavy": (100_000, 1920, 1080, -3.0, False),
    "cfg3-heavy": (1_000_000, 1920, 1080, -3.0, False),
    "cfg4": (6_000_000, 1600, 1063, -4.0, True),
    "cfg5": (5_000_000, 3840, 2160, -4.0, False),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="also time the multi-view batch entry point (runs two views concurrently, so it "
                         "is kept out of the default run whose rocprof kernel averages must match)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    # Rehearsal only (the N > 1 path on a single-GPU box): MOJOSPLAT_BENCH_REHEARSE=1 puts every rank
    # on device 0 and uses gloo, which moves device tensors; RCCL refuses two ranks on one device.
    rehearse = os.environ.get("MOJOSPLAT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import mojosplat_amd as ms
    from mojosplat_amd import _hip, render as render_mod
    from mojosplat_amd.distributed import render_gaussians_sharded
    from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

    _hip.lib()  # hard failure if the HIP library is missing: nothing below has a CPU fallback
    N, W, H, ell, fp16 = WORKLOADS[args.workload]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    if fp16:
        sc["features"] = sc["features"].half()
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])

    # one untimed pass through the per-stage API for the workload's statistics (N, M, T)
    m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip")
    M, T = int(ids.numel()), int(ranges.shape[0] * ranges.shape[1])
    del m2, con, dep, rad, ids, ranges

    if world == 1:
        def step():
            return ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
    else:
        # frames are independent: frame k's framebuffer all-gather (RCCL's stream) overlaps frame
        # k+1's render; a frame is consumed (its gather awaited) one step later.  The closing
        # barrier() drains the last one, so exactly K complete frames are inside the timed region.
        in_flight = []
        mode = {"async": True}

        def step():
            if not mode["async"]:
                return render_gaussians_sharded(*g, cam, background_color=bg)
            in_flight.append(render_gaussians_sharded(*g, cam, background_color=bg, async_op=True))
            if len(in_flight) > 1:
                in_flight.pop(0).wait()

        # one probe frame before anything is timed: should the pipelined path raise on this node
        # (it cannot be rehearsed with RCCL on the single-GPU build box), every rank falls back to
        # the blocking gather -- the decision is agreed on with an all-reduce so ranks never diverge
        ok = 1
        try:
            step()
            while in_flight:
                in_flight.pop(0).wait()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: pipelined sharded path failed ({e!r}); using blocking gathers", flush=True)
            ok = 0
            in_flight.clear()
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        mode["async"] = bool(flag.item())

    def barrier():
        if world > 1:
            while in_flight:
                in_flight.pop(0).wait()
            dist.barrier()
        torch.cuda.synchronize()

    binning = None
    if world == 1:
        # render_gaussians settles its binning granularity by racing the modes over a scene's first
        # frames (render.py, _BinTuner): let that finish before the W warm-up steps, untimed
        for _ in range(24):
            step()
            tuners = list(render_mod._BIN_CHOICE.values())
            if tuners and not tuners[0].queue:
                binning = {"chosen_bin_px": tuners[0].choice,
                           "race_ms": {str(k): round(v * 1e3, 4) for k, v in tuners[0].times.items()}}
                break
    for _ in range(args.warmup):
        step()

    # In-situ kernel timing: ms_render_fwd records HIP events on the launch stream (torch's current
    # stream, whose handle is what every ms_* call is given).  Inside the timed region only the
    # two that bracket the dominant kernel (the rasteriser) are recorded -- every event between two
    # kernels costs the GPU a ~6 us bubble -- and the full stage breakdown comes from a short
    # untimed pass afterwards.
    stage_events = []
    pool = []

    def make_events(n):
        out = []
        for _ in range(n):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for e in evs:
                e.record()   # recorded once, so the hipEvent_t exists
            out.append(evs)
        torch.cuda.synchronize()
        return out

    pool = make_events(args.steps)   # created outside the timed region

    # N > 1: this rank's band of the frame (for its share of the algorithmic bytes)
    band_stats = None
    if world > 1:
        from mojosplat_amd.binning import bin_gaussians_to_tiles_hip
        from mojosplat_amd.distributed import band_plan
        th_, tw_ = -(-H // 16), -(-W // 16)
        _, bands_ = band_plan(th_, world)
        r0_, r1_ = bands_[rank]
        m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
        ids_b, _ = bin_gaussians_to_tiles_hip(m2, rad, dep, 16, tw_, th_, row_range=(r0_, r1_))
        band_stats = dict(M=int(ids_b.numel()), T=(r1_ - r0_) * tw_, px=(min(r1_ * 16, H) - min(r0_ * 16, H)) * W)
        del m2, con, dep, rad, ids_b

    # Every event between two kernels costs the GPU a bubble (measured: the pair around the rasteriser on
    # every frame costs 7-8 us per frame, 3 % of the headline): long runs instrument every 8th frame of the
    # timed region, short ones every frame.  `avg_kernel_us` is the mean over the instrumented launches.
    every = 8 if args.steps >= 64 else 1
    calls = [0]

    def hook():
        calls[0] += 1
        if calls[0] % every:
            return None
        evs = pool.pop()
        stage_events.append(evs)
        return [None, None, evs[2], evs[3]]

    render_mod._STAGE_HOOK = hook   # (N > 1: the sharded entry point consults the same hook)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    render_mod._STAGE_HOOK = None

    breakdown = []
    if world == 1:   # untimed: all four stage boundaries
        extra = make_events(20)

        def hook_all():
            evs = extra.pop()
            breakdown.append(evs)
            return evs
        render_mod._STAGE_HOOK = hook_all
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        render_mod._STAGE_HOOK = None

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    fps = args.steps / dt

    stage_us = {}
    if stage_events:
        raster = [evs[2].elapsed_time(evs[3]) * 1e3 for evs in stage_events]
        stage_us["raster"] = sum(raster) / len(raster)          # inside the timed region
        if breakdown:
            for i, n in enumerate(("project", "bin", "raster_untimed_pass")):
                v = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for evs in breakdown]
                stage_us[n] = sum(v) / len(v)

    out = None
    if rank == 0:
        b_raster = (40 - (6 if fp16 else 0)) * M + 8 * T + 12 * H * W
        if band_stats is not None:   # N > 1: the kernel timed is rank 0's band
            b_raster = (40 - (6 if fp16 else 0)) * band_stats["M"] + 8 * band_stats["T"] + 12 * band_stats["px"]
        b_frame = 96 * N + (84 - (6 if fp16 else 0)) * M + 12 * T + 12 * H * W
        roofline = None
        if "raster" in stage_us:
            ach = b_raster / (stage_us["raster"] * 1e-6) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and world == 1:   # measured on the whole frame
                traffic = json.load(open(tpath)).get(args.workload, {}).get("rasterize_fwd_bytes")
            roofline = {"bound": "hbm", "kernel": "k_rasterize_fwd" + ("" if world == 1 else " (rank 0's band)"),
                        "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "algorithmic_bytes": b_raster,
                        "avg_kernel_us": round(stage_us["raster"], 1), "instrumented_launches": len(stage_events),
                        "alpha_evals": 256 * M,
                        "frame": {"algorithmic_bytes": b_frame,
                                  "achieved": round(b_frame / (ms_per_step * 1e-3) / 1e9, 1),
                                  "frac": round(b_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                        "stage_us": {k: round(v, 1) for k, v in stage_us.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sc, cam, W, H, BACKGROUND_V1, args.workload)
        extras = None
        if world == 1 and args.extras:
            # not the headline: the same scene through the multi-view entry point (16 cameras per
            # call, two views in flight on two streams), reported beside the single-call rate
            cams = [cam] * 16
            ms.render_gaussians_batch(*g, cams, background_color=bg)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(4):
                ms.render_gaussians_batch(*g, cams, background_color=bg)
            torch.cuda.synchronize()
            extras = {"multi_view_batch16_views_per_s": round(64 / (time.perf_counter() - tb), 1)}
        out = {
            "metric": "frames/sec at 1M Gaussians 1920x1080 fwd; achieved HBM GB/s vs peak",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)" if rehearse else ""),
            "config": {"workload": f"{args.workload}: randscene-v1 N={N} {W}x{H} ell={ell} seed=42 forward",
                       "gaussians": N, "intersections": M, "tiles": T, "tile_size": 16, "binning": binning,
                       "colour_dtype": "f16" if fp16 else "f32",
                       "parallelism": "single GPU" if world == 1 else f"{world} tile-row bands + RCCL all-gather" + (", gather of frame k overlapped with render of frame k+1" if mode["async"] else " (blocking)")},
            "roofline": roofline, "cpu_baseline": cpu, "extras": extras,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, cam, W, H, bg, workload):
    """The scalar C oracle (kind 'port', 1 core) on a bounded sample of the same workload: whole
    frames of the scene (its first <= 1M Gaussians), repeated until ~10 s of CPU work, reported
    as frames/s of that sample (stated in `sample`)."""
    import numpy as np

    import oracle
    cpu = {k: v.float().cpu().numpy() for k, v in sc.items()}
    vm = cam.view_matrix.cpu().numpy()
    n = min(len(cpu["means3d"]), 1_000_000)
    args = tuple(cpu[k][:n] for k in ("means3d", "scales", "quats", "opacities", "features"))
    frames, t0 = 0, time.perf_counter()
    while True:
        _, aux = oracle.render_fwd(*args, vm, cam.fx, cam.fy, cam.cx, cam.cy, W, H,
                                   background=np.array(bg, np.float32))
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or frames >= 8:
            break
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{frames} frame(s) of {workload} restricted to its first {n} Gaussians "
                      f"(M={aux['M']}), oracle/gsplat_oracle.c, {dt:.1f} s of CPU work",
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
