"""Headline benchmark: frames/s of the forward render path on synthetic random-Gaussian scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3]

A step = one `render_gaussians(..., backend="hip")` forward (project -> bin/sort -> rasterise)
of the whole frame, inputs resident in HBM.  N=1 workload = BASELINE config 3's forward
(1M Gaussians, 1920x1080, the config the metric is quoted on).  N>1 = the SAME frame rendered
as tile-row bands, one band per rank, + an RCCL all-gather of the framebuffer (strong scaling:
total work per step is fixed).  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process,
before anything in this process has touched a GPU), relays rank 0's JSON line and exits with the
child's return code; launched under torch.distributed.run it is simply one of the ranks.

After the timed loop the frame that was timed is VERIFIED: rendered once more through the same
call and compared bit for bit with the per-stage path (project -> bin -> rasterise through the
per-stage C entry points: gsplat-exact, fully sorted lists), and -- N=1 -- against the C oracle's
frame under the test-suite's bar (tests/helpers.py check_image_strict).  A mismatch fails the run.

`roofline` prices the dominant kernel (the tile rasteriser) with SURVEY.md 8(d)'s algorithmic
bytes (40 B/intersection + 8 B/tile + 12 B/pixel) over its average duration measured with HIP
events on the launch stream inside the timed region (the two events that bracket the kernel, on
every 8th frame of a run of >= 64 steps, since each event between two kernels costs the GPU a bubble;
the project / bin split comes from a short untimed pass with all four stage events).  The
intersection count M is that of the lists the timed kernel is GIVEN (tight binning drops pairs that
cannot blend; `algorithmic_bytes_gsplat_M` keeps round 1's figure on gsplat's M for comparison).
`cpu_baseline` times the scalar C oracle (1 core) on frames of the same workload on this box's host
and, beside it, the package's backend="torch" projection on CPU tensors (the reference's CPU path).

`value` is K steps over the barrier-bracketed wall time of exactly those K steps (the contract);
`ms_per_step` is the MEDIAN of the per-step periods (host time stamps taken as each call returns -- the
host is paced by every frame's size record, so in steady state a period is a frame) when K >= 16, and
`ms_per_step_mean` the plain wall time / K beside it (SURVEY 8(d) asks for a median).

N = 1 also runs, after the timed region and unless --no-extras (what the rocprofv3 summaries under profiles/
are taken with, so that their kernel averages are those of the headline frame alone):
  extras.orbit          256 frames of the same scene from a camera on an orbit (a new view matrix every frame):
                        frames/s, the miss counters of the sync-free path, four sampled frames verified bit for
                        bit against the per-stage path;
  extras.cfg3_fwd_bwd   BASELINE config 3 as it is named (forward + backward, grads for means / scales / quats /
                        opacities / colours): ms per step, stage times from HIP events, rooflines of the two
                        backward kernels on SURVEY 8(d)'s bytes;
  extras.two_frames_in_flight, extras.morton_order   the asynchronous entry point; the same scene sorted along a Morton curve.
N > 1 runs, unless --no-extras: extras.cfg5 (BASELINE config 5 on the N ranks' bands: frames/s, the same frame on one GPU in
the same run, the bands bit-identical to it; round 5: the same on a PREPARED scene and with a float16 exchange) and
extras.view_sharded (whole views per rank, one all-gather per call).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle leg of the verification")
    ap.add_argument("--extras", action="store_true",
                    help="N=1: also time the multi-view batch entry point and config 5 (kept out of the default "
                         "run, whose rocprof kernel averages must match the timed workload); N>1 runs always "
                         "report config 5 (the config BASELINE names for 8 GPUs) under `extras`")
    ap.add_argument("--no-extras", action="store_true",
                    help="N>1: skip the config-5 leg; N=1: skip the moving-camera and forward+backward legs")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the ranks as a child process.
    Nothing in THIS process has initialised a GPU (torch is not even imported yet)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in proc.stdout:          # relay; rank 0's JSON line is printed once more, last
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks exited 0 without a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    from mojosplat_amd import _fused, _hip, render as render_mod
    from mojosplat_amd.distributed import render_gaussians_sharded
    from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

    _hip.lib()  # hard failure if the HIP library is missing: nothing below has a CPU fallback
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    in_flight = []
    mode = {"async": True}

    def barrier():
        if world > 1:
            while in_flight:
                in_flight.pop(0).wait()
            dist.barrier()
        torch.cuda.synchronize()

    def load(name):
        N, W, H, ell, fp16 = WORKLOADS[name]
        sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
        if fp16:
            sc["features"] = sc["features"].half()
        return sc, cam, (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])

    def make_step(g, cam, **kw):
        if world == 1:
            return lambda: ms.render_gaussians(*g, cam, background_color=bg, backend="hip")

        # frames are independent: frame k's framebuffer all-gather (RCCL's stream) overlaps frame
        # k+1's render; a frame is consumed (its gather awaited) one step later.  The closing
        # barrier() drains the last one, so exactly K complete frames are inside the timed region.
        def step():
            if not mode["async"]:
                return render_gaussians_sharded(*g, cam, background_color=bg, **kw)
            in_flight.append(render_gaussians_sharded(*g, cam, background_color=bg, async_op=True, **kw))
            if len(in_flight) > 1:
                return in_flight.pop(0).wait()
        return step

    def stagewise(g, cam):
        """The per-stage path: gsplat-exact binning, fully sorted lists.  -> (image, M, T)."""
        m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
        ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, cam.H, cam.W, 16, backend="hip")
        img = ms.rasterize_gaussians(m2, con, g[4], g[3], bg.to(g[4].dtype), ranges, ids, cam, tile_size=16,
                                     backend="hip")
        return img, int(ids.numel()), int(ranges.shape[0] * ranges.shape[1])

    def timed_fps(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return steps / float(t.item())

    N, W, H, ell, fp16 = WORKLOADS[args.workload]
    sc, cam, g = load(args.workload)
    step = make_step(g, cam)

    # one untimed pass through the per-stage API: the workload's statistics (N, M, T) and the frame the
    # timed path has to reproduce
    ref_img, M, T = stagewise(g, cam)

    if world > 1:
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

    # render_gaussians picks its binning granularity from the previous frame's size record (render.py,
    # bin_rule): the first frame of a scene is a split frame, the second already runs on the rule's grid
    for _ in range(3):
        step()
    binning = None
    if world == 1:
        binning = {"chosen_bin_px": render_mod._bin_mode.get(render_mod._bin_key(g[0], cam), 16),
                   "how": "rule on the previous frame's size record (footprint diameter, density); no timing"}
    # In-situ kernel timing: ms_render_fwd records HIP events on the launch stream (torch's current
    # stream, whose handle is what every ms_* call is given).  Inside the timed region only the
    # two that bracket the dominant kernel (the rasteriser) are recorded -- every event between two
    # kernels costs the GPU a ~6 us bubble -- and the full stage breakdown comes from a short
    # untimed pass afterwards.
    stage_events = []

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
    # timed region (the driver's 20-step run: 2 launches), runs under 16 steps every frame.
    # `avg_kernel_us` is the mean over the instrumented launches.
    every = 8 if args.steps >= 16 else 1
    calls = [0]

    def hook():
        calls[0] += 1
        if calls[0] % every:
            return None
        evs = pool.pop()
        stage_events.append(evs)
        return [None, None, evs[2], evs[3]]

    # Everything that idles the GPU (event creation, the band statistics above) is done: now the untimed frames.
    # A short run (the driver's 20 steps are 4 ms) otherwise starts on a chip whose clock has not ramped yet:
    # spin-up frames, then the W warm-up steps the contract asks for, then the barrier and the timed loop.
    # (round 4: until two consecutive blocks of 64 frames agree to 1.5 % -- at least 128 frames, at most 2 048: how long
    # the ramp takes differs between boxes, and anything that idles the GPU for ~10 ms before the region -- a cyclic GC
    # pass of the interpreter was measured -- costs the next frames 10 %)
    spin_blocks, prev_block, settled = 0, None, 0
    while spin_blocks < 32:
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        for _ in range(64):
            step()
        torch.cuda.synchronize(dev)
        tb = time.perf_counter() - tb
        spin_blocks += 1
        settled = settled + 1 if prev_block is not None and abs(tb - prev_block) <= 0.015 * prev_block else 0
        prev_block = tb
        # (N > 1: every rank must make the same calls -- the frames carry collectives -- so a fixed four blocks there)
        if (spin_blocks >= 2 and settled >= 1 and world == 1) or (world > 1 and spin_blocks >= 4):
            break
    for _ in range(args.warmup):
        step()
    render_mod._STAGE_HOOK = hook   # (N > 1: the sharded entry point consults the same hook)
    barrier()
    stamps = [0.0] * (args.steps + 1)
    t0 = stamps[0] = time.perf_counter()
    for k in range(args.steps):
        step()
        stamps[k + 1] = time.perf_counter()   # (the call returns once the frame's size record is in: one frame behind the GPU)
    barrier()
    dt = time.perf_counter() - t0
    render_mod._STAGE_HOOK = None
    periods = sorted(b - a for a, b in zip(stamps[1:-1], stamps[2:]))   # (the first period holds the pipeline's fill)
    # where a short run's wall time goes: the first call (pipeline fill), the periods, the drain at the closing barrier
    period_us = {"first_call": round((stamps[1] - stamps[0]) * 1e6, 1), "drain": round((t0 + dt - stamps[-1]) * 1e6, 1)}
    if periods:
        period_us.update(min=round(periods[0] * 1e6, 1), median=round(periods[len(periods) // 2] * 1e6, 1),
                         max=round(periods[-1] * 1e6, 1), over_1p5x_median=sum(1 for p_ in periods if p_ > 1.5 * periods[len(periods) // 2]))

    # ---- verification of the timed path (untimed) -------------------------------------------------
    if world == 1:
        img = step()
    else:
        img = render_gaussians_sharded(*g, cam, background_color=bg)    # blocking form of the same frame
    max_abs = float((img.float() - ref_img.float()).abs().max())
    verified = bool(torch.equal(img, ref_img))
    # (bit 6 of the size record's flag word: the frame dropped the pairs behind its bins' depth cut-offs)
    depth_cut = bool(world == 1 and (dev, 0) in _fused._state and int(_fused._state[(dev, 0)]["host_np"][7]) & 64)
    # the lists the timed kernel was given: the frame's tile (or block) ranges still sit in lane 0's workspace
    m_lists = None
    pairs_on_grid = None   # every (Gaussian, bin) pair of the frame's own binning grid, kept or dropped (the size record's word 0)
    if world == 1:
        try:
            px = binning["chosen_bin_px"]   # 16 = split frame: the ranges are those of the 16x16-block lists
            pairs_on_grid = int(_fused._state[(dev, 0)]["host_np"][0])
            m_lists = _fused.last_frame_list_entries(dev, N, -(-W // px), -(-H // px))
        except Exception as e:  # noqa: BLE001
            print(f"[bench] list-entry count unavailable: {e!r}", file=sys.stderr)
    del ref_img

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

    band_bounds_end = None
    if world > 1:
        from mojosplat_amd.distributed import band_bounds
        band_bounds_end = band_bounds(g[0], cam, 16, world)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    vflag = torch.tensor([1 if verified else 0], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(vflag, op=dist.ReduceOp.MIN)
    dt = float(t.item())
    verified = bool(vflag.item())
    ms_mean = dt / args.steps * 1e3
    fps = args.steps / dt
    # the median per-step period (K >= 16, one GPU: the ranks of a sharded run are paced by their collectives, and
    # the contract's max-over-ranks wall time is the figure there)
    ms_per_step = ms_mean
    if args.steps >= 16 and world == 1 and periods:
        ms_per_step = periods[len(periods) // 2] * 1e3

    stage_us = {}
    if stage_events:
        raster = [evs[2].elapsed_time(evs[3]) * 1e3 for evs in stage_events]
        stage_us["raster"] = sum(raster) / len(raster)          # inside the timed region
        if breakdown:
            for i, n in enumerate(("project", "bin", "raster_untimed_pass")):
                v = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for evs in breakdown]
                stage_us[n] = sum(v) / len(v)

    # ---- N = 1, after the timed region: a moving camera, and config 3 as BASELINE names it (forward + backward) ----
    legs = {}
    if world == 1 and not args.no_extras:
        legs["orbit"] = orbit_leg(ms, _fused, render_mod, g, cam, bg, stagewise, ms_mean)
        legs["two_frames_in_flight"] = two_in_flight_leg(ms, g, cam, bg, img, args.steps)
        legs["morton_order"] = morton_leg(ms, _fused, g, cam, bg, img, ms_per_step)
        if not fp16:
            legs["cfg3_fwd_bwd" if args.workload == "cfg3" else args.workload + "_fwd_bwd"] = \
                fwd_bwd_leg(_fused, render_mod, g, cam, bg, N, W, H, T, dev, m_gsplat=M)
        verified = verified and legs["orbit"]["sampled_frames_bit_identical_to_stagewise"]

    # ---- extras: config 5 (the config BASELINE names for 8 GPUs) ----------------------------------
    extras = None
    if args.workload != "cfg5" and (world > 1 or args.extras) and not args.no_extras:
        del sc, g
        torch.cuda.empty_cache()
        sc5, cam5, g5 = load("cfg5")
        step5 = make_step(g5, cam5)
        for _ in range(3):
            step5()
        fps5 = timed_fps(step5, 30, 5)
        extras = {"cfg5": {"workload": "cfg5: randscene-v1 N=5000000 3840x2160 ell=-4.0 seed=42 forward",
                           "frames_per_s": round(fps5, 2), "n_gpus": world, "steps": 30, "scaling": "strong"}}
        if world > 1:
            # the same frame on ONE GPU in the same run (rank 0 renders the whole frame, the others wait):
            # the denominator of config 5's speed-up, and the frame the banded one must equal
            img5 = render_gaussians_sharded(*g5, cam5, background_color=bg)
            one = None
            if rank == 0:
                full5 = ms.render_gaussians(*g5, cam5, background_color=bg, backend="hip")
                extras["cfg5"]["bands_equal_single_gpu_frame"] = bool(torch.equal(img5, full5))
                for _ in range(5):
                    ms.render_gaussians(*g5, cam5, background_color=bg, backend="hip")
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(30):
                    ms.render_gaussians(*g5, cam5, background_color=bg, backend="hip")
                torch.cuda.synchronize()
                one = 30 / (time.perf_counter() - tb)
                extras["cfg5"]["frames_per_s_1_gpu_same_run"] = round(one, 2)
                extras["cfg5"]["speedup_vs_1_gpu"] = round(fps5 / one, 3)
                verified = verified and extras["cfg5"]["bands_equal_single_gpu_frame"]
            barrier()
            # not the headline: round 5's two options for the band path on this very frame -- a PREPARED scene (Morton order +
            # block bounds: the band pre-cull skips blocks, the count kernel's gathers coalesce; a one-time preprocessing
            # of the scene, untimed) and a float16 EXCHANGE (each rank rounds its band once; half the bytes over xGMI; the
            # image comes back in float16)
            try:
                from mojosplat_amd.scene_order import prepare_scene
                gp = prepare_scene(*g5).arrays
                stepp = make_step(gp, cam5)
                for _ in range(4):
                    stepp()
                extras["cfg5"]["prepared_scene_frames_per_s"] = round(timed_fps(stepp, 30, 5), 2)
                step16 = make_step(gp, cam5, exchange_dtype=torch.float16)
                for _ in range(4):
                    step16()
                extras["cfg5"]["prepared_scene_f16_exchange_frames_per_s"] = round(timed_fps(step16, 30, 5), 2)
                img16 = render_gaussians_sharded(*gp, cam5, background_color=bg, exchange_dtype=torch.float16)
                if rank == 0:
                    fullp = ms.render_gaussians(*gp, cam5, background_color=bg, backend="hip")
                    extras["cfg5"]["f16_exchange_is_the_rounded_frame"] = bool(torch.equal(img16, fullp.half()))
                del gp, img16
                barrier()
            except Exception as e:  # noqa: BLE001  (same inputs, same code on every rank: they fail alike)
                extras["cfg5"]["prepared_scene_error"] = repr(e)
        elif args.extras:
            # not the headline: the multi-view entry point (16 cameras per call, two views in flight)
            sc, cam, g = load(args.workload)
            cams = [cam] * 16
            ms.render_gaussians_batch(*g, cams, background_color=bg)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(4):
                ms.render_gaussians_batch(*g, cams, background_color=bg)
            torch.cuda.synchronize()
            extras["multi_view_batch16_views_per_s"] = round(64 / (time.perf_counter() - tb), 1)
        del sc5, g5
        torch.cuda.empty_cache()
        if world > 1:
            # not the headline either: the OTHER sharding axis -- whole views split over the ranks, one in-place
            # all-gather per call (render_gaussians_batch_sharded).  No per-frame fixed cost grows with the rank count
            # here; the exchange moves the same bytes per view as the band gather.
            try:
                from mojosplat_amd.distributed import render_gaussians_batch_sharded
                sc, cam, g = load(args.workload)
                cams = [cam] * (2 * world)
                for _ in range(3):
                    v = render_gaussians_batch_sharded(*g, cams, background_color=bg)
                same = bool(torch.equal(v[rank], ms.render_gaussians(*g, cam, background_color=bg, backend="hip")))
                barrier()
                tb = time.perf_counter()
                pend = None
                for _ in range(10):   # one call ahead: call k + 1 renders while call k's views are exchanged
                    nxt = render_gaussians_batch_sharded(*g, cams, background_color=bg, async_op=True)
                    if pend is not None:
                        pend.wait()
                    pend = nxt
                pend.wait()
                barrier()
                tv = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=dev)
                dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                extras["view_sharded"] = {"workload": f"{args.workload}, {2 * world} views per call split by view over {world} ranks",
                                          "views_per_s": round(10 * 2 * world / float(tv.item()), 1),
                                          "view_equals_single_gpu_frame": same}
                del sc, g, v
            except Exception as e:  # noqa: BLE001  (same inputs, same code on every rank: they fail alike)
                extras["view_sharded"] = {"error": repr(e)}

    rc = 0
    if rank == 0:
        per_m = 40 - (6 if fp16 else 0)
        # the lists the timed kernel is given: N = 1 the frame's own (read back); N > 1 rank 0's BAND (what the stage events
        # time), never the whole frame's
        m_kernel = m_lists if m_lists is not None else (band_stats["M"] if band_stats is not None else M)
        b_raster = per_m * m_kernel + 8 * T + 12 * H * W
        b_raster_gsplat = per_m * M + 8 * T + 12 * H * W
        if band_stats is not None:   # N > 1: the kernel timed is rank 0's band
            b_raster = b_raster_gsplat = per_m * band_stats["M"] + 8 * band_stats["T"] + 12 * band_stats["px"]
        b_frame_survey = 96 * N + (84 - (6 if fp16 else 0)) * M + 12 * T + 12 * H * W
        # A depth-cut frame (configs 4 / 5) never writes, sorts or rasterises the pairs behind its bins' cut-offs: SURVEY's
        # 84 bytes per pair price work that is legitimately not done (the frame is bit-identical to the full path).  The
        # frame's own figure therefore counts the pairs it KEPT -- gsplat's M scaled by the share of its grid's pairs that
        # are on the lists the rasteriser was given -- and the SURVEY model's figure is printed beside it, labelled.
        kept_share = 1.0
        if depth_cut and m_lists is not None and pairs_on_grid:
            kept_share = min(1.0, m_lists / pairs_on_grid)
        b_frame = 96 * N + int((84 - (6 if fp16 else 0)) * M * kept_share) + 12 * T + 12 * H * W
        roofline = None
        if "raster" in stage_us:
            ach = b_raster / (stage_us["raster"] * 1e-6) / 1e9
            # HBM bytes of the kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE, scripts/collect_profiles.sh): a stored
            # measurement, tied to the kernel sources it was taken on -- sources that have changed since void it
            traffic, traffic_note = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and world == 1:   # measured on the whole frame
                rec = json.load(open(tpath)).get(args.workload, {})
                have = raster_sha16()
                if rec.get("raster_sha16") == have:
                    traffic = rec.get("rasterize_fwd_bytes")
                    traffic_note = (f"PMC passes on rasterize.hip + its headers {have} (written by scripts/collect_profiles.sh at commit "
                                    f"{rec.get('commit', '?')}), profiles/{rec.get('source', 'traffic.json')}")
                else:
                    traffic_note = (f"null: profiles/traffic.json was measured on rasterize.hip + headers {rec.get('raster_sha16')}, "
                                    f"this library is built from {have}")
            roofline = {"bound": "hbm", "kernel": "k_rasterize_fwd" + ("" if world == 1 else " (rank 0's band)"),
                        "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes": b_raster,
                        "intersections_in_kernel_lists": m_kernel,
                        "algorithmic_bytes_gsplat_M": b_raster_gsplat,
                        "frac_gsplat_M": round(b_raster_gsplat / (stage_us["raster"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        "avg_kernel_us": round(stage_us["raster"], 1), "instrumented_launches": len(stage_events),
                        "alpha_evals": 256 * m_kernel,
                        "frame": {"algorithmic_bytes": b_frame, "on": "ms_per_step_mean (wall time of the K steps / K)",
                                  "pairs": "kept pairs only: gsplat's M x %.4f (depth-cut frame)" % kept_share if kept_share < 1.0
                                           else "gsplat's M (every pair is written, sorted and offered to the rasteriser)",
                                  "achieved": round(b_frame / (ms_mean * 1e-3) / 1e9, 1),
                                  "frac": round(b_frame / (ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                        "frame_survey_model": {"algorithmic_bytes": b_frame_survey,
                                               "what": "SURVEY 8(d)'s 96 N + 84 M + 12 T + 12 HW on gsplat's M whatever the frame "
                                                       "skipped: a MODEL of a full frame's traffic, not bytes this frame moved -- "
                                                       "not a fraction of anything when pairs are dropped",
                                               "model_GBps": round(b_frame_survey / (ms_mean * 1e-3) / 1e9, 1)},
                        "stage_us": {k: round(v, 1) for k, v in stage_us.items()}}
        cpu = None
        verification = {"bit_identical_to_stagewise": verified, "max_abs_vs_stagewise": max_abs}
        if world == 1 and not (args.no_cpu_baseline and args.no_verify):
            sc, cam, g = load(args.workload)
            cpu, ocheck = cpu_baseline(sc, cam, W, H, BACKGROUND_V1, args.workload,
                                             img if not args.no_verify else None, not args.no_cpu_baseline)
            if ocheck is not None:
                verification["oracle"] = ocheck
                verified = verified and ocheck["unexplained_px"] == 0 and ocheck["within_flip_cap"]
        out = {
            "metric": "frames/sec at 1M Gaussians 1920x1080 fwd; achieved HBM GB/s vs peak",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_step_mean": round(ms_mean, 4),
            "timing": "value = steps / wall time of the timed region; ms_per_step = median per-step period"
                      if ms_per_step != ms_mean else "value = steps / wall time of the timed region = 1000 / ms_per_step",
            "period_us": period_us, "spin_up_frames": 64 * spin_blocks,
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)" if rehearse else ""),
            "config": {"workload": f"{args.workload}: randscene-v1 N={N} {W}x{H} ell={ell} seed=42 forward",
                       # (whether the timed frames dropped the pairs behind their bins' depth cut-offs: csrc/binning.hip,
                       # k_project_hist<.., CUT>; from 6 M pairs per frame by default -- config 3 runs uncut)
                       "depth_cut": depth_cut,
                       "gaussians": N, "intersections": M, "tiles": T, "tile_size": 16, "binning": binning,
                       "colour_dtype": "f16" if fp16 else "f32",
                       "parallelism": "single GPU" if world == 1 else f"{world} tile-row bands + RCCL all-gather" + (", gather of frame k overlapped with render of frame k+1" if mode["async"] else " (blocking)")},
            "verified": verified, "max_abs_vs_stagewise": max_abs, "verification": verification,
            "rccl": None if world == 1 else {"world": dist.get_world_size(), "backend": dist.get_backend(),
                                             "devices": "all ranks on cuda:0 (rehearsal)" if rehearse else "one per rank",
                                             "algo": {"framebuffer": "grouped isend / irecv into the image's rows (MOJOSPLAT_GATHER=direct)"
                                                      if os.environ.get("MOJOSPLAT_GATHER") == "direct"
                                                      else "in-place all_gather_into_tensor (padded slots + one compaction copy once bands are ragged)",
                                                      "status": "all_gather of 32 bytes per rank per frame (on-grid count, Gaussians reaching a pre-culled band or -1, pairs, frame stamp)",
                                                      "band_balance": os.environ.get("MOJOSPLAT_BALANCE", "1") != "0",
                                                      "bands_at_end": band_bounds_end}},
            "roofline": roofline, "cpu_baseline": cpu,
            "extras": (dict(extras or {}, **legs) or None),
        }
        print(json.dumps(out), flush=True)
        if not verified:
            print("[bench] VERIFICATION FAILED: the timed frame differs from the per-stage path / the oracle",
                  file=sys.stderr, flush=True)
            rc = 3
    if world > 1:
        code = torch.tensor([rc], device=dev)
        dist.all_reduce(code, op=dist.ReduceOp.MAX)
        rc = int(code.item())
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def morton_leg(ms, _fused, g, cam, bg, ref_img, given_ms, frames=200):
    """The same scene with its Gaussians sorted along a Morton curve of their means (a caller-side permutation of the five
    arrays: scene_order.morton_permutation) through the same blocking call.  The order does not change a pixel; since
    round 5 it must not cost a frame anything either (the count / scatter kernels deal their positions over the
    workgroups and queue the large boxes per workgroup: csrc/binning.hip, Deal / BigQ) -- round 4 measured +5 % here and
    +16 % at config 5."""
    import torch
    from mojosplat_amd.scene_order import morton_permutation
    perm = morton_permutation(g[0])
    gm = tuple(t[perm].contiguous() for t in g)
    _fused.release_scratch()
    for _ in range(64):
        img = ms.render_gaussians(*gm, cam, background_color=bg, backend="hip")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        img = ms.render_gaussians(*gm, cam, background_color=bg, backend="hip")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    # (back to the given order's scratch and hints for the legs that follow: a fresh lane, warmed up again)
    _fused.release_scratch()
    for _ in range(8):
        ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
    torch.cuda.synchronize()
    # (a permutation reorders equal-depth ties and the float sums' order nowhere: lists are sorted by (depth bits, index) --
    # the INDEX differs, so two Gaussians at bit-equal depth may swap; compared to rounding, not bit for bit)
    d = (img - ref_img).abs().max(-1).values
    return {"frames": frames, "ms_per_frame": round(dt, 4), "frames_per_s": round(1e3 / dt, 1),
            "vs_given_order": round(dt / given_ms, 4),
            "px_differing_from_the_given_order_frame": int((d > 0).sum()), "max_abs": float(d.max()),
            "why_any": "lists are sorted by (depth bits, index): Gaussians at bit-equal depth swap when the indices are permuted"}


def two_in_flight_leg(ms, g, cam, bg, ref_img, steps):
    """The same frames through render_gaussians(async_op=True), used one frame ahead: two frames in flight on two lane
    streams, each with its own scratch.  The headline `value` stays the blocking call's (one frame at a time on the
    caller's stream: its kernels' durations are what `roofline` prices); this is what a caller that renders frame after
    frame gets by asking one call earlier."""
    import torch
    steps = max(int(steps), 64)

    def run(n):
        cur = ms.render_gaussians(*g, cam, background_color=bg, backend="hip", async_op=True)
        for _ in range(n - 1):
            nxt = ms.render_gaussians(*g, cam, background_color=bg, backend="hip", async_op=True)
            cur.wait()
            cur = nxt
        return cur.wait()
    run(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames": steps, "frames_per_s": round(steps / dt, 1), "ms_per_frame": round(dt / steps * 1e3, 4),
            "last_frame_bit_identical_to_the_blocking_call": bool(torch.equal(img, ref_img)),
            "how": "nxt = render_gaussians(..., async_op=True); img = cur.wait(); cur = nxt"}


def orbit_leg(ms, _fused, render_mod, g, cam, bg, stagewise, static_ms, frames=256):
    """The same scene from a camera that moves every frame (the reference's callers render changing views:
    render_sample.py:60-71, examples/benchmark_proj.py:124-145): one orbit around the scene at the static camera's
    radius and height, `frames` poses, a new view matrix per frame.  Everything the sync-free path bets on from the
    previous frame's size record (buffer capacity, "no heavy tile", front depth, the binning rule's grid) is now a
    bet against a DIFFERENT frame: the counters say how often it was lost."""
    import math

    import torch

    from mojosplat_amd.utils import Camera, look_at
    dev = g[0].device
    c = cam._campos()
    radius, height = math.hypot(c[0], c[2]), c[1]
    cams = []
    for k in range(frames):
        th = 2.0 * math.pi * k / frames
        vm = look_at(torch.tensor([radius * math.sin(th), height, radius * math.cos(th)]), torch.zeros(3),
                     torch.tensor([0.0, 1.0, 0.0]))
        cams.append(Camera(R=vm[:3, :3].contiguous().to(dev), T=vm[:3, 3].contiguous().to(dev), H=cam.H, W=cam.W,
                           fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, near=cam.near, far=cam.far))
    for cm in cams:
        cm._viewmat_f32()      # the 64-byte device copy of each pose exists before the clock starts
    key = render_mod._bin_key(g[0], cam)
    host = _fused._state[(dev, 0)]["host_np"]
    for cm in cams[-8:]:       # approach the start of the orbit
        ms.render_gaussians(*g, cm, background_color=bg, backend="hip")
    torch.cuda.synchronize()
    sample = {0, frames // 3, 2 * frames // 3, frames - 1}
    kept, ms_pairs, modes, switches = {}, [], [], 0
    stats = _fused.FRAME_STATS = {}
    stamps = [time.perf_counter()]
    for k, cm in enumerate(cams):
        mode = render_mod._bin_mode.get(key, 16)
        img = ms.render_gaussians(*g, cm, background_color=bg, backend="hip")
        stamps.append(time.perf_counter())
        ms_pairs.append(int(host[0]))
        switches += bool(modes and modes[-1] != mode)
        modes.append(mode)
        if k in sample:
            kept[k] = img      # (a reference to the frame as rendered here, no copy)
    torch.cuda.synchronize()
    dt = time.perf_counter() - stamps[0]
    _fused.FRAME_STATS = None
    periods = sorted(b - a for a, b in zip(stamps[1:-1], stamps[2:]))
    same = True
    for k, img in sorted(kept.items()):
        ref, _, _ = stagewise(g, cams[k])
        same = same and bool(torch.equal(img, ref))
    fps = frames / dt
    out = {"what": f"{frames} frames, camera on one orbit of the scene (radius {radius:.2f}, height {height:.2f}), a new "
                   "view matrix every frame; render_gaussians(backend='hip') per frame, blocking API",
           "frames": frames, "frames_per_s": round(fps, 1), "ms_per_frame_mean": round(dt / frames * 1e3, 4),
           "ms_per_frame_median": round(periods[len(periods) // 2] * 1e3, 4),
           "vs_static_camera": round((dt / frames * 1e3) / static_ms, 3),
           "pairs_min_max": [min(ms_pairs), max(ms_pairs)], "bin_px_used": sorted(set(modes)),
           "bin_rule_switches": int(switches),
           "misses": {k: int(stats.get(k, 0)) for k in ("frames", "speculated", "redone_exact", "overflow", "light_bet_lost",
                                                       "other_miss", "buffer_grown", "redo_tiles", "front_level_up",
                                                       "full_sort_on", "depth_cut", "cut_redo_tiles")},
           "sampled_frames": sorted(kept), "sampled_frames_bit_identical_to_stagewise": same}
    return out


def fwd_bwd_leg(_fused, render_mod, g, cam, bg, N, W, H, T, dev, steps=30, warm=6, m_gsplat=None):
    """BASELINE config 3 as named: forward + backward, grads for means / scales / quats / opacities / colours
    (dL/dimage = rand(H, W, 3), seed 43: SURVEY 8(d)).  ms per step over `steps` un-instrumented steps; stage times
    from HIP events on the launch stream in a separate short pass (an event between two kernels costs a bubble)."""
    import torch

    from mojosplat_amd import autograd as ag
    leaves = [t.float().clone().requires_grad_(True) for t in g]
    v_img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(43)).to(dev)
    host = _fused._state[(dev, 0)]["host_np"]

    def step():
        for l in leaves:
            l.grad = None
        img = ag.render_gaussians_trainable(*leaves, cam, background_color=bg)
        img.backward(v_img)           # dL/dimage = v_img; no loss kernels inside the step

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    stamps = [time.perf_counter()]
    for _ in range(steps):
        step()
        torch.cuda.synchronize()      # a training step ends with its gradients
        stamps.append(time.perf_counter())
    dt = stamps[-1] - stamps[0]
    periods = sorted(b - a for a, b in zip(stamps[:-1], stamps[1:]))
    # ... and streamed: the host runs ahead of the GPU (as it does when the optimiser step is enqueued behind the
    # backward and nothing is read back), one synchronisation at the end
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_streamed = time.perf_counter() - t0
    m_lists = int(host[0])            # pairs on the differentiable frame's lists (the binning rule's grid, tight binning)
    finite = all(bool(torch.isfinite(l.grad).all()) for l in leaves)
    # stage times
    n_ev = 10
    fev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_ev)]
    bev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_ev)]
    for evs in fev + bev:
        for e in evs:
            e.record()
    torch.cuda.synchronize()
    fi, bi = iter(fev), iter(bev)
    render_mod._STAGE_HOOK = lambda: next(fi)
    ag._BWD_HOOK = lambda: next(bi)
    for _ in range(n_ev):
        step()
    torch.cuda.synchronize()
    render_mod._STAGE_HOOK = None
    ag._BWD_HOOK = None
    mean = lambda v: sum(v) / len(v)
    us = {"fwd_project_count": mean([e[0].elapsed_time(e[1]) for e in fev]) * 1e3,
          "fwd_bin": mean([e[1].elapsed_time(e[2]) for e in fev]) * 1e3,
          "fwd_raster": mean([e[2].elapsed_time(e[3]) for e in fev]) * 1e3,
          "bwd_raster_call": mean([e[0].elapsed_time(e[1]) for e in bev]) * 1e3,
          "bwd_project": mean([e[1].elapsed_time(e[2]) for e in bev]) * 1e3}
    b_rbwd = 40 * m_lists + 24 * H * W + 36 * N
    b_rbwd_gsplat = 40 * m_gsplat + 24 * H * W + 36 * N if m_gsplat else None
    b_pbwd = 108 * N
    return {"what": "render_gaussians_trainable forward + img.backward(dL/dimage = rand(H, W, 3) seed 43), grads for means3d / scales / quats / "
                    "opacities / colours; every step synchronised (a training step ends with its gradients)",
            "steps": steps, "ms_per_step_mean": round(dt / steps * 1e3, 4),
            "ms_per_step_median": round(periods[len(periods) // 2] * 1e3, 4),
            "ms_per_step_streamed": round(dt_streamed / steps * 1e3, 4), "grads_finite": finite,
            "pairs_on_lists": m_lists,
            "stage_us": {k: round(v, 1) for k, v in us.items()},
            "stage_us_note": "HIP events on the launch stream, separate pass of 10 steps; the forward is the inference frame "
                             "(lazily sorted fronts on the binning rule's grid) keeping its alphas; bwd_raster_call = the rows' "
                             "memset + k_rasterize_bwd_quads + its (normally empty) redo launch (the first stage of ms_render_bwd)",
            "roofline": {"bwd_raster": {"bound": "hbm", "algorithmic_bytes": b_rbwd, "formula": "40 M + 24 HW + 36 N, M = the pairs on the lists the kernel walks",
                                        "achieved": round(b_rbwd / (us["bwd_raster_call"] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": round(b_rbwd / (us["bwd_raster_call"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                        "algorithmic_bytes_gsplat_M": b_rbwd_gsplat,
                                        "frac_gsplat_M": None if not b_rbwd_gsplat else
                                        round(b_rbwd_gsplat / (us["bwd_raster_call"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                         "bwd_project": {"bound": "hbm", "algorithmic_bytes": b_pbwd, "formula": "108 N",
                                         "achieved": round(b_pbwd / (us["bwd_project"] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(b_pbwd / (us["bwd_project"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}}


def raster_sha16():
    """sha256[:16] over the forward rasteriser's translation unit: rasterize.hip and the headers it includes.  What
    profiles/traffic.json's rasteriser figure is tied to (an edit to binning.hip cannot void it; nobody re-stamps it by
    hand: scripts/collect_profiles.sh writes it with the measurement)."""
    import hashlib
    d = os.path.join(ROOT, "mojosplat_amd", "csrc")
    h = hashlib.sha256()
    for f in ("rasterize.hip", "ms_common.hpp", os.path.join("..", "..", "include", "mojosplat_hip.h")):
        h.update(os.path.basename(f).encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def oracle_check(gpu_img, ref, margin, *, atol=1e-4, eps=2e-5, flip_cap=1e-2):
    """The suite's end-to-end bar (tests/helpers.py check_image_strict with eps = 2e-5, as tests/test_hip_configs.py
    and __graft_entry__.smoke use it): a pixel may differ from the oracle's by more than atol only where a branch of
    the oracle's walk sits within eps of its threshold; even those are capped at flip_cap."""
    import numpy as np
    diff = np.abs(gpu_img.astype(np.float64) - ref.astype(np.float64)).max(-1)
    bad = diff > atol
    sens = margin < eps
    return {"max_abs": float(diff.max()), "px_beyond_1e-4": int(bad.sum()),
            "explained_by_branch_margin": int((bad & sens).sum()), "unexplained_px": int((bad & ~sens).sum()),
            "max_abs_where_no_branch_is_close": float(diff[~sens].max()) if (~sens).any() else 0.0,
            "margin_eps": eps, "flip_cap": flip_cap, "within_flip_cap": bool(diff.max() <= flip_cap),
            "what": "GPU frame (its own projection) vs oracle.render_fwd end to end; tests/helpers.py "
                    "check_image_strict's rule with the suite's end-to-end eps"}


def cpu_baseline(sc, cam, W, H, bg, workload, gpu_img, timed):
    """The scalar C oracle (kind 'port', 1 core) on a bounded sample of the same workload: whole
    frames of the scene (its first <= 1M Gaussians), repeated until ~10 s of CPU work, reported
    as frames/s of that sample (stated in `sample`); beside it the package's backend='torch'
    projection (the reference's CPU path, mojosplat/projection.py:285-346 restated) on CPU tensors.
    gpu_img: the timed path's frame, checked here against the oracle's (None: skipped).
    -> (cpu_baseline record or None, oracle check record or None)"""
    import numpy as np
    import torch

    import oracle
    cpu = {k: v.float().cpu().numpy() for k, v in sc.items()}
    vm = cam.view_matrix.cpu().numpy()
    n_all = len(cpu["means3d"])
    n = min(n_all, 1_000_000)
    full = tuple(cpu[k] for k in ("means3d", "scales", "quats", "opacities", "features"))
    bgn = np.array(bg, np.float32)

    check = None
    if gpu_img is not None:
        # all host threads here: this is the checker, not the baseline
        ref, aux = oracle.render_fwd(*full, vm, cam.fx, cam.fy, cam.cx, cam.cy, W, H, background=bgn, margin=True)
        check = oracle_check(gpu_img.float().cpu().numpy(), ref, aux["margin"])
    rec = None
    if timed:
        args = tuple(a[:n] for a in full)
        frames, t0 = 0, time.perf_counter()
        while True:
            _, aux = oracle.render_fwd(*args, vm, cam.fx, cam.fy, cam.cx, cam.cy, W, H, background=bgn, threads=1)
            frames += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or frames >= 8:
                break
        # the reference's CPU path: backend="torch" projection on CPU tensors at this workload's N
        from mojosplat_amd.projection import project_gaussians
        tcpu = [torch.from_numpy(a) for a in full[:4]]
        ccam = type(cam)(R=cam.R.cpu(), T=cam.T.cpu(), H=cam.H, W=cam.W, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         near=cam.near, far=cam.far)
        project_gaussians(*[t[:1000] for t in tcpu], ccam, backend="torch")   # warm-up
        reps, tt0 = 0, time.perf_counter()
        while True:
            project_gaussians(*tcpu, ccam, backend="torch")
            reps += 1
            tdt = time.perf_counter() - tt0
            if tdt >= 5.0 or reps >= 5:
                break
        rec = {"value": round(frames / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{frames} frame(s) of {workload} restricted to its first {n} Gaussians "
                         f"(M={aux['M']}), oracle/gsplat_oracle.c, {dt:.1f} s of CPU work",
               "host_cpus": os.cpu_count(),
               "torch_backend": {"stage": "projection only (backend='torch' on CPU tensors; the reference's CPU "
                                          "path has no rasteriser and a Python-loop binning)",
                                 "gaussians": n_all, "seconds_per_call": round(tdt / reps, 4), "calls": reps,
                                 "gaussians_per_s": round(n_all * reps / tdt, 1),
                                 "torch_num_threads": torch.get_num_threads(), "os_cpu_count": os.cpu_count()}}
    return rec, check


if __name__ == "__main__":
    main()
