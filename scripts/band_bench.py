"""Strong-scaling rehearsal on ONE GPU: every rank's band of the same frame through the sharded entry point
(rehearse = act as that rank, no exchange), for world = 1, 2, 4, 8 -- with the band boundaries the ranks' own pair
counts lead to (distributed.rebalance, iterated here as the live group iterates it) beside the equal bands.

    python scripts/band_bench.py [--workload cfg3] [--frames 300]

Per rank: the MEDIAN time of `frames` blocking frames (GPU work + the host's enqueue, each frame synchronised), the
per-frame period of the pipelined entry point (two frames in flight: half the median period of two consecutive frames --
the periods of two alternating lanes can alternate, so the median of single periods, what rounds 2-4 first printed, can
be off either way; the mean is printed too, but one slow frame in 150 moves it) and the host time of an enqueue.  The
slowest rank's median bounds what N GPUs can reach before the exchange is added.  One JSON line per (world, plan).
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from mojosplat_amd import _fused  # noqa: E402
from mojosplat_amd.distributed import band_plan, rebalance, render_gaussians_sharded  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--worlds", default="1,2,4,8")
    # given: the scene as randscene_v1 draws it (random order); morton: sorted along a Morton curve of the means by the
    # CALLER (no bounds); prepared: scene_order.prepare_scene -- the same order plus the block bounds the band pre-cull uses
    ap.add_argument("--order", default="given", choices=["given", "morton", "prepared"])
    # f16: the band is rounded to float16 for the exchange (render_gaussians_sharded's exchange_dtype): the rank's frame then
    # includes that one cast of its band
    ap.add_argument("--exchange", default="f32", choices=["f32", "f16", "bf16"])
    args = ap.parse_args()
    xdt = {"f32": None, "f16": torch.float16, "bf16": torch.bfloat16}[args.exchange]
    dev = torch.device("cuda", 0)
    N, W, H, ell, fp16 = WORKLOADS[args.workload]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    if fp16:
        sc["features"] = sc["features"].half()
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    if args.order == "morton":
        from mojosplat_amd.scene_order import morton_permutation
        perm = morton_permutation(g[0])
        g = tuple(t[perm].contiguous() for t in g)
    elif args.order == "prepared":
        from mojosplat_amd.scene_order import prepare_scene
        g = prepare_scene(*g).arrays
    th = -(-H // 16)

    def render(r, world, bounds, **kw):
        return render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(r, world), bounds=bounds, exchange_dtype=xdt, **kw)

    def pairs_of(r, world, bounds):
        """the weight a live rank reports for its band: the Gaussians that reach it (pre-culled band) or its pairs"""
        for _ in range(3):      # (a band's first frames settle its bin size)
            render(r, world, bounds)
        torch.cuda.synchronize()
        h = _fused._state[(dev, 0)]["host_np"]
        culled = bool(int(h[7]) & 2048)   # (the library says whether it pre-culled the band)
        return int(h[6]) if culled else int(h[0])

    def measure(world, bounds, label):
        ranks = []
        for r in range(world):
            for _ in range(12):
                render(r, world, bounds)
            torch.cuda.synchronize()
            blocking, host = [], []
            for _ in range(args.frames):
                t0 = time.perf_counter()
                render(r, world, bounds)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                blocking.append(time.perf_counter() - t0)
                host.append(t1 - t0)
            # pipelined: frame k + 1 begun before frame k is finished (no exchange in rehearsal mode)
            cur = None
            for _ in range(24):
                nxt = render(r, world, bounds, async_op=True)
                if cur is not None:
                    cur.wait()
                cur = nxt
            cur.wait()
            torch.cuda.synchronize()
            stamps = [time.perf_counter()]
            cur = None
            for _ in range(args.frames):
                nxt = render(r, world, bounds, async_op=True)
                if cur is not None:
                    cur.wait()
                cur = nxt
                stamps.append(time.perf_counter())
            cur.wait()
            torch.cuda.synchronize()
            periods = [b - a for a, b in zip(stamps[1:-1], stamps[2:])]
            # the HOST's own cost of a band frame (round 5): the asynchronous entry point's begin half (never waits), and its
            # finishing half called once the device has drained (so that no wait for the GPU is in it) -- the blocking
            # call's `host_us_median` above includes the wait for the band's size record, i.e. most of the band's GPU time
            hb, hf = [], []
            for _ in range(min(args.frames, 100)):
                t0 = time.perf_counter()
                pnd = render(r, world, bounds, async_op=True)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                pnd.wait()
                t3 = time.perf_counter()
                hb.append(t1 - t0)
                hf.append(t3 - t2)
            torch.cuda.synchronize()
            ranks.append(dict(band=[bounds[r], bounds[r + 1]], pairs=int(_fused._state[(dev, 1)]["host_np"][0]),
                              blocking_us_median=round(statistics.median(blocking) * 1e6, 1),
                              blocking_us_p90=round(sorted(blocking)[int(0.9 * len(blocking))] * 1e6, 1),
                              pipelined_us_median=round(statistics.median(periods) * 1e6, 1),
                              # (two lanes alternate: the periods can alternate short / long, and their median then is
                              # neither -- the mean over the run is what a rank sustains)
                              pipelined_us_mean=round((stamps[-1] - stamps[1]) / (len(stamps) - 2) * 1e6, 1),
                              # (robust to both: half the median period of TWO consecutive frames)
                              pipelined_us=round(statistics.median(b - a for a, b in zip(stamps[1:-2], stamps[3:])) * 0.5e6, 1),
                              host_us_median=round(statistics.median(host) * 1e6, 1),
                              host_begin_us_median=round(statistics.median(hb) * 1e6, 1),
                              host_finish_us_median=round(statistics.median(hf) * 1e6, 1),
                              host_own_us_median=round((statistics.median(hb) + statistics.median(hf)) * 1e6, 1)))
        worst_b = max(x["blocking_us_median"] for x in ranks)
        worst_p = max(x["pipelined_us"] for x in ranks)
        best_b = min(x["blocking_us_median"] for x in ranks)
        print(json.dumps(dict(workload=args.workload, order=args.order, exchange=args.exchange, world=world, plan=label, bounds=bounds, frames=args.frames,
                              slowest_blocking_us_median=worst_b, rank_spread=round(worst_b / best_b, 3),
                              slowest_pipelined_us=worst_p, fps_bound_pipelined=round(1e6 / worst_p, 1),
                              host_own_us_median_max=max(x["host_own_us_median"] for x in ranks),
                              notes="host_us_median is the BLOCKING call's duration on the host: it contains the wait for the band's size record, "
                                    "i.e. the band's GPU time up to its scatter launch (and, with the clean-up deferred, a pipelined frame's "
                                    "finish waits for the band's end).  What the HOST itself spends on a band frame is host_own_us_median = "
                                    "host_begin_us_median (the asynchronous entry point's begin half: never waits) + host_finish_us_median (its "
                                    "finishing half called with the device drained)", ranks=ranks)),
              flush=True)

    for world in [int(v) for v in args.worlds.split(",")]:
        rows, bands = band_plan(th, world)
        equal = [b[0] for b in bands] + [th]
        measure(world, equal, "equal bands")
        if world == 1:
            continue
        # the plan a live group converges to: re-plan from the bands' pair counts until the spread is under 8 %
        b = list(equal)
        for it in range(8):
            nb, spread = rebalance(b, [pairs_of(r, world, b) for r in range(world)])
            if spread <= 1.08 or nb == b:
                break
            b = nb
        if b != equal:
            measure(world, b, f"balanced on the bands' reported weights ({it + 1} re-plans)")


if __name__ == "__main__":
    main()
