"""Strong-scaling rehearsal on ONE GPU: time each rank's tile-row band of the same frame (no
collective) for world = 1, 2, 4, 8, and the host-side enqueue cost of a frame.  The slowest band
bounds what N GPUs can reach before the framebuffer all-gather is added.

    python scripts/band_bench.py [--workload cfg3] [--steps 100]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from mojosplat_amd import _fused  # noqa: E402
from mojosplat_amd.distributed import band_plan, render_gaussians_sharded  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    N, W, H, ell, fp16 = WORKLOADS[args.workload]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    if fp16:
        sc["features"] = sc["features"].half()
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    th = -(-H // 16)
    for world in (1, 2, 4, 8):
        rows, bands = band_plan(th, world)
        frame = torch.empty((max(world * rows * 16, H), W, 3), device=dev)
        per_rank = []
        for r, band in enumerate(bands):
            # the blocking sharded entry point acting as rank r (no exchange): pre-cull, the band's own bin size
            for _ in range(6):
                render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(r, world))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(r, world))
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            from mojosplat_amd.render import _bin_mode
            per_rank.append(dict(band=band, bin_px=[v for k, v in _bin_mode.items() if k[0] == "band" and k[5] == tuple(band)][-1:] or [16],
                                 us=round(dt / args.steps * 1e6, 1), host_us=round(t_host / args.steps * 1e6, 1)))
        # the same through the asynchronous sharded entry point (frame k+1 begun before frame k is
        # finished; no exchange in rehearsal mode): what a rank's host + GPU can sustain
        for r, rec in enumerate(per_rank):
            def run(n):
                cur = None
                for _ in range(n):
                    nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True, rehearse=(r, world))
                    if cur is not None:
                        cur.wait()
                    cur = nxt
                cur.wait()
            run(24)   # (a rank's first pipelined frames grow its lanes' buffers and settle its bin size)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            rec["pipelined_us"] = round((time.perf_counter() - t0) / args.steps * 1e6, 1)
        worst = max(r["us"] for r in per_rank)
        worst_p = max(r["pipelined_us"] for r in per_rank)
        print(json.dumps(dict(workload=args.workload, world=world, slowest_band_us=worst,
                              fps_bound=round(1e6 / worst, 1), slowest_pipelined_us=worst_p,
                              fps_bound_pipelined=round(1e6 / worst_p, 1), ranks=per_rank)), flush=True)


if __name__ == "__main__":
    main()
