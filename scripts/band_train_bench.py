"""One rank's share of the sharded TRAINING step (render_gaussians_trainable_sharded, rehearse=(rank, world)) on one GPU, no
exchange: the band's differentiable forward (every Gaussian projected: a differentiable band is not pre-culled), the band's
backward rasteriser into the per-Gaussian rows, the backward projection on them.  Per world size: each rank's streamed step time
(no synchronisation between forward and backward, one at the end), the slowest rank, and beside it what the two exchanges of a
live step move (the forward's framebuffer all-gather; the backward's all-reduce of 64 bytes per Gaussian).

    python scripts/band_train_bench.py --workload cfg3 [--worlds 1,2,4,8] [--steps 40]
"""
import argparse, json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from mojosplat_amd.distributed import render_gaussians_trainable_sharded  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402

XGMI_GBS = 7 * 153.0   # a GPU's seven links, all busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    N, W, H, ell, fp16 = WORKLOADS[args.workload]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    leaves = [sc[k].clone().requires_grad_(True) for k in ("means3d", "scales", "quats", "opacities", "features")]
    gen = torch.Generator(device=dev).manual_seed(43)
    v_img = torch.rand((H, W, 3), device=dev, generator=gen)

    def step(rank, world):
        for t in leaves:
            t.grad = None
        img = render_gaussians_trainable_sharded(*leaves, cam, background_color=bg, rehearse=(rank, world))
        img.backward(v_img)

    if os.environ.get("PROFILE_HOST"):   # where the HOST's time of a band's step goes (the GPU work of 1 / 8 of a frame is short)
        import cProfile, pstats
        for _ in range(6):
            step(3, 8)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            step(3, 8)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
        return
    for world in [int(v) for v in args.worlds.split(",")]:
        ranks = []
        for r in range(world):
            for _ in range(6):
                step(r, world)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(r, world)
            torch.cuda.synchronize()
            streamed = (time.perf_counter() - t0) / args.steps
            sync = []
            for _ in range(min(args.steps, 20)):
                t0 = time.perf_counter()
                step(r, world)
                torch.cuda.synchronize()
                sync.append(time.perf_counter() - t0)
            ranks.append(dict(rank=r, ms_per_step_streamed=round(streamed * 1e3, 4), ms_per_step_synchronised=round(statistics.median(sync) * 1e3, 4)))
        worst = max(x["ms_per_step_streamed"] for x in ranks)
        gather_mb = H * W * 3 * 4 * (world - 1) / world / 1e6
        reduce_mb = 2 * 64 * N * (world - 1) / world / 1e6   # reduce-scatter + all-gather of the rows, per GPU, each way
        print(json.dumps(dict(workload=args.workload, world=world, steps=args.steps, slowest_rank_ms_streamed=worst,
                              ranks=ranks,
                              exchange=dict(forward_all_gather_MB_per_gpu=round(gather_mb, 1), backward_all_reduce_MB_per_gpu=round(reduce_mb, 1),
                                            floor_us_at_7x153GBs=round((gather_mb + reduce_mb) / XGMI_GBS * 1e3, 1)),
                              note="rehearsal on ONE GPU: compute only, no exchange; nothing here was measured on more than one GPU")),
              flush=True)


if __name__ == "__main__":
    main()
