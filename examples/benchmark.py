"""Full-pipeline benchmark (the script the reference's README.md:129-130 names but does not ship):
per-stage and end-to-end timings of the HIP backend for a list of scene sizes, with N, M and T printed
beside every figure.

    python examples/benchmark.py [--sizes 100000 1000000] [--width 1920 --height 1080] [--ell -4.0] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def timeit(fn, iters=20, warm=3):
    """-> (median, min) milliseconds between a pair of stream events around fn()."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def run(N, W, H, ell, iters, dev):
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    args = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"])
    m2, con, dep, rad = ms.project_gaussians(*args, cam, backend="hip")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip")
    cnt = (ranges[..., 1] - ranges[..., 0]).flatten()
    rec = {"N": N, "W": W, "H": H, "ell": ell, "M": int(ids.numel()), "T": int(cnt.numel()),
           "max_per_tile": int(cnt.max()), "mean_per_tile": round(float(cnt.float().mean()), 1)}
    stages = {
        "project": lambda: ms.project_gaussians(*args, cam, backend="hip"),
        "bin": lambda: ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip"),
        "raster": lambda: ms.rasterize_gaussians(m2, con, sc["features"], sc["opacities"], bg, ranges, ids, cam,
                                                 backend="hip"),
        "render": lambda: ms.render_gaussians(*args, sc["features"], cam, background_color=bg, backend="hip"),
    }
    for name, fn in stages.items():
        med, mn = timeit(fn, iters)
        rec[name + "_us"] = {"median": round(med * 1e3, 1), "min": round(mn * 1e3, 1)}
    t0 = time.perf_counter()
    for _ in range(iters):
        stages["render"]()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    rec["render_wall_us"] = round(wall * 1e6, 1)
    rec["fps"] = round(1.0 / wall, 1)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[100_000, 1_000_000])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ell", type=float, default=-4.0)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("backend='hip' needs a ROCm GPU")
    dev = torch.device("cuda:0")
    for n in a.sizes:
        print(json.dumps(run(n, a.width, a.height, a.ell, a.iters, dev)), flush=True)


if __name__ == "__main__":
    main()
