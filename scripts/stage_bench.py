"""Development timing of each stage (torch.cuda events on the current stream).

python scripts/stage_bench.py [N] [W] [H] [ell] [iters]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    ell = float(sys.argv[4]) if len(sys.argv) > 4 else -4.0
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    dev = torch.device("cuda:0")
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    args = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"])
    m2, con, dep, rad = ms.project_gaussians(*args, cam, backend="hip")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip")
    M = ids.numel()
    cnt = (ranges[..., 1] - ranges[..., 0]).flatten()
    print(f"N={N} {W}x{H} ell={ell}: M={M} M/N={M / N:.2f} tiles={cnt.numel()} "
          f"max/tile={cnt.max().item()} mean/tile={cnt.float().mean().item():.1f}")
    t_proj = timeit(lambda: ms.project_gaussians(*args, cam, backend="hip"), iters)
    t_bin = timeit(lambda: ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip"), iters)
    t_ras = timeit(lambda: ms.rasterize_gaussians(m2, con, sc["features"], sc["opacities"], bg, ranges, ids,
                                                  cam, backend="hip"), iters)
    t_all = timeit(lambda: ms.render_gaussians(*args, sc["features"], cam, background_color=bg,
                                               backend="hip"), iters)
    t0 = time.perf_counter()
    for _ in range(iters):
        ms.render_gaussians(*args, sc["features"], cam, background_color=bg, backend="hip")
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    for name, (med, mn) in (("project", t_proj), ("bin", t_bin), ("raster", t_ras), ("render", t_all)):
        print(f"  {name:8s} median {med * 1e3:9.1f} us   min {mn * 1e3:9.1f} us")
    print(f"  render wall/frame {wall * 1e3:.1f} us -> {1e3 / wall:.1f} fps")


if __name__ == "__main__":
    main()
