"""Where a small frame's wall time goes on the host: render_gaussians -> render_fwd_hip -> the bare
ms_render_fwd ctypes call (arguments marshalled once), on a scene so small the GPU work is noise.

    python scripts/host_overhead.py
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd import _fused  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def timeit(f, n=2000):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = torch.device("cuda", 0)
    sc, cam = randscene_v1(2000, 128, 128, ell=-3.0, seed=1, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    print("render_gaussians        us/frame", round(timeit(lambda: ms.render_gaussians(*g, cam, background_color=bg, backend="hip")), 1))
    print("render_fwd_hip          us/frame", round(timeit(lambda: _fused.render_fwd_hip(*g, cam, bg, 16)), 1))

    from mojosplat_amd.distributed import render_gaussians_sharded

    def pipelined(n, **kw):
        cur = None
        for _ in range(n):
            nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True, **kw)
            if cur is not None:
                cur.wait()
            cur = nxt
        cur.wait()
    pipelined(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipelined(2000)
    torch.cuda.synchronize()
    print("sharded async pipelined us/frame", round((time.perf_counter() - t0) / 2000 * 1e6, 1))
    fr = _fused.render_begin_hip(*g, cam, bg, 16, lane=1)
    fr.finish()

    def begin_finish():
        _fused.render_begin_hip(*g, cam, bg, 16, lane=1).finish()
    print("begin+finish (1 lane)   us/frame", round(timeit(begin_finish), 1))
    t0 = time.perf_counter()
    for _ in range(2000):
        fr.run(_fused.BEGIN)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("bare BEGIN calls: host us/call", round((t1 - t0) / 2000 * 1e6, 1), " incl. drain",
          round((time.perf_counter() - t0) / 2000 * 1e6, 1))
    fr.finish()

    pr = cProfile.Profile()
    pr.enable()
    pipelined(2000)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
