"""Throughput of multi-view batches (two views in flight) vs one render_gaussians call per view."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc, cam = randscene_v1(N, 1920, 1080, ell=-4.0, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
cams = [cam] * 16
for _ in range(3):
    ms.render_gaussians_batch(*g, cams, background_color=bg)
    for c in cams:
        ms.render_gaussians(*g, c, background_color=bg)
torch.cuda.synchronize()
for name, fn in (("batch(16)", lambda: ms.render_gaussians_batch(*g, cams, background_color=bg)),
                 ("16 single calls", lambda: [ms.render_gaussians(*g, c, background_color=bg) for c in cams])):
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 160
    print(f"{name:18s} {dt * 1e6:8.1f} us/view  {1 / dt:8.1f} views/s")
