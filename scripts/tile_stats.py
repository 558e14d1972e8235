"""Distribution of per-tile list lengths (gsplat-exact lists) for a workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from bench import WORKLOADS
from mojosplat_amd.scenes import randscene_v1
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["cfg3"]:
    N, W, H, ell, fp16 = WORKLOADS[wl]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
    ts = int(os.environ.get("TS", "16"))
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, ts, backend="hip")
    n = (ranges[..., 1] - ranges[..., 0]).flatten().float()
    qs = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], device=dev)
    print(wl, "tiles", n.numel(), "M", int(n.sum()), "quantiles", [int(v) for v in torch.quantile(n, qs)],
          "n>512", int((n > 512).sum()), "n>1024", int((n > 1024).sum()), "n>2048", int((n > 2048).sum()),
          "n>4096", int((n > 4096).sum()), "share of M in tiles>1024", round(float(n[n > 1024].sum() / n.sum()), 3))
