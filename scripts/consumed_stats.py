"""How deep into its list does the rasteriser actually go?  Per tile: the largest list position any
of its pixels blended (from last_ids) against the list length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from bench import WORKLOADS
from mojosplat_amd.rasterization import rasterize_gaussians_hip
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["cfg3"]:
    N, W, H, ell, fp16 = WORKLOADS[wl]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, H, W, 16, backend="hip")
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    img, alphas, last = rasterize_gaussians_hip(m2, con, sc["features"], sc["opacities"], bg, ranges, ids, cam, 16, return_aux=True)
    th, tw = ranges.shape[:2]
    Hp, Wp = th * 16, tw * 16
    lp = torch.zeros(Hp, Wp, dtype=torch.int64, device=dev)
    lp[:H, :W] = last
    tmax = lp.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th, tw, 256).max(-1).values
    n = (ranges[..., 1] - ranges[..., 0]).long()
    used = (tmax - ranges[..., 0].long() + 1).clamp(min=0)
    used = torch.where(n > 0, torch.minimum(used, n), torch.zeros_like(used))
    heavy = n > 1024
    a_min = (1 - alphas).view(-1)
    print(wl, "heavy tiles", int(heavy.sum()), "consumed quantiles (heavy)",
          [int(v) for v in torch.quantile(used[heavy].float(), torch.tensor([.5, .9, .99, 1.0], device=dev))] if heavy.any() else [],
          "heavy tiles consuming >1024:", int((used[heavy] > 1024).sum()), ">2048:", int((used[heavy] > 2048).sum()),
          "| sum consumed / M:", round(float(used.sum()) / max(int(n.sum()), 1), 3))
