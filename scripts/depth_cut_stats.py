"""How many (Gaussian, bin) pairs a per-bin depth cut-off would drop: for every bin of a config's grid, the pairs behind
the depth at which its first `front` entries end (x `margin` in entries).  python scripts/depth_cut_stats.py cfg4 32"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd.binning import bin_gaussians_to_tiles_hip
from mojosplat_amd.scenes import randscene_v1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
px = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
th, tw = -(-H // px), -(-W // px)
ids, ranges = bin_gaussians_to_tiles_hip(m2, rad, dep, px, tw, th)
cnt = (ranges[..., 1] - ranges[..., 0]).flatten().long()
front = 1536 if px == 32 else 2048 if px == 64 else 1024
out = {"config": name, "bin_px": px, "pairs_gsplat_boxes": int(ids.numel()), "bins": int(cnt.numel()), "heavy_bins": int((cnt > 1024).sum())}
for margin in (1.0, 1.25, 1.5, 2.0):
    keep = torch.minimum(cnt, torch.full_like(cnt, int(front * margin)))
    keep = torch.where(cnt > 1024, keep, cnt)
    out[f"kept_fraction_margin_{margin}"] = round(float(keep.sum()) / max(1, int(cnt.sum())), 3)
print(json.dumps(out))
