"""How the (Gaussian, bin) pairs of a workload split over box sizes: python scripts/box_stats.py cfg4 32"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd.scenes import randscene_v1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
px = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
tw, th = -(-W // px), -(-H // px)
ok = (rad > 0).all(-1) if rad.dim() == 2 else rad > 0
r = rad.float() if rad.dim() == 2 else rad.float()[:, None].expand(-1, 2)
x0 = ((m2[:, 0] - r[:, 0]) / px).floor().clamp(0, tw); x1 = ((m2[:, 0] + r[:, 0]) / px).ceil().clamp(0, tw)
y0 = ((m2[:, 1] - r[:, 1]) / px).floor().clamp(0, th); y1 = ((m2[:, 1] + r[:, 1]) / px).ceil().clamp(0, th)
n = ((x1 - x0) * (y1 - y0)).long() * ok.long()
tot = int(n.sum())
out = {"config": name, "bin_px": px, "gaussians": N, "on_grid": int((n > 0).sum()), "box_pairs": tot}
edges = [1, 2, 4, 9, 16, 32, 64, 256, 1 << 30]
lo = 0
for e in edges:
    sel = (n > lo) & (n <= e)
    out[f"boxes_{lo + 1}_to_{e if e < 1 << 30 else 'inf'}"] = {"gaussians": int(sel.sum()), "pair_share": round(float(n[sel].sum()) / max(tot, 1), 4)}
    lo = e
# per wave of 64 consecutive Gaussians: the largest own-lane box (n <= 32) and the whole-wave passes for the big ones
nn = n[: (N // 64) * 64].view(-1, 64)
small = torch.where(nn <= 32, nn, torch.zeros_like(nn))
big_passes = torch.where(nn > 32, (nn + 63) // 64, torch.zeros_like(nn)).sum(1)
out["per_wave_step"] = {"max_small_box_mean": round(float(small.max(1).values.float().mean()), 2), "sum_small_mean": round(float(small.sum(1).float().mean()), 1),
                        "big_box_passes_mean": round(float(big_passes.float().mean()), 2), "big_box_passes_max": int(big_passes.max())}
print(json.dumps(out))
