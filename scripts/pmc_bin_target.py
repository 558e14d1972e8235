"""Per-stage binning of config 3, a few times: a target for PMC passes on the binning kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd.scenes import randscene_v1
dev = torch.device("cuda", 0)
sc, cam = randscene_v1(1_000_000, 1920, 1080, ell=-4.0, seed=42, device=dev)
m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
for _ in range(6):
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, 1080, 1920, 16, backend="hip")
torch.cuda.synchronize()
