"""Frames of one workload in the given or the Morton order, for rocprofv3 (--kernel-trace --stats / --pmc):
    python scripts/morton_kernels.py cfg3 given|morton [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd.scenes import randscene_v1, BACKGROUND_V1
from mojosplat_amd.scene_order import morton_permutation
from bench import WORKLOADS

name, order = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 120
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
if order == "morton":
    perm = morton_permutation(sc["means3d"])
    sc = {k: v[perm].contiguous() for k, v in sc.items()}
bg = torch.tensor(BACKGROUND_V1, device=dev).to(sc["features"].dtype)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
for _ in range(frames):
    ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
torch.cuda.synchronize()
