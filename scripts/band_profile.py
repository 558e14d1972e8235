"""One rank's band of a frame through the sharded entry point (rehearse), many times, for
`rocprofv3 --kernel-trace --stats` (per-kernel picture of a band frame).
python scripts/band_profile.py cfg3 8 3   -> workload, world, rank"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from mojosplat_amd.distributed import render_gaussians_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
name, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda", 0)
sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
if os.environ.get("SCENE_ORDER") == "prepared":   # (round 5: Morton order + block bounds: scene_order.prepare_scene)
    from mojosplat_amd.scene_order import prepare_scene
    g = prepare_scene(*g).arrays
for _ in range(44):
    render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(rank, world))
torch.cuda.synchronize()
