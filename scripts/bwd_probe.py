import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused, render as render_mod
import bench
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0")
# (round 6: python scripts/bwd_probe.py [cfg3 | cfg4 | cfg5] -- the step at the other BASELINE scene sizes, float32 colours)
WL = {"cfg3": (1_000_000, 1920, 1080), "cfg4": (6_000_000, 1600, 1063), "cfg5": (5_000_000, 3840, 2160)}
N, W, H = WL[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
sc, cam = randscene_v1(N, W, H, ell=-4.0, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
ms.render_gaussians(*g, cam, background_color=bg)
out = bench.fwd_bwd_leg(_fused, render_mod, g, cam, bg, N, W, H, -(-W // 16) * -(-H // 16), dev)
print(json.dumps({k: out[k] for k in ("ms_per_step_mean", "ms_per_step_median", "ms_per_step_streamed", "stage_us", "pairs_on_lists")}))
