"""Which frame of a fresh lane first takes the depth cut: python scripts/cut_when.py cfg4 [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1, BACKGROUND_V1
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
bg = torch.tensor(BACKGROUND_V1, device=dev).to(sc["features"].dtype)
_fused._state.clear(); ms.render._bin_mode.clear(); ms.render._bin_left.clear()
for k in range(frames):
    ms.render_gaussians(*g, cam, background_color=bg)
    torch.cuda.synchronize()
    h = _fused._state[(dev, 0)]["host_np"]
    print(k, "mode", dict(ms.render._bin_mode), "pairs", int(h[0]), "flags", hex(int(h[7])), "cut", bool(int(h[7]) & 64), "h5", hex(int(h[5])))
