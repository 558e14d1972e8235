"""How many tiles does the lazy-sorting clean-up pass redo per frame?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from bench import WORKLOADS
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["cfg3"]:
    N, W, H, ell, fp16 = WORKLOADS[wl]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    if fp16:
        sc["features"] = sc["features"].half()
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    for i in range(4):
        ms.render_gaussians(*g, cam, background_color=bg)
        torch.cuda.synchronize()
        host = _fused._dev_state(dev, 0)["host"]
        print(wl, "frame", i, "M", int(host[0]), "heavy", int(host[2] + host[3] + host[4]), "redo tiles of previous frame", int(host[5]) & 0xffffffff, "cut redos", (int(host[5]) >> 32) & 0x3fffffff,
              "flags", int(host[7]), flush=True)
