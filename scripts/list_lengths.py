import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from bench import WORKLOADS
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda", 0)
for wl in sys.argv[1:] or ["cfg3"]:
    N, W, H, ell, fp16 = WORKLOADS[wl]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"].half() if fp16 else sc["features"])
    for _ in range(6):
        ms.render_gaussians(*g, cam, background_color=bg)
    torch.cuda.synchronize()
    h = _fused._state[(dev, 0)]["host_np"]
    print(wl, "pairs", int(h[0]), "longest list", int(h[1]), "lists > 1024", int(h[2]), "> 8192", int(h[3]), "> 16384", int(h[4]), "flags", hex(int(h[7])))
    _fused.release_scratch()
