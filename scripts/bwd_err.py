"""Config-3 gradients of the fused differentiable frame against the per-stage autograd functions: max error per tensor
relative to the tensor's max.  python scripts/bwd_err.py"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mojosplat_amd.autograd import render_gaussians_trainable
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1920, 1080
sc, cam = randscene_v1(N, W, H, ell=-4.0, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
names = ("means3d", "scales", "quats", "opacities", "features")
v_img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(43)).to(dev)
res = []
for stagewise in (False, False, True):
    leaves = [sc[k].clone().requires_grad_(True) for k in names]
    img = render_gaussians_trainable(*leaves, cam, background_color=bg, stagewise=stagewise)
    img.backward(v_img)
    res.append([l.grad for l in leaves])
out = {}
for name, a, b, c in zip(names, *res):
    scale = float(c.abs().max())
    out[name] = {"scale": scale, "repeat_err": float((a - b).abs().max()) / scale, "vs_stagewise": float((a - c).abs().max()) / scale,
                 "finite": bool(torch.isfinite(a).all())}
print(json.dumps(out))
