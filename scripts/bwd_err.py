"""Config-3 gradients of the fused differentiable frame against the per-stage autograd functions: per tensor the max error
relative to the tensor's max (the max-norm bar), and -- round 5 -- the per-ELEMENT relative error over the elements of at
least 1e-3 of the tensor's max: its maximum and its 99.9th percentile (tests/helpers.py::grad_stats).
    python scripts/bwd_err.py [N W H ell]"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from helpers import grad_stats
from mojosplat_amd.autograd import render_gaussians_trainable
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0")
N, W, H, ell = 1_000_000, 1920, 1080, -4.0
if len(sys.argv) > 4:
    N, W, H, ell = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
names = ("means3d", "scales", "quats", "opacities", "features")
v_img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(43)).to(dev)
res = []
for stagewise in (False, False, True):
    leaves = [sc[k].clone().requires_grad_(True) for k in names]
    img = render_gaussians_trainable(*leaves, cam, background_color=bg, stagewise=stagewise)
    img.backward(v_img)
    res.append([l.grad for l in leaves])
out = {"scene": [N, W, H, ell]}
for name, a, b, c in zip(names, *res):
    rep, vs = grad_stats(a, b), grad_stats(a, c)
    out[name] = {"scale": vs["scale"], "repeat_err": rep["max_norm_err"], "vs_stagewise": vs["max_norm_err"],
                 "elem_checked": vs["checked"], "elem_rel_max": vs["elem_rel_max"], "elem_rel_p999": vs["elem_rel_p999"],
                 "repeat_elem_rel_max": rep["elem_rel_max"], "finite": bool(torch.isfinite(a).all())}
print(json.dumps(out))
