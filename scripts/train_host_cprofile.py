import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mojosplat_amd import autograd as ag
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0")
sc, cam = randscene_v1(1_000_000, 1920, 1080, ell=-4.0, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
leaves = [t.float().clone().requires_grad_(True) for t in g]
v_img = torch.rand(1080, 1920, 3).to(dev)
def step():
    for l in leaves: l.grad = None
    img = ag.render_gaussians_trainable(*leaves, cam, background_color=bg)
    img.backward(v_img)
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step(); torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue())
