"""Host time of ms_render_fwd's enqueueing half by phase (diagnostic build, MS_HOST_PROF=1):
    python -m mojosplat_amd.csrc.build --diag
    MS_HOST_PROF=1 MOJOSPLAT_HIP_LIB=mojosplat_amd/csrc/libmojosplat_hip_diag.so python scripts/host_prof.py [cfg3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
dev = torch.device("cuda:0")
if name == "tiny":
    sc, cam = randscene_v1(2000, 128, 128, ell=-3.0, seed=1, device=dev)
else:
    N, W, H, ell, fp16 = WORKLOADS[name]
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
for _ in range(50):
    ms.render_gaussians(*g, cam, background_color=bg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    ms.render_gaussians(*g, cam, background_color=bg)
torch.cuda.synchronize()
print(name, "render_gaussians us/frame", round((time.perf_counter() - t0) / 3000 * 1e6, 1))
