"""Raster kernel time per waves-per-block setting (MOJOSPLAT_RASTER_PARTS, read once per process):
run as   for p in 0 1 2 4; do MOJOSPLAT_RASTER_PARTS=$p python scripts/raster_parts.py cfg3; done"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms
from mojosplat_amd import render as R
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
CFG = {"cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg5": (5_000_000, 3840, 2160, -4.0),
       "cfg3-heavy": (1_000_000, 1920, 1080, -3.0), "cfg4": (6_000_000, 1600, 1063, -4.0)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N, W, H, ell = CFG[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if name == "cfg4":
    sc["features"] = sc["features"].half()
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
for _ in range(10):
    ms.render_gaussians(*g, cam, background_color=bg)
evs = []
def hook():
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for x in e: x.record()
    evs.append(e)
    return e
torch.cuda.synchronize()
R._STAGE_HOOK = hook
for _ in range(30):
    ms.render_gaussians(*g, cam, background_color=bg)
torch.cuda.synchronize()
R._STAGE_HOOK = None
med = lambda v: sorted(v)[len(v) // 2]
st = {n: round(med([e[i].elapsed_time(e[i + 1]) * 1e3 for e in evs]), 1) for i, n in enumerate(("project", "bin", "raster"))}
import time
t0 = time.perf_counter()
for _ in range(200):
    ms.render_gaussians(*g, cam, background_color=bg)
torch.cuda.synchronize()
print(json.dumps({"config": name, "parts": os.environ.get("MOJOSPLAT_RASTER_PARTS", "auto"), "stage_us": st,
                  "frame_us": round((time.perf_counter() - t0) / 200 * 1e6, 1)}))
