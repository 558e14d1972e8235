"""What a frame costs when its depth cut-offs are stale (a scene swap): per-frame wall times, synchronised, around the swap.
python scripts/cut_miss_cost.py cfg4 64"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mojosplat_amd import _hip
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
px = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
other = dict(sc)
depth = (sc["means3d"] @ cam.R.T + cam.T)[:, 2]
other["opacities"] = torch.where(depth < depth.median(), sc["opacities"] * 0.02, sc["opacities"])   # the near half all but gone
g = lambda s: (s["means3d"], s["scales"], s["quats"], s["opacities"], s["features"])
for mode in ("1", "0"):
    _hip.config_depth_cut(int(mode))
    _fused._state.clear()
    _fused.FRAME_STATS = {}
    times = []
    for k in range(14):
        s = sc if k < 6 or k >= 10 else other
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms.render_gaussians(*g(s), cam, backend="hip", bin_size=px)
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(json.dumps({"config": name, "bin_px": px, "depth_cut": mode, "ms_per_frame_synchronised": times,
                      "swap_at_frames": [6, 10], "stats": _fused.FRAME_STATS}))
    _fused.FRAME_STATS = None
