"""Frame time of render_gaussians over a grid of scene shapes (N, log-scale): does the path (binning rule, lazily
sorted fronts, rasteriser shape) behave away from the BASELINE configs?  python scripts/scene_scan.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms
from mojosplat_amd import render as R
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0")
bg = torch.tensor(BACKGROUND_V1, device=dev)
for N, ell, W, H in ((10_000, -2.0, 1920, 1080), (100_000, -2.0, 1920, 1080), (100_000, -3.0, 1920, 1080), (1_000_000, -2.5, 1920, 1080),
                     (1_000_000, -3.5, 1920, 1080), (1_000_000, -5.0, 1920, 1080), (3_000_000, -4.0, 1280, 720), (200_000, -1.5, 640, 360)):
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    out = {"N": N, "ell": ell, "WH": [W, H]}
    for mode in (None, 16, 32, 64):
        for _ in range(8):
            ms.render_gaussians(*g, cam, background_color=bg, bin_size=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            ms.render_gaussians(*g, cam, background_color=bg, bin_size=mode)
        torch.cuda.synchronize()
        out["rule" if mode is None else str(mode)] = round((time.perf_counter() - t0) / 30 * 1e6, 1)
    out["rule_chose"] = R._bin_mode.get(R._bin_key(g[0], cam))
    import math
    from mojosplat_amd import _fused
    est = {}
    for mode in (16, 32, 64):
        info = {}
        for _ in range(2):
            _, m = _fused.render_fwd_hip(*g, cam, bg, mode, info=info)
        gpx = 32 if mode == 16 else mode
        est[mode] = (m, info["on_grid"], round(gpx * (math.sqrt(max(m / max(info["on_grid"], 1), 1.0)) - 1.0), 1))
    out["M_ongrid_d"] = est
    print(json.dumps(out), flush=True)
    del sc, g
