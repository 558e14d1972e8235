"""Per-workgroup phase stamps of the binning kernels of one config-3 frame (diagnostic build, -DMS_DIAG):

    python -m mojosplat_amd.csrc.build --diag
    MOJOSPLAT_HIP_LIB=mojosplat_amd/csrc/libmojosplat_hip_diag.so python scripts/bin_phases.py [cfg3]

For k_project_hist (0), k_tile_scan_wg (1), k_isect_scatter (2): when each workgroup started relative to the
kernel's first, and how long each phase between two stamps took (percentiles over the workgroups, microseconds).
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd import _hip  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402

CFG = {"cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg5": (5_000_000, 3840, 2160, -4.0)}
NAMES = {0: ("k_project_hist", ["loop", "barrier", "row out"]),
         1: ("k_tile_scan_wg", ["loads + wave scan", "barrier", "combine + stores"]),
         2: ("k_isect_scatter", ["prefetch issue", "tile prefix", "walk + stores", "last wave"])}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    N, W, H, ell = CFG[name]
    dev = torch.device("cuda:0")
    L = _hip.lib()
    assert hasattr(L, "ms_diag_set_bin_stamps"), "load the -DMS_DIAG build through MOJOSPLAT_HIP_LIB"
    L.ms_diag_set_bin_stamps.argtypes = [ctypes.c_void_p]
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    for _ in range(8):
        ms.render_gaussians(*g, cam, background_color=bg)
    buf = torch.zeros(3 * 1024 * 8, dtype=torch.int64, device=dev)
    _hip.check(L.ms_diag_set_bin_stamps(ctypes.c_void_p(buf.data_ptr())), "diag")
    ms.render_gaussians(*g, cam, background_color=bg)
    torch.cuda.synchronize()
    _hip.check(L.ms_diag_set_bin_stamps(None), "diag")
    d = buf.cpu().numpy().reshape(3, 1024, 8).astype(np.float64)
    pct = lambda v: {k: round(float(np.percentile(v, q)) / 100.0, 2) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}
    for k, (kname, phases) in NAMES.items():
        wg = d[k][d[k][:, 0] != 0]
        if not len(wg):
            continue
        t0 = wg[:, 0].min()
        last = max(wg[:, j].max() for j in range(8))
        out = {"kernel": kname, "workgroups": int(len(wg)), "first_start_to_last_stamp_us": round((last - t0) / 100.0, 2),
               "start_offset_us": pct(wg[:, 0] - t0)}
        for j, ph in enumerate(phases):
            ok = (wg[:, j + 1] != 0) & (wg[:, j] != 0)
            if ok.any():
                out[ph + "_us"] = pct(wg[ok, j + 1] - wg[ok, j])
        ok = wg[:, 1] != 0
        ends = np.max(wg[:, 1:], axis=1)
        out["workgroup_life_us"] = pct(ends - wg[:, 0])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
