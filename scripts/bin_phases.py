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

CFG = {"cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg4": (6_000_000, 1600, 1063, -4.0),
       "cfg5": (5_000_000, 3840, 2160, -4.0)}
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
    if os.environ.get("SCENE_ORDER") == "morton":   # (round 5: the same scene sorted along a Morton curve of its means)
        from mojosplat_amd.scene_order import morton_permutation
        perm = morton_permutation(sc["means3d"])
        sc = {k: v[perm].contiguous() for k, v in sc.items()}
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    # python scripts/bin_phases.py cfg3 8 3: rank 3's band of the frame cut 8 ways (the sharded entry point, rehearsed)
    if len(sys.argv) > 3:
        from mojosplat_amd.distributed import render_gaussians_sharded
        world, rank = int(sys.argv[2]), int(sys.argv[3])
        frame = lambda: render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(rank, world))
    else:
        frame = lambda: ms.render_gaussians(*g, cam, background_color=bg)
    for _ in range(8):
        frame()
    buf = torch.zeros(5 * 1024 * 8, dtype=torch.int64, device=dev)   # (slot 4: k_band_precull, scripts/precull_phases.py)
    _hip.check(L.ms_diag_set_bin_stamps(ctypes.c_void_p(buf.data_ptr())), "diag")
    frame()
    torch.cuda.synchronize()
    _hip.check(L.ms_diag_set_bin_stamps(None), "diag")
    raw = buf.cpu().numpy()
    # k_tile_front (kernel 3): up to three blocks per slot (blockIdx & 1023, blockIdx >> 10): start, end, list length
    fr = raw[3 * 1024 * 8:4 * 1024 * 8].reshape(1024, 8)
    rows = []
    for q in range(2):
        blk = fr[:, 3 * q:3 * q + 3]
        rows.append(blk[blk[:, 0] != 0])
    fr = np.concatenate(rows).astype(np.float64)
    if len(fr):
        t0 = fr[:, 0].min()
        life = (fr[:, 1] - fr[:, 0]) / 100.0
        out = {"kernel": "k_tile_front<merged>", "workgroups": int(len(fr)), "first_start_to_last_end_us": round((fr[:, 1].max() - t0) / 100.0, 2)}
        for lo, hi in ((0, 256), (256, 1024), (1024, 2048), (2048, 4096), (4096, 1 << 30)):
            m = (fr[:, 2] > lo) & (fr[:, 2] <= hi)
            if m.any():
                out[f"n in ({lo}, {hi}]"] = {"count": int(m.sum()), "life_us_p50": round(float(np.percentile(life[m], 50)), 2),
                                           "life_us_max": round(float(life[m].max()), 2),
                                           "start_us_p50": round(float(np.percentile((fr[m, 0] - t0) / 100.0, 50)), 2),
                                           "start_us_max": round(float(((fr[m, 0] - t0) / 100.0).max()), 2)}
        rel0, rel1 = (fr[:, 0] - t0) / 100.0, (fr[:, 1] - t0) / 100.0
        out["resident_workgroups_at_us"] = {str(t): int(((rel0 <= t) & (rel1 > t)).sum()) for t in (1, 3, 5, 7, 9, 11, 13)}
        print(json.dumps(out), flush=True)
    d = raw[:3 * 1024 * 8].reshape(3, 1024, 8).astype(np.float64)
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
