"""Per-wave picture of k_rasterize_fwd on a BASELINE config (diagnostic build, -DMS_DIAG):
when each wave started and ended (shader clock), how many (quad, entry) evaluations and batches it ran.

    python -m mojosplat_amd.csrc.build --diag
    MOJOSPLAT_HIP_LIB=mojosplat_amd/csrc/libmojosplat_hip_diag.so python scripts/raster_waves.py [cfg3]

Answers: how much of the kernel is tail (few waves resident), cycles per evaluation at full and at low
occupancy, what a work-ordered launch could gain.
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

CFG = {"cfg5edge": (5_000_000, 3840, 2160, -4.0),   # the first of eight balanced bands of config 5 (rows 0-16 of 135)
       "cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg5": (5_000_000, 3840, 2160, -4.0),
       "cfg3-heavy": (1_000_000, 1920, 1080, -3.0)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    N, W, H, ell = CFG[name]
    dev = torch.device("cuda:0")
    L = _hip.lib()
    assert hasattr(L, "ms_diag_set_stamps"), "load the -DMS_DIAG build through MOJOSPLAT_HIP_LIB"
    L.ms_diag_set_stamps.argtypes = [ctypes.c_void_p]
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    band = None
    if name == "cfg5edge":
        from mojosplat_amd import _fused
        band = (0, 17)

    def frame():
        if band is None:
            return ms.render_gaussians(*g, cam, background_color=bg)
        return _fused.render_fwd_hip(*g, cam, bg, 16, row_range=band)
    for _ in range(5):
        frame()
    nmax = 4 * (-(-W // 16) + 2) * (-(-H // 16) + 2)
    buf = torch.zeros(nmax * 8, dtype=torch.int64, device=dev)
    _hip.check(L.ms_diag_set_stamps(ctypes.c_void_p(buf.data_ptr())), "diag")
    frame()
    torch.cuda.synchronize()
    _hip.check(L.ms_diag_set_stamps(None), "diag")
    d = buf.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 1] != 0]
    t0, t1 = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
    evals, batches = (d[:, 2] & 0xffffffff).astype(np.float64), (d[:, 2] >> 32).astype(np.float64)
    listlen = (d[:, 3] >> 32).astype(np.float64)
    xcc = (d[:, 3] >> 24) & 0xf
    cyc = d[:, 4].astype(np.float64)      # shader cycles of each wave's life
    base = t0.min()                       # t0 / t1: 100 MHz ticks, chip-wide
    # s_memtime ticks are shader cycles (MI355X_MICROARCH.md, cycle constants)
    span = t1.max() - base
    clock_ghz = float(np.median(cyc[cyc > 2000] / (t1 - t0)[cyc > 2000])) / 10.0
    out = {"config": name, "waves": int(len(d)), "kernel_us": float(span) / 100.0, "clock_ghz": clock_ghz, "evals_total": float(evals.sum()),
           "batches_total": float(batches.sum()), "list_entries_total": float(listlen.sum())}
    life = t1 - t0
    out["wave_life_cycles"] = {k: float(np.percentile(cyc, q)) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
    out["wave_life_ticks"] = {k: float(np.percentile(life, q)) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
    out["evals_per_wave"] = {k: float(np.percentile(evals, q)) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
    # residency over time: waves in flight at 100 sample points
    ts = np.linspace(base, t1.max(), 101)[:-1]
    res = [(int(((t0 <= t) & (t1 > t)).sum())) for t in ts]
    out["resident_waves_at_percent_of_kernel"] = {str(p): res[p] for p in (0, 5, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 99)}
    out["mean_resident_waves"] = float(life.sum() / span)
    out["last_start_percent"] = float((t0.max() - base) / span * 100)
    # ticks per evaluation as a function of how crowded the chip was during the wave's life
    mid = (t0 + t1) / 2
    crowd = np.array([res[min(99, int((m - base) / span * 100))] for m in mid])
    sel = evals > 50
    tpe = cyc[sel] / evals[sel]
    for lo, hi in ((0, 1000), (1000, 3000), (3000, 6000), (6000, 9000)):
        m = (crowd[sel] >= lo) & (crowd[sel] < hi)
        if m.any():
            out[f"cycles_per_eval_when_{lo}_{hi}_waves_resident"] = float(np.median(tpe[m]))
    # round 6: the start-up chain (wave start -> first batch staged: tile range -> ids -> records -> LDS, three dependent
    # round trips) against the wave's life
    first = d[:, 5].astype(np.float64)
    has = first > 0
    if has.any():
        out["startup_cycles"] = {k: float(np.percentile(first[has], q)) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99))}
        out["startup_share_of_wave_life_p50"] = float(np.median(first[has] / np.maximum(cyc[has], 1.0)))
        out["startup_share_of_all_wave_cycles"] = float(first[has].sum() / cyc.sum())
        out["waves_with_an_empty_list"] = int((~has).sum())
        one = has & (batches <= 1)
        out["one_batch_waves"] = int(one.sum())
        if one.any():
            out["one_batch_wave_life_cycles_p50"] = float(np.median(cyc[one]))
            out["one_batch_startup_cycles_p50"] = float(np.median(first[one]))
    out["xcc_wave_counts"] = np.bincount(xcc, minlength=8).tolist()
    out["xcc_end_percent"] = [float((t1[xcc == x].max() - base) / span * 100) if (xcc == x).any() else None for x in range(8)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
