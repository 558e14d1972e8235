"""Frame time of every binning granularity on the BASELINE scenes: python scripts/bin_modes.py [cfg...]
(run once as is and once with MOJOSPLAT_SPLIT=0 to see 16-px tiles binned directly)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mojosplat_amd import _fused
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
from scripts.config_sweep import CFG

dev = torch.device("cuda", 0)
bg = torch.tensor(BACKGROUND_V1, device=dev)
for name in sys.argv[1:] or list(CFG):
    N, W, H, ell, fp16 = CFG[name]
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    if fp16:
        sc["features"] = sc["features"].half()
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    row = {}
    for ts in [int(x) for x in os.environ.get("MODES", "16,32,64").split(",")]:
        _fused._state.clear()
        for _ in range(6):
            img, m = _fused.render_fwd_hip(*g, cam, bg, ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            img, m = _fused.render_fwd_hip(*g, cam, bg, ts)
        torch.cuda.synchronize()
        st = _fused._dev_state(dev, 0)
        row[ts] = (round((time.perf_counter() - t0) / 40 * 1e3, 4), m, st.get("front_level", 0), bool(st.get("full_sort")))
    print(name, "split" if os.environ.get("MOJOSPLAT_SPLIT", "1") != "0" else "nosplit", row, flush=True)
    del sc, g
