import os, sys, json, time, torch
sys.path.insert(0, os.getcwd())
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
dev = torch.device("cuda:0"); bg = torch.tensor(BACKGROUND_V1, device=dev)
for name, (N, W, H, ell, mode) in {"cfg3": (1_000_000, 1920, 1080, -4.0, 32), "cfg2": (100_000, 1920, 1080, -4.0, 32), "cfg4": (6_000_000, 1600, 1063, -4.0, 32), "cfg5": (5_000_000, 3840, 2160, -4.0, 64), "heavy": (1_000_000, 1920, 1080, -3.0, 64)}.items():
    sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
    for _ in range(8): _fused.render_fwd_hip(*g, cam, bg, mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): _fused.render_fwd_hip(*g, cam, bg, mode)
    torch.cuda.synchronize()
    st = _fused._dev_state(dev, 0)
    print(name, os.environ.get("MOJOSPLAT_FRONT_K"), round((time.perf_counter() - t0) / 40 * 1e6, 1), "redo(prev)", int(st["host"][5]), "level", st.get("front_level", 0), "full", bool(st.get("full_sort")), flush=True)
    del sc, g; _fused._state.clear(); torch.cuda.empty_cache()
