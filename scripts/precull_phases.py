"""Per-workgroup phase stamps of k_band_precull for one rank's band (diagnostic build, -DMS_DIAG):

    MOJOSPLAT_HIP_LIB=mojosplat_amd/csrc/libmojosplat_hip_diag.so SCENE_ORDER=prepared python scripts/precull_phases.py cfg5 8 3

When each workgroup started relative to the first, how long its block verdicts and its walk took, how many survivors it
wrote -- and the same for the longest-lived workgroups."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from mojosplat_amd import _hip
from mojosplat_amd.distributed import render_gaussians_sharded
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
name, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda", 0)
L = _hip.lib()
assert hasattr(L, "ms_diag_set_bin_stamps"), "load the -DMS_DIAG build through MOJOSPLAT_HIP_LIB"
L.ms_diag_set_bin_stamps.argtypes = [ctypes.c_void_p]
sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
bg = torch.tensor(BACKGROUND_V1, device=dev)
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
order = os.environ.get("SCENE_ORDER", "given")
if order == "prepared":
    from mojosplat_amd.scene_order import prepare_scene
    g = prepare_scene(*g).arrays
frame = lambda: render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(rank, world))
for _ in range(8):
    frame()
buf = torch.zeros(5 * 1024 * 8, dtype=torch.int64, device=dev)
_hip.check(L.ms_diag_set_bin_stamps(ctypes.c_void_p(buf.data_ptr())), "diag")
frame()
torch.cuda.synchronize()
_hip.check(L.ms_diag_set_bin_stamps(None), "diag")
d = buf.cpu().numpy()[4 * 1024 * 8:].reshape(1024, 8).astype(np.float64)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
start, verdict, walk, life, wrote = (d[:, 0] - t0) / 100, (d[:, 1] - d[:, 0]) / 100, (d[:, 2] - d[:, 1]) / 100, (d[:, 2] - d[:, 0]) / 100, d[:, 3]
pct = lambda v: {k: round(float(np.percentile(v, q)), 2) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}
out = {"kernel": "k_band_precull", "workload": name, "order": order, "world": world, "rank": rank, "workgroups": int(len(d)),
       "first_start_to_last_end_us": round(float((d[:, 2].max() - t0) / 100), 2), "start_us": pct(start), "verdicts_us": pct(verdict),
       "walk_us": pct(walk), "life_us": pct(life), "survivors": pct(wrote), "survivors_total": int(wrote.sum())}
idx = np.argsort(-life)[:6]
out["longest"] = [{"wg": int(i), "start": round(float(start[i]), 2), "verdicts": round(float(verdict[i]), 2), "walk": round(float(walk[i]), 2),
                   "survivors": int(wrote[i])} for i in idx]
c = np.corrcoef(wrote, walk)[0, 1] if len(d) > 2 and wrote.std() > 0 else float("nan")
out["corr_survivors_walk"] = round(float(c), 3)
for lo, hi in ((0, 1), (1, 2000), (2000, 6000), (6000, 1 << 30)):
    m = (wrote >= lo) & (wrote < hi)
    if m.any():
        out[f"survivors in [{lo}, {hi})"] = {"count": int(m.sum()), "walk_us_p50": round(float(np.percentile(walk[m], 50)), 2), "walk_us_max": round(float(walk[m].max()), 2)}
print(json.dumps(out))
