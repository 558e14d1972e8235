"""Per-bin cut-offs of a depth-cut frame against what the bins hold: python scripts/depth_cut_debug.py cfg4 32"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mojosplat_amd import _hip
import mojosplat_amd as ms
from mojosplat_amd import _fused, _hip
from mojosplat_amd.scenes import randscene_v1
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
px = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
tw, th = -(-W // px), -(-H // px)
T = tw * th
out = {}
for mode in ("0", "2"):
    _hip.config_depth_cut(int(mode))
    _fused._state.clear()
    for _ in range(5):
        ms.render_gaussians(*g, cam, backend="hip", bin_size=px)
    torch.cuda.synchronize()
    st = _fused._dev_state(dev, 0)
    off = (ctypes.c_size_t * 6)()
    _hip.lib().ms_render_workspace_layout(N, tw, th, off)
    ranges = st["ws"][off[4]:off[4] + T * 8].view(torch.int32).view(T, 2).long()
    out[mode] = (ranges[:, 1] - ranges[:, 0]).cpu()
    keys = st["isect"][: int(ranges[:, 1].max()) * 8].view(torch.int64)
    depth = (keys >> 32).to(torch.int32).view(torch.float32)
    cnt = out[mode]
    pick = [int(torch.argmax(full_cnt if mode == "2" else cnt)), int(torch.argsort(full_cnt if mode == "2" else cnt)[int(T * 0.75)])]
    for t in pick:
        a, b = int(ranges[t, 0]), int(ranges[t, 1])
        d = depth[a:b].sort().values
        print(json.dumps({"mode": mode, "bin": t, "entries": b - a, "depth_first": float(d[0]), "depth_at_1536": float(d[min(1535, b - a - 1)]),
                          "depth_at_2048": float(d[min(2047, b - a - 1)]), "depth_last": float(d[-1])}))
    if mode == "0":
        full_cnt = cnt
full, near = out["0"], out["2"]
heavy = full > 1024
print(json.dumps({"config": name, "bin_px": px, "bins": T, "pairs_uncut": int(full.sum()), "pairs_cut_frame": int(near.sum()),
                  "bins_over_1024": int(heavy.sum()), "pairs_in_bins_over_1024": int(full[heavy].sum()),
                  "bins_cut_at_all": int((near < full).sum()),
                  "kept_in_heavy_bins_p10_p50_p90": [int(x) for x in torch.quantile(near[heavy].float(), torch.tensor([0.1, 0.5, 0.9])).tolist()] if heavy.any() else None,
                  "full_heavy_bins_p10_p50_p90": [int(x) for x in torch.quantile(full[heavy].float(), torch.tensor([0.1, 0.5, 0.9])).tolist()] if heavy.any() else None}))
