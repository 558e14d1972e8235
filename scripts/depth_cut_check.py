"""Depth-cut frames against uncut ones on one workload: python scripts/depth_cut_check.py cfg3 [bin_px]
static camera, a short orbit, and a swap to a different scene between two frames (stale cut-offs); prints the lane's
frame statistics (depth_cut frames, cut_redo_tiles) and whether every frame equals its uncut twin bit for bit."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.scenes import randscene_v1
from mojosplat_amd.utils import Camera
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
px = int(sys.argv[2]) if len(sys.argv) > 2 else None
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
g = lambda s: (s["means3d"], s["scales"], s["quats"], s["opacities"], s["features"])


def orbit(cam, a):
    c, s_ = math.cos(a), math.sin(a)
    Ry = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]], device=dev)
    return Camera(R=cam.R @ Ry, T=cam.T, H=cam.H, W=cam.W, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy)


def frames(mode, seq):
    _hip.config_depth_cut(int(mode))
    _fused._state.clear()
    _fused.FRAME_STATS = {}
    out = []
    for s, c in seq:
        out.append(ms.render_gaussians(*g(s), c, backend="hip", bin_size=px).clone())
    torch.cuda.synchronize()
    st, _fused.FRAME_STATS = _fused.FRAME_STATS, None
    return out, st


# a second scene on the same shapes: the same Gaussians, the nearest third made all but transparent
sc2 = dict(sc)
depth = (sc["means3d"] @ cam.R.T + cam.T)[:, 2] if hasattr(cam, "R") else sc["means3d"][:, 2]
near = depth < depth.median()
sc2["opacities"] = torch.where(near, sc["opacities"] * 0.02, sc["opacities"])
seqs = {
    "static": [(sc, cam)] * 6,
    "orbit": [(sc, orbit(cam, 0.004 * i)) for i in range(12)],
    "swap": [(sc, cam)] * 3 + [(sc2, cam)] * 3 + [(sc, cam)] * 2,
}
for label, seq in seqs.items():
    ref, st0 = frames("0", seq)
    got, st = frames("2", seq)
    same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
    worst = max(float((a - b).abs().max()) for a, b in zip(ref, got))
    print(json.dumps({"config": name, "bin_px": px, "sequence": label, "frames": len(seq), "all_equal": all(same), "equal": same,
                      "max_abs_diff": worst, "stats_cut": st, "stats_uncut": {k: st0[k] for k in ("frames", "redo_tiles") if k in st0}}))

# the split of the last depth-cut frame's pairs (device words 8 / 9 of the size record: near, far)
import ctypes
from mojosplat_amd import _hip
_hip.config_depth_cut(2)
_fused._state.clear()
for _ in range(4):
    ms.render_gaussians(*g(sc), cam, backend="hip", bin_size=px)
torch.cuda.synchronize()
st = _fused._dev_state(dev, 0)
bpx = px or 32
tw, th = -(-W // bpx), -(-H // bpx)
off = (ctypes.c_size_t * 6)()
_hip.lib().ms_render_workspace_layout(N, tw, th, off)
info_off = off[4] + ((tw * th * 8 + 255) // 256) * 256
info = st["ws"][info_off:info_off + 96].view(torch.int64).cpu().tolist()
print(json.dumps({"config": name, "bin_px": bpx, "pairs": info[0], "near": info[8], "far": info[9], "near_fraction": round(info[8] / max(1, info[0]), 3)}))
