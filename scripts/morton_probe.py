"""Does the ORDER of the Gaussians in the caller's arrays matter?  The same scene in its given order and sorted along a
3-D Morton curve of the means (a caller-side permutation of all five arrays: nothing in the library changes): blocking
frames, and one rank's band of eight through the sharded entry point.
    python scripts/morton_probe.py cfg5 [band rank]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojosplat_amd as ms
from mojosplat_amd import _fused
from mojosplat_amd.distributed import render_gaussians_sharded
from mojosplat_amd.scenes import randscene_v1, BACKGROUND_V1
from bench import WORKLOADS
from mojosplat_amd.scene_order import morton_permutation as morton_perm

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, W, H, ell, fp16 = WORKLOADS[name]
dev = torch.device("cuda:0")
sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
if fp16:
    sc["features"] = sc["features"].half()
bg = torch.tensor(BACKGROUND_V1, device=dev).to(sc["features"].dtype)


def _unused_morton_perm(p):
    q = ((p - p.min(0).values) / (p.max(0).values - p.min(0).values + 1e-9) * 1023.0).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)


def timed(fn, n):
    for _ in range(60):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)


out = {"config": name}
perm = morton_perm(sc["means3d"])
for label, s_ in (("given_order", sc), ("morton_order", {k: v[perm].contiguous() for k, v in sc.items()})):
    g = (s_["means3d"], s_["scales"], s_["quats"], s_["opacities"], s_["features"])
    _fused._state.clear()
    out[label + "_frame_ms"] = timed(lambda: ms.render_gaussians(*g, cam, background_color=bg, backend="hip"), 300)
    _fused._state.clear()
    def pipelined(n, cur=[None]):
        for _ in range(n):
            nxt = render_gaussians_sharded(*g, cam, background_color=bg, async_op=True, rehearse=(rank, 8))
            if cur[0] is not None:
                cur[0].wait()
            cur[0] = nxt
    pipelined(60)
    torch.cuda.synchronize(); t0 = time.perf_counter(); pipelined(300); torch.cuda.synchronize()
    out[label + f"_band{rank}of8_pipelined_ms"] = round((time.perf_counter() - t0) / 300 * 1e3, 4)
print(json.dumps(out))
