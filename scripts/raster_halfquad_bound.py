"""Round 6: what would a rasteriser wave save if its two 32-lane halves walked lists of their OWN (half-quads of 8x4 or 4x8
pixels) instead of one list per 8x8 quad?  From the projected scene, without occlusion (an upper bound on every count alike):

    evals(quad)   = sum over quads of the Gaussians that reach the quad                 (a wave instruction stream per entry)
    evals(half)   = sum over quads of max(reach the first half, reach the second half)   (the halves walk in lockstep)
    ideal(half)   = sum over half-quads / 2                                              (if the halves were always balanced)

    python scripts/raster_halfquad_bound.py [cfg3]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd.scenes import randscene_v1  # noqa: E402

WORKLOADS = {"cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg5": (5_000_000, 3840, 2160, -4.0)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    N, W, H, ell = WORKLOADS[name]
    dev = torch.device("cuda:0")
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
    op = sc["opacities"].reshape(-1)
    vis = (rad[:, 0] > 0) & (rad[:, 1] > 0)
    idx = vis.nonzero().reshape(-1)
    R = min(int(rad[vis].max().item()), 40)
    offs = torch.arange(-R, R + 1, device=dev, dtype=torch.float32)
    qw, qh = -(-W // 8), -(-H // 8)
    n_quad = torch.zeros(qw * qh, dtype=torch.int64, device=dev)
    n_h84 = torch.zeros(qw * qh * 2, dtype=torch.int64, device=dev)   # 8 wide x 4 tall: index (quad, half)
    n_h48 = torch.zeros(qw * qh * 2, dtype=torch.int64, device=dev)   # 4 wide x 8 tall
    chunk = 20000 if N <= 1_000_000 else 8000
    for s in range(0, idx.numel(), chunk):
        g = idx[s:s + chunk]
        mx, my = m2[g, 0], m2[g, 1]
        cx, cy = torch.floor(mx), torch.floor(my)
        X = cx[:, None] + offs[None, :] + 0.5
        Y = cy[:, None] + offs[None, :] + 0.5
        dx = (mx[:, None] - X)[:, None, :]
        dy = (my[:, None] - Y)[:, :, None]
        sig = 0.5 * (con[g, 0, None, None] * dx * dx + con[g, 2, None, None] * dy * dy) + con[g, 1, None, None] * dx * dy
        alpha = torch.clamp(op[g, None, None] * torch.exp(-sig), max=0.999)
        inside = ((X >= 0) & (X < W))[:, None, :] & ((Y >= 0) & (Y < H))[:, :, None]
        hit = (alpha >= 1.0 / 255.0) & (sig >= 0) & inside
        px = torch.floor(X)[:, None, :].expand_as(hit).long()[hit]
        py = torch.floor(Y)[:, :, None].expand_as(hit).long()[hit]
        gi = torch.arange(g.numel(), device=dev)[:, None, None].expand_as(hit)[hit]
        quad = (py >> 3) * qw + (px >> 3)
        for arr, sub in ((n_quad, None), (n_h84, (py >> 2) & 1), (n_h48, (px >> 2) & 1)):
            cell = quad if sub is None else quad * 2 + sub
            key = torch.unique(gi * (arr.numel() + 1) + cell)
            arr += torch.bincount(key % (arr.numel() + 1), minlength=arr.numel())
    out = {"workload": name, "visible": int(vis.sum().item()), "evals_quad": int(n_quad.sum().item())}
    for nm, arr in (("8x4", n_h84), ("4x8", n_h48)):
        a = arr.view(-1, 2)
        out[f"evals_half_{nm}_lockstep"] = int(a.max(1).values.sum().item())
        out[f"evals_half_{nm}_ideal"] = int(a.sum().item() // 2)
        out[f"saving_{nm}_lockstep"] = round(1 - out[f"evals_half_{nm}_lockstep"] / out["evals_quad"], 4)
        out[f"saving_{nm}_ideal"] = round(1 - out[f"evals_half_{nm}_ideal"] / out["evals_quad"], 4)
    out["note"] = "no occlusion / early termination; a wave's two halves walking their own lists finish with the longer one"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
