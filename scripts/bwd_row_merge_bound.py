"""Round 6, review item 2: how many atomic rows could a MERGED backward rasteriser save?

The quad-wave backward (csrc/rasterize_bwdq.hip) adds one 64-byte row of raw sums per (8x8 quad, Gaussian that blended into
it): 2.53 M rows at config 3.  Merging "before the rows leave" can at best merge the quads of ONE workgroup's area -- a 16x16
block (4 quads) or a 32-px bin (16 quads).  This counts, from the projected scene alone and without occlusion (an upper bound
on every count alike), per Gaussian the quads / blocks / bins that hold a pixel centre with alpha >= 1/255:

    rows(quads)  = sum over Gaussians of reached 8x8 quads     (what the kernel does today, before early termination)
    rows(blocks) = ... of reached 16x16 blocks                 (merge a block's four quad-waves)
    rows(bins)   = ... of reached 32x32 bins                   (merge a bin's sixteen)
    rows(min)    = visible Gaussians                           (one row each: not reachable by any workgroup-local merge)

    python scripts/bwd_row_merge_bound.py [--workload cfg3]
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd.scenes import randscene_v1  # noqa: E402

WORKLOADS = {"cfg2": (100_000, 1920, 1080, -4.0), "cfg3": (1_000_000, 1920, 1080, -4.0), "cfg3_heavy": (1_000_000, 1920, 1080, -3.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    a = ap.parse_args()
    N, W, H, ell = WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=dev)
    m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
    op = sc["opacities"].reshape(-1)
    vis = (rad[:, 0] > 0) & (rad[:, 1] > 0)
    idx = vis.nonzero().reshape(-1)
    tot = {"quads": 0, "blocks": 0, "bins": 0}
    R = int(rad[vis].max().item())
    R = min(R, 40)
    big = int(((rad[vis] > R).any(1)).sum().item())
    offs = torch.arange(-R, R + 1, device=dev, dtype=torch.float32)
    for s in range(0, idx.numel(), 20000):
        g = idx[s:s + 20000]
        mx, my = m2[g, 0], m2[g, 1]
        # pixel centres in a (2R+1)^2 window around the mean's pixel
        cx, cy = torch.floor(mx), torch.floor(my)
        X = cx[:, None] + offs[None, :] + 0.5    # (n, K)
        Y = cy[:, None] + offs[None, :] + 0.5
        dx = (mx[:, None] - X)[:, None, :]       # (n, 1, K)
        dy = (my[:, None] - Y)[:, :, None]       # (n, K, 1)
        sig = 0.5 * (con[g, 0, None, None] * dx * dx + con[g, 2, None, None] * dy * dy) + con[g, 1, None, None] * dx * dy
        alpha = torch.clamp(op[g, None, None] * torch.exp(-sig), max=0.999)
        inside = ((X >= 0) & (X < W))[:, None, :] & ((Y >= 0) & (Y < H))[:, :, None]
        hit = (alpha >= 1.0 / 255.0) & (sig >= 0) & inside
        px = torch.floor(X)[:, None, :].expand_as(hit).long()
        py = torch.floor(Y)[:, :, None].expand_as(hit).long()
        gi = torch.arange(g.numel(), device=dev)[:, None, None].expand_as(hit)
        for name, sh in (("quads", 3), ("blocks", 4), ("bins", 5)):
            key = (gi[hit] << 40) | ((py[hit] >> sh) << 20) | (px[hit] >> sh)
            tot[name] += int(torch.unique(key).numel())
    out = {"workload": a.workload, "gaussians": N, "visible": int(vis.sum().item()), "window_px": 2 * R + 1,
           "gaussians_wider_than_window": big,
           "rows_per_quad": tot["quads"], "rows_per_block": tot["blocks"], "rows_per_bin": tot["bins"],
           "block_merge_saves": round(1 - tot["blocks"] / max(tot["quads"], 1), 4),
           "bin_merge_saves": round(1 - tot["bins"] / max(tot["quads"], 1), 4),
           "note": "no occlusion / early termination: upper bounds on all three alike; the kernel's measured rows at config 3: 2.53 M"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
