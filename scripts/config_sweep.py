"""Run every BASELINE config on the GPU once: correctness properties that do not need the oracle at
full size (idempotence, stage-wise == fused, band partition == full frame) + timings.

python scripts/config_sweep.py [cfg2 cfg3 cfg3-bwd cfg4 cfg5 ...]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojosplat_amd as ms  # noqa: E402
from mojosplat_amd.autograd import render_gaussians_trainable  # noqa: E402
from mojosplat_amd.distributed import band_plan  # noqa: E402
from mojosplat_amd.rasterization import rasterize_gaussians_hip  # noqa: E402
from mojosplat_amd.binning import bin_gaussians_to_tiles_hip  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402

CFG = {
    "cfg2": (100_000, 1920, 1080, -4.0, False),
    "cfg3": (1_000_000, 1920, 1080, -4.0, False),
    "cfg2-heavy": (100_000, 1920, 1080, -3.0, False),   # the reference benchmark's scale
    "cfg3-heavy": (1_000_000, 1920, 1080, -3.0, False),
    "cfg4": (6_000_000, 1600, 1063, -4.0, True),
    "cfg5": (5_000_000, 3840, 2160, -4.0, False),
}


def timed(fn, iters=100, warm=16):
    for _ in range(warm):   # (render_gaussians moves to the binning rule's grid after a scene's first frame)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    # --order morton: every scene with its Gaussians sorted along a Morton curve of their means first (round 5: the order
    # must not cost a frame anything -- csrc/binning.hip, Deal / BigQ)
    argv = sys.argv[1:]
    order = "given"
    if "--order" in argv:
        k = argv.index("--order")
        order = argv[k + 1]
        argv = argv[:k] + argv[k + 2:]
    assert order in ("given", "morton")
    names = argv or ["cfg2", "cfg3", "cfg3-bwd", "cfg4", "cfg5", "cfg2-heavy", "cfg3-heavy"]
    dev = torch.device("cuda:0")
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    for name in names:
        bwd = name.endswith("-bwd")
        N, W, H, ell, fp16 = CFG[name[:-4] if bwd else name]
        sc, cam = randscene_v1(N, W, H, ell=ell, device=dev)
        if fp16:
            sc["features"] = sc["features"].half()
        if order == "morton":
            from mojosplat_amd.scene_order import morton_permutation
            perm = morton_permutation(sc["means3d"])
            sc = {k: v[perm].contiguous() for k, v in sc.items()}
        g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])
        out = {"config": name, "order": order, "N": N, "W": W, "H": H}
        if bwd:
            leaves = [t.float().clone().requires_grad_(True) for t in g]
            gen = torch.Generator().manual_seed(43)
            v_img = torch.rand(H, W, 3, generator=gen).to(dev)

            def step():
                for l in leaves:
                    l.grad = None
                img = render_gaussians_trainable(*leaves, cam, background_color=bg)
                (img * v_img).sum().backward()

            out["ms_fwd_bwd"] = round(timed(step, iters=5), 3)
            out["grads_finite"] = all(bool(torch.isfinite(l.grad).all()) for l in leaves)
            out["grad_norms"] = [round(float(l.grad.norm()), 4) for l in leaves]
        else:
            img = ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
            m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
            th, tw = -(-H // 16), -(-W // 16)
            ids, ranges = bin_gaussians_to_tiles_hip(m2, rad, dep, 16, tw, th)
            cnt = (ranges[..., 1] - ranges[..., 0])
            out.update(M=int(ids.numel()), max_per_tile=int(cnt.max()), finite=bool(torch.isfinite(img).all()))
            # properties: run-to-run bit equality; fused == stage-wise; band partition == full frame
            img2 = ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
            bgc = bg.to(sc["features"].dtype)
            st = rasterize_gaussians_hip(m2, con, sc["features"], sc["opacities"], bgc, ranges, ids, cam)
            out["idempotent"] = bool(torch.equal(img, img2))
            out["fused_eq_stagewise"] = bool(torch.equal(img, st))
            # sortedness of every tile's list by (depth bits, id)
            key = (dep.view(torch.int32).to(torch.int64)[ids.long()] << 32) | ids.long()
            tile_of = torch.repeat_interleave(torch.arange(th * tw, device=dev), cnt.flatten().long())
            full = key
            ok = (tile_of[1:] > tile_of[:-1]) | ((tile_of[1:] == tile_of[:-1]) & (key[1:] > key[:-1]))
            out["sorted"] = bool(ok.all()) if ids.numel() > 1 else True
            rows, bands = band_plan(th, 8)
            frame = torch.empty_like(img)
            for (r0, r1) in bands:
                if r1 > r0:
                    bi, br = bin_gaussians_to_tiles_hip(m2, rad, dep, 16, tw, th, row_range=(r0, r1))
                    rasterize_gaussians_hip(m2, con, sc["features"], sc["opacities"], bgc, br, bi, cam,
                                            row_range=(r0, r1), out=frame)
            out["bands8_eq_full"] = bool(torch.equal(frame, img))
            del frame, st, img2, key, tile_of, full
            out["ms_fwd"] = round(timed(lambda: ms.render_gaussians(*g, cam, background_color=bg, backend="hip")), 3)
            from mojosplat_amd import render as R
            out["bin_px"] = R._bin_mode.get(R._bin_key(g[0], cam), 16)    # the binning rule's choice (render.py)
            out["fps"] = round(1e3 / out["ms_fwd"], 1)
            # (SURVEY 8(d)'s byte model of a FULL frame -- every pair written, sorted and offered -- over the frame's time: a
            # lazily sorted, depth-cut frame never touches most of those bytes, so this can exceed the 8 000 GB/s peak; it is
            # not a bandwidth.  Measured traffic per frame: profiles/r04_pmc_fetch_write*.txt)
            out["GBps_of_survey_model"] = round((96 * N + (78 if fp16 else 84) * out["M"] + 12 * th * tw + 12 * H * W)
                                    / (out["ms_fwd"] * 1e-3) / 1e9, 1)
        print(json.dumps(out), flush=True)
        del sc, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
