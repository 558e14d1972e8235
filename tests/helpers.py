"""Shared scene builders for the test-suite (CPU generators -> identical bytes everywhere).

Scenes restate the fixtures of the reference's tests: tests/test_projection_mojo.py:16-46,
tests/test_rasterization.py:18-36.
"""
import glob
import os

import numpy as np
import torch

from mojosplat_amd import Camera

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(path):
    d = np.load(path)
    H, W = (int(v) for v in d["HW"])
    fx, fy, cx, cy = (float(v) for v in d["intr"])
    near, far = (float(v) for v in d["nearfar"])
    return d, dict(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, near=near, far=far)


def camera_from_golden(d, cam, device="cpu"):
    vm = torch.from_numpy(d["viewmat"]).to(device)
    return Camera(R=vm[:3, :3].contiguous(), T=vm[:3, 3].contiguous(), H=cam["H"], W=cam["W"],
                  fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], near=cam["near"],
                  far=cam["far"])


def simple_camera(device="cpu", H=64, W=64, f=100.0, T=(0.0, 0.0, 0.0)):
    return Camera(R=torch.eye(3, device=device), T=torch.tensor(T, dtype=torch.float32, device=device),
                  H=H, W=W, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, near=0.1, far=100.0)


def proj_scene(N, seed=42, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    means3d = torch.randn(N, 3, generator=g) * 2.0
    means3d[:, 2] = means3d[:, 2].abs() + 1.0
    scales = torch.log(torch.rand(N, 3, generator=g) * 0.3 + 0.05)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), p=2, dim=-1)
    opac = torch.sigmoid(torch.randn(N, generator=g))
    return [t.to(device) for t in (means3d, scales, quats, opac)]


def raster_scene(N, seed=0, device="cpu", depth_range=(1.5, 5.0), scale_log=-2.0,
                 opacity_range=(0.5, 0.95), channels=3):
    g = torch.Generator().manual_seed(seed)
    means3d = torch.randn(N, 3, generator=g)
    means3d[:, 2] = torch.rand(N, generator=g) * (depth_range[1] - depth_range[0]) + depth_range[0]
    ls = torch.ones(N, 3) * scale_log + torch.randn(N, 3, generator=g) * 0.1
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
    opac = torch.rand(N, generator=g) * (opacity_range[1] - opacity_range[0]) + opacity_range[0]
    colors = torch.rand(N, channels, generator=g)
    return [t.to(device) for t in (means3d, ls, quats, opac, colors)]


def np_(t):
    return t.detach().cpu().numpy()


def oracle_project(oracle, means3d, scales, quats, opac, cam, **kw):
    vm = np_(cam.view_matrix)
    return oracle.project_fwd(np_(means3d), np_(scales), np_(quats), None if opac is None else np_(opac),
                              vm, cam.fx, cam.fy, cam.cx, cam.cy, cam.W, cam.H, near=cam.near,
                              far=cam.far, **kw)


# ------------------------------------------------------------------ the raster parity bar
# North star: <= 1e-4 abs per pixel fp32 against the reference rasteriser
# (reference tests/test_rasterization.py:110: atol = rtol = 1e-4).  Two fp32 implementations of the
# compositor agree to rounding everywhere EXCEPT where one of its three data-dependent branches
# (alpha >= 1/255, T(1-alpha) <= 1e-4, sigma < 0) sits within rounding of its threshold: there one of
# them blends a Gaussian the other skips and the pixel moves by up to ~1/255.  The oracle reports, per
# pixel, how close any branch of its walk came to its threshold (`margin`, see orc_rasterize_fwd_rows);
# a pixel may exceed `atol` only if that margin is below `eps` -- zero unexplained pixels is asserted,
# and even explained ones are capped.  The counts go to stdout and gpurun_out/parity_counts.jsonl.
PARITY_LOG = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out", "parity_counts.jsonl")


def check_image_strict(img, ref, margin, *, tag, atol=1e-4, eps=1e-5, flip_cap=1e-2, f64=None):
    """img, ref (H,W,C); margin (H,W) from the oracle.  f64: optional float64 oracle frame (its
    disagreement with the fp32 oracle is recorded beside the counts).  -> the record (dict)."""
    import json
    img = np_(img) if torch.is_tensor(img) else np.asarray(img)
    assert img.shape == ref.shape and np.isfinite(img).all()
    diff = np.abs(img.astype(np.float64) - ref.astype(np.float64)).max(axis=-1)
    bad = diff > atol
    sens = margin < eps
    rec = dict(tag=tag, pixels=int(diff.size), atol=atol, margin_eps=eps, beyond_atol=int(bad.sum()),
               explained_by_branch_margin=int((bad & sens).sum()), unexplained=int((bad & ~sens).sum()),
               branch_sensitive_pixels=int(sens.sum()), max_abs=float(diff.max()),
               max_abs_where_no_branch_is_close=float(diff[~sens].max()) if (~sens).any() else 0.0)
    if f64 is not None:
        d64 = np.abs(f64 - ref.astype(np.float64)).max(axis=-1)
        rec["oracle_f32_vs_f64_beyond_atol"] = int((d64 > atol).sum())
        rec["oracle_f32_vs_f64_beyond_atol_unexplained"] = int(((d64 > atol) & ~sens).sum())
    line = json.dumps(rec)
    print("PARITY", line)
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert rec["unexplained"] == 0, f"{tag}: {rec['unexplained']} px beyond {atol} with no branch near its threshold: {rec}"
    assert rec["max_abs"] <= flip_cap, f"{tag}: max abs diff {rec['max_abs']:.3g}"
    return rec
