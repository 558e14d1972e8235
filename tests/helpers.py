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


# ------------------------------------------------------------------ gradient bars (round 5)
# Two bars per tensor.  (1) max norm: |g - g_ref| <= rel * max|g_ref| -- what rounds 1-4 asserted; it is blind to the error
# of the small-gradient majority.  (2) per element: every element with |g_ref| >= floor * max|g_ref| must agree to
# elem_rel RELATIVE to itself (and the 99.9th percentile of those errors to elem_p999).  The backward rasteriser's
# two-term bf16 tile (2^-17 per term) and its approximate reciprocal are what bar (2) watches.
# Where the bars sit (measured, profiles/r05_grad_stats.txt): against float64 autograd the worst element of any test is
# 3.4e-4 off -> 2e-3 on EVERY element.  Fused against the per-stage functions BOTH sides are fp32 with different orders of
# summation (the per-stage kernel walks back to front and recovers T by division): regular scenes up to 3e-3 on a single
# element of ~7 000 (a sum that nearly cancels), 99.9 % within 5e-4 -> 5e-3 on every element, 1e-3 on the 99.9th
# percentile; the adversarial stacks (thousands of faint entries at one depth behind every pixel) 1.6e-2 / 2.3e-3 ->
# 5e-2 / 5e-3.
def grad_stats(got, ref, floor=1e-3):
    """-> dict(scale, max_norm_err, checked, elem_rel_max, elem_rel_p999, worst_index): got / ref arrays or tensors."""
    got = (np_(got) if torch.is_tensor(got) else np.asarray(got)).astype(np.float64).reshape(-1)
    ref = (np_(ref) if torch.is_tensor(ref) else np.asarray(ref)).astype(np.float64).reshape(-1)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    err = np.abs(got - ref)
    out = dict(scale=scale, max_norm_err=float(err.max() / scale) if scale > 0 else 0.0, checked=0, elem_rel_max=0.0,
               elem_rel_p999=0.0, worst_index=-1)
    if scale > 0:
        sel = np.abs(ref) >= floor * scale
        if sel.any():
            rel = err[sel] / np.abs(ref[sel])
            out.update(checked=int(sel.sum()), elem_rel_max=float(rel.max()), elem_rel_p999=float(np.quantile(rel, 0.999)),
                       worst_index=int(np.flatnonzero(sel)[int(rel.argmax())]))
    return out


GRAD_LOG = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out", "grad_stats.jsonl")


def assert_grad_close(name, got, ref, rel=2e-3, elem_rel=None, floor=1e-3, elem_p999=None):
    """GRAD_BARS_SOFT=1 in the environment records the per-element statistics (gpurun_out/grad_stats.jsonl) without
    asserting them -- for calibrating the bars; the max-norm bar is always asserted."""
    import inspect
    import json
    st = grad_stats(got, ref, floor)
    try:
        os.makedirs(os.path.dirname(GRAD_LOG), exist_ok=True)
        with open(GRAD_LOG, "a") as f:
            f.write(json.dumps(dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], name=name, rel=rel,
                                    elem_rel=elem_rel, **st)) + "\n")
    except OSError:
        pass
    assert st["max_norm_err"] * st["scale"] <= rel * st["scale"] + 1e-6, f"{name}: max-norm bar {rel}: {st}"
    if os.environ.get("GRAD_BARS_SOFT") != "1":
        if elem_rel is not None:
            assert st["elem_rel_max"] <= elem_rel, f"{name}: per-element bar {elem_rel} (|g_ref| >= {floor} max): {st}"
        if elem_p999 is not None:
            assert st["elem_rel_p999"] <= elem_p999, f"{name}: 99.9th percentile of the per-element error > {elem_p999}: {st}"
    return st
