"""Spherical-harmonic colours: the operator behind ``render_gaussians(sh_degree=...)``.

The reference only has a placeholder there -- "TODO: Implement SH evaluation", features sliced
to three channels (mojosplat/render.py:82-87) -- and SURVEY.md section 8(f) row 2 lists the
real thing as the next row of the hot path.  Convention = gsplat's (the reference's parity
target everywhere else): real SH up to degree 4 with 3DGS signs, coefficient k = l*(l+1)+m,
direction = normalise(mean - camera position), colour = max(sum_k b_k(dir) * coeff_k + 0.5, 0).

    colors = evaluate_sh(means3d, sh_coeffs, camera, sh_degree)            # (N, 3)
    image  = render_gaussians(means3d, scales, quats, opac, sh_coeffs, cam, sh_degree=3)

`sh_coeffs` is (N, K, 3) with K >= (sh_degree+1)^2.  backend="hip" runs csrc/sh.hip (forward and
backward, differentiable w.r.t. the coefficients and the means); backend="torch" is a plain
PyTorch restatement for CPU tensors.  2-D features keep the reference's placeholder behaviour.
"""
from typing import Optional

import torch

from . import _hip
from .utils import Camera

MAX_SH_DEGREE = 4


def camera_position(camera: Camera) -> torch.Tensor:
    """World-space camera centre of the world->camera view matrix [R|T]: -R^T T."""
    vm = camera.view_matrix
    return -(vm[:3, :3].transpose(0, 1) @ vm[:3, 3])


def sh_basis_torch(degree: int, dirs: torch.Tensor) -> torch.Tensor:
    """(..., 3) unit directions -> (..., (degree+1)^2) basis values, explicit polynomials."""
    x, y, z = dirs.unbind(-1)
    b = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        b += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        b += [1.092548430592079 * x * y, -1.092548430592079 * y * z,
              0.9461746957575601 * zz - 0.3153915652525201, -1.092548430592079 * x * z,
              0.5462742152960395 * (xx - yy)]
    if degree >= 3:
        b += [-0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * x * y * z,
              0.4570457994644658 * y * (1 - 5 * zz), 0.3731763325901154 * z * (5 * zz - 3),
              0.4570457994644658 * x * (1 - 5 * zz), 1.445305721320277 * z * (xx - yy),
              -0.5900435899266435 * x * (xx - 3 * yy)]
    if degree >= 4:
        b += [2.5033429417967046 * x * y * (xx - yy), -1.7701307697799304 * y * z * (3 * xx - yy),
              0.9461746957575601 * x * y * (7 * zz - 1), -0.6690465435572892 * y * z * (7 * zz - 3),
              0.10578554691520431 * (35 * zz * zz - 30 * zz + 3), -0.6690465435572892 * x * z * (7 * zz - 3),
              0.47308734787878004 * (xx - yy) * (7 * zz - 1), -1.7701307697799304 * x * z * (xx - 3 * yy),
              0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(b, dim=-1)


def _check(means3d, sh_coeffs, sh_degree):
    if not (0 <= sh_degree <= MAX_SH_DEGREE):
        raise ValueError(f"sh_degree must be in [0, {MAX_SH_DEGREE}], got {sh_degree}")
    if sh_coeffs.dim() != 3 or sh_coeffs.shape[0] != means3d.shape[0] or sh_coeffs.shape[2] != 3:
        raise ValueError(f"sh_coeffs must be (N, K, 3), got {tuple(sh_coeffs.shape)}")
    if sh_coeffs.shape[1] < (sh_degree + 1) ** 2:
        raise ValueError(f"sh_degree {sh_degree} needs {(sh_degree + 1) ** 2} coefficients per Gaussian, "
                         f"got {sh_coeffs.shape[1]}")


def evaluate_sh_torch(means3d, sh_coeffs, camera: Camera, sh_degree: int, radii=None, clamp=True):
    _check(means3d, sh_coeffs, sh_degree)
    d = means3d - camera_position(camera).to(means3d)
    d = d / d.norm(dim=-1, keepdim=True)
    ku = (sh_degree + 1) ** 2
    col = (sh_basis_torch(sh_degree, d).unsqueeze(-1) * sh_coeffs[:, :ku].to(d.dtype)).sum(dim=1)
    if clamp:
        col = (col + 0.5).clamp_min(0.0)
    if radii is not None:
        col = col * ((radii[:, 0] > 0) & (radii[:, 1] > 0)).unsqueeze(-1).to(col.dtype)
    return col


class _ShHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3d, sh_coeffs, campos, sh_degree, radii, clamp, half):
        L = _hip.lib()
        means3d, sh_coeffs = _hip.f32c(means3d), _hip.f32c(sh_coeffs)
        N, K = sh_coeffs.shape[0], sh_coeffs.shape[1]
        dev = means3d.device
        colors = torch.empty((N, 3), dtype=torch.float16 if half else torch.float32, device=dev)
        if radii is not None:
            radii = radii.to(torch.int32).contiguous()
        with _hip.on_device(dev):
            _hip.check(L.ms_spherical_harmonics_fwd(
                N, K, sh_degree, _hip.ptr(means3d), campos[0], campos[1], campos[2], _hip.ptr(sh_coeffs),
                _hip.ptr(radii), int(clamp), 1 if half else 0, _hip.ptr(colors), _hip.stream(dev)),
                "ms_spherical_harmonics_fwd")
        ctx.cfg = (campos, sh_degree, clamp, half)
        ctx.save_for_backward(means3d, sh_coeffs, radii, colors)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        means3d, sh_coeffs, radii, colors = ctx.saved_tensors
        campos, sh_degree, clamp, half = ctx.cfg
        L = _hip.lib()
        N, K = sh_coeffs.shape[0], sh_coeffs.shape[1]
        dev = means3d.device
        need_c, need_m = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        v_coeffs = torch.empty_like(sh_coeffs) if need_c else None
        v_means = torch.empty_like(means3d) if need_m else None
        if need_c or need_m:
            with _hip.on_device(dev):
                _hip.check(L.ms_spherical_harmonics_bwd(
                    N, K, sh_degree, _hip.ptr(means3d), campos[0], campos[1], campos[2], _hip.ptr(sh_coeffs),
                    _hip.ptr(radii), int(clamp), _hip.ptr(colors.float() if half else colors),
                    _hip.ptr(_hip.f32c(v_colors)), _hip.ptr(v_coeffs), _hip.ptr(v_means), _hip.stream(dev)),
                    "ms_spherical_harmonics_bwd")
        return v_means, v_coeffs, None, None, None, None, None


def evaluate_sh_hip(means3d, sh_coeffs, camera: Camera, sh_degree: int, radii=None, clamp=True, half=False):
    _check(means3d, sh_coeffs, sh_degree)
    _hip.require_cuda(means3d, sh_coeffs, radii)
    campos = camera._campos()
    return _ShHip.apply(means3d, sh_coeffs, campos, int(sh_degree), radii, bool(clamp), bool(half))


def evaluate_sh(means3d: torch.Tensor, sh_coeffs: torch.Tensor, camera: Camera, sh_degree: int,
                radii: Optional[torch.Tensor] = None, clamp: bool = True, backend: str = "hip") -> torch.Tensor:
    """(N, 3) colours of the Gaussians as seen from `camera`.  radii (N, 2) int, optional: rows with
    a zero radius (culled by the projection) are skipped and come back as 0."""
    if backend == "hip":
        return evaluate_sh_hip(means3d, sh_coeffs, camera, sh_degree, radii, clamp)
    if backend == "torch":
        return evaluate_sh_torch(means3d, sh_coeffs, camera, sh_degree, radii, clamp)
    if backend in ("gsplat", "mojo"):
        raise RuntimeError(f"backend='{backend}' is not available in this build (use 'hip' or 'torch')")
    raise ValueError(f"Invalid backend: {backend}")
