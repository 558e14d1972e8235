"""Stage 1: EWA projection of 3D Gaussians to the image plane.

Drop-in for the reference dispatcher ``mojosplat.projection.project_gaussians``
(reference mojosplat/projection.py:15-48): same arguments, same 4-tuple
``(means2d (N,2) f32, conics (N,3) f32, depths (N,) f32, radii (N,2) i32)``
(dtypes pinned by reference tests/test_projection_mojo.py:52-67), same
``ValueError("Invalid backend: ...")`` for unknown backends.

Backends:
  "hip"    hand-written gfx950 kernel through the C ABI (ms_project_gaussians_fwd), gsplat
           semantics -- what the reference's "gsplat"/"mojo" backends compute
           (projection.py:357-409 / kernels/projection.mojo:13-257).  No fallback.
  "torch"  pure-PyTorch path with the semantics of the reference's own torch backend
           (projection.py:285-346: no opacity term, culled rows keep their values); runs on
           whatever device the tensors are on, CPU included.
  "gsplat", "mojo"  accepted names, but those third-party runtimes are not part of this
           package: a clear RuntimeError, never a silent alias.
"""
from typing import Tuple

import torch
from torch import Tensor

from . import _hip
from .utils import Camera

EPS2D = 0.3

_FOREIGN = ("gsplat", "mojo")


def _foreign(backend: str):
    raise RuntimeError(
        f"backend='{backend}' is provided by a third-party runtime that this package does not "
        "ship; use backend='hip' (MI355X) or backend='torch'")


def project_gaussians(
    means3d: Tensor,           # (N, 3)
    scales: Tensor,            # (N, 3) log-space
    quats: Tensor,             # (N, 4) w, x, y, z
    opacity_features: Tensor,  # (N,) or (N, 1) activated opacities
    camera: Camera,
    backend: str = "torch",
) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    if backend == "torch":
        return project_gaussians_torch(means3d, scales, quats, opacity_features, camera)
    if backend == "hip":
        return project_gaussians_hip(means3d, scales, quats, opacity_features, camera)
    if backend in _FOREIGN:
        _foreign(backend)
    raise ValueError(f"Invalid backend: {backend}")


# --------------------------------------------------------------------------- hip backend
def project_gaussians_hip(means3d, scales, quats, opacities, camera: Camera,
                          scales_are_log: bool = True, radius_clip: float = 0.0):
    """HIP projection.  Layouts at the boundary are the reference kernel's row-major AoS
    (kernels/projection.mojo:270-281); the exp() on the log-scales the reference applies in
    the wrapper (projection.py:454) is fused into the kernel."""
    _hip.require_cuda(means3d, scales, quats, opacities, what="gaussian tensor")
    L = _hip.lib()
    N = means3d.shape[0]
    dev = means3d.device
    means3d, scales, quats = _hip.f32c(means3d), _hip.f32c(scales), _hip.f32c(quats)
    op = None if opacities is None else _hip.f32c(opacities.reshape(-1))
    assert means3d.shape == (N, 3) and scales.shape == (N, 3) and quats.shape == (N, 4)
    assert op is None or op.shape == (N,)
    vm = camera._viewmat_f32()
    if vm.device != dev:
        vm = vm.to(dev)
    means2d = torch.empty((N, 2), dtype=torch.float32, device=dev)
    conics = torch.empty((N, 3), dtype=torch.float32, device=dev)
    depths = torch.empty((N,), dtype=torch.float32, device=dev)
    radii = torch.empty((N, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(L.ms_project_gaussians_fwd(
            N, _hip.ptr(means3d), _hip.ptr(scales), int(scales_are_log), _hip.ptr(quats),
            _hip.ptr(op), _hip.ptr(vm), camera.fx, camera.fy, camera.cx, camera.cy,
            camera.W, camera.H, EPS2D, camera.near, camera.far, radius_clip,
            _hip.ptr(means2d), _hip.ptr(conics), _hip.ptr(depths), _hip.ptr(radii),
            _hip.stream(dev)), "ms_project_gaussians_fwd")
    return means2d, conics, depths, radii


# ------------------------------------------------------------------------- torch backend
def _rotmat(quats: Tensor) -> Tensor:
    q = torch.nn.functional.normalize(quats, p=2, dim=-1)
    w, x, y, z = q.unbind(-1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=-1).reshape(q.shape[:-1] + (3, 3))


def project_gaussians_torch(means3d, scales, quats, opacity_features, camera: Camera):
    """Batched-matmul restatement of the reference torch backend's rules
    (projection.py:199-283 via :285-346): eps2d 0.3, det clamp 1e-10, radius
    ceil(3.33 sqrt(diag)) with NO opacity term, radius 0 outside (near, far) or off-screen,
    culled rows keep their computed means2d/conics/depths.  Opacities are ignored, as there."""
    dt = means3d.dtype
    W, H = camera.W, camera.H
    V = camera.view_matrix.to(means3d.device, dt)
    Rv, tv = V[:3, :3], V[:3, 3]
    K = camera.Ks.to(means3d.device, dt)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]

    M = _rotmat(quats) * torch.exp(scales)[..., None, :]       # R diag(s)
    cov = M @ M.transpose(-1, -2)                              # (N,3,3)
    mc = means3d @ Rv.T + tv                                   # (N,3)
    cov_c = Rv @ cov @ Rv.T

    x, y, z = mc.unbind(-1)
    tan_fovx, tan_fovy = 0.5 * W / fx, 0.5 * H / fy
    lim_xp, lim_xn = (W - cx) / fx + 0.3 * tan_fovx, cx / fx + 0.3 * tan_fovx
    lim_yp, lim_yn = (H - cy) / fy + 0.3 * tan_fovy, cy / fy + 0.3 * tan_fovy
    tx = z * torch.minimum(torch.maximum(x / z, -lim_xn), lim_xp)
    ty = z * torch.minimum(torch.maximum(y / z, -lim_yn), lim_yp)
    zero = torch.zeros_like(z)
    J = torch.stack([fx / z, zero, -fx * tx / (z * z), zero, fy / z, -fy * ty / (z * z)],
                    dim=-1).reshape(-1, 2, 3)
    cov2d = J @ cov_c @ J.transpose(-1, -2)
    means2d = torch.stack([fx * x / z + cx, fy * y / z + cy], dim=-1)

    a = cov2d[:, 0, 0] + EPS2D
    c = cov2d[:, 1, 1] + EPS2D
    b01, b10 = cov2d[:, 0, 1], cov2d[:, 1, 0]
    det = (a * c - b01 * b10).clamp(min=1e-10)
    conics = torch.stack([c / det, -(b01 + b10) / 2.0 / det, a / det], dim=-1)
    radius = torch.stack([torch.ceil(3.33 * torch.sqrt(a)), torch.ceil(3.33 * torch.sqrt(c))], -1)
    valid = (det > 0) & (z > camera.near) & (z < camera.far)
    radius = torch.where(valid[:, None], radius, torch.zeros_like(radius))
    inside = ((means2d[:, 0] + radius[:, 0] > 0) & (means2d[:, 0] - radius[:, 0] < W)
              & (means2d[:, 1] + radius[:, 1] > 0) & (means2d[:, 1] - radius[:, 1] < H))
    radius = torch.where(inside[:, None], radius, torch.zeros_like(radius))
    return means2d, conics, z, radius.int()
