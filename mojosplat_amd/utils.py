"""Camera description shared by every stage.

Drop-in for the reference's ``mojosplat.utils.Camera`` (reference mojosplat/utils.py:5-31):
same field names, order and defaults, the same derived ``view_matrix`` (4x4 world->camera,
``[R|T; 0 0 0 1]``) and ``Ks`` (3x3 pinhole intrinsics), both living on ``R.device``.
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Camera:
    R: torch.Tensor  # (3, 3) world->camera rotation
    T: torch.Tensor  # (3,)   world->camera translation
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float
    near: float = 0.1
    far: float = 100.0
    view_matrix: Optional[torch.Tensor] = None
    Ks: Optional[torch.Tensor] = None

    def __post_init__(self):
        dev, dt = self.R.device, self.R.dtype
        if self.view_matrix is None:
            vm = torch.zeros(4, 4, device=dev, dtype=dt)
            vm[:3, :3] = self.R
            vm[:3, 3] = self.T
            vm[3, 3] = 1.0
            self.view_matrix = vm
        if self.Ks is None:
            self.Ks = torch.tensor(
                [[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]],
                device=dev, dtype=dt)

    # -- helpers used by the HIP backend (not part of the reference surface) ------------
    def _viewmat_f32(self) -> torch.Tensor:
        """Contiguous fp32 (16,) world->camera matrix on the camera's device, cached so the
        HIP projection kernel can read it straight from HBM (no per-call H2D upload, unlike
        the (1,9) intrinsics tensor the reference rebuilds per call at projection.py:446)."""
        vm = self.view_matrix
        cached = getattr(self, "_vm_cache", None)
        if cached is not None and cached[0] is vm and cached[1] == vm._version:
            return cached[2]
        flat = vm.detach().to(torch.float32).contiguous().view(16)
        self._vm_cache = (vm, vm._version, flat)
        return flat

    def _campos(self):
        """World-space camera centre (x, y, z) as Python floats, from the same view matrix the
        projection uses: -R^T T.  Cached (one D2H read per camera, not per frame)."""
        vm = self.view_matrix
        cached = getattr(self, "_cp_cache", None)
        if cached is not None and cached[0] is vm and cached[1] == vm._version:
            return cached[2]
        m = vm.detach().to(torch.float64).cpu()
        pos = tuple(float(v) for v in (-(m[:3, :3].T @ m[:3, 3])).tolist())
        self._cp_cache = (vm, vm._version, pos)
        return pos


def look_at(eye: torch.Tensor, target: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """World->camera 4x4 in the gsplat convention (+X right, +Y down, +Z forward), the one
    the reference's sample renderer uses (render_sample.py:12-30)."""
    eye, target, up = eye.float(), target.float(), up.float()
    fwd = torch.nn.functional.normalize(target - eye, dim=0)
    right = torch.nn.functional.normalize(torch.linalg.cross(fwd, up), dim=0)
    down = torch.linalg.cross(right, fwd)
    Rt = torch.stack([right, down, fwd], dim=0)
    vm = torch.eye(4, dtype=torch.float32, device=eye.device)
    vm[:3, :3] = Rt
    vm[:3, 3] = -(Rt @ eye)
    return vm


# The package's switches are environment variables read on EVERY frame (a test or a host can flip them at run time);
# os.environ.get costs ~1.2 us (encode, lookup, decode) and a band frame asks four times.  The same live lookup on the bytes
# dictionary underneath (CPython on posix) costs 0.1 us.
_ENV_BYTES = getattr(__import__("os").environ, "_data", None)
if not isinstance(_ENV_BYTES, dict):
    _ENV_BYTES = None


def getenv(name: bytes, default=None):
    """os.environ.get(name.decode(), default) -- `name` as bytes, the value as str."""
    if _ENV_BYTES is not None:
        v = _ENV_BYTES.get(name)
        return default if v is None else v.decode()
    import os
    return os.environ.get(name.decode(), default)
