"""Seeded synthetic scenes used by bench.py, the tests and the sample renderer.

``randscene_v1`` is the recipe SURVEY.md section 8(d) fixes: the random-Gaussian scene of the
reference's sample renderer (render_sample.py:60-71, 86-102) drawn from a *CPU* generator so the
same bytes are produced on every box, with the log-scale mean as a parameter
(render_sample.py uses -2.0, examples/benchmark_proj.py:88-89 uses -3.0; the BASELINE configs
use -4.0).
"""
from typing import Dict, Tuple

import torch

from .utils import Camera, look_at


def randscene_v1(N: int, W: int, H: int, ell: float = -4.0, seed: int = 42,
                 device: str = "cpu", channels: int = 3) -> Tuple[Dict[str, torch.Tensor], Camera]:
    g = torch.Generator().manual_seed(seed)
    means3d = torch.randn(N, 3, generator=g) * 2.0
    log_scales = ell + 0.3 * torch.randn(N, 3, generator=g)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
    opacities = torch.sigmoid(torch.randn(N, generator=g) + 1.0)
    colors = torch.rand(N, channels, generator=g)
    scene = dict(means3d=means3d, scales=log_scales, quats=quats, opacities=opacities,
                 features=colors)
    scene = {k: v.float().contiguous().to(device) for k, v in scene.items()}

    vm = look_at(torch.tensor([0.0, 1.5, 5.0]), torch.zeros(3), torch.tensor([0.0, 1.0, 0.0]))
    f = 500.0 * W / 1920.0
    cam = Camera(R=vm[:3, :3].contiguous().to(device), T=vm[:3, 3].contiguous().to(device),
                 H=H, W=W, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, near=0.1, far=100.0)
    return scene, cam


BACKGROUND_V1 = (0.1, 0.1, 0.1)
