"""Scene order (round 5).  The order in which a caller stores its Gaussians does not change a pixel, and since round 5 it
barely moves a whole frame's time either (the count / scatter kernels deal their positions over the workgroups:
csrc/binning.hip, Deal).  A multi-GPU rank's BAND frame does care: in a spatially coherent order the Gaussians that can
reach its band sit in runs, so the band pre-cull can skip whole blocks of them by their bounding boxes and the count
kernel's gathers through the candidate list coalesce.  `prepare_scene` gives a scene that order once (a Morton curve over
the means) and computes the boxes.

The reference has no counterpart (no multi-GPU path; its wrappers take the tensors as they come: render.py:20-41)."""
from dataclasses import dataclass
from typing import Optional

import torch


def morton_permutation(means3d: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Indices that sort the Gaussians along a 3-D Morton (Z-order) curve of their means, `bits` bits per axis over the
    scene's bounding box; stable (equal codes keep their order)."""
    p = means3d.detach().float()
    lo, hi = p.min(0).values, p.max(0).values
    top = float((1 << bits) - 1)
    q = ((p - lo) / (hi - lo + 1e-9) * top).long().clamp(0, int(top))

    def spread(v):   # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    assert bits <= 10
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)
