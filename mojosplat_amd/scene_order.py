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


# ---- prepared scenes ---------------------------------------------------------------------------------------------------
BLOCK_SIZE = 256   # Gaussians per block of a prepared scene (config 5 cut 8 ways: the centre band's candidates come out 1.6x
                   # the exact set at 256, 1.8x at 1024, 1.4x at 64 -- the boxes' own size against the band's 270 px)


@dataclass
class PreparedScene:
    """A scene in a spatially coherent order with its block bounds (`prepare_scene`).  Pass `.arrays` to the render calls:
    the sharded entry points recognise the tensors and hand the bounds to the library (ms_scene).  `perm[i]` = index in the
    caller's arrays of the Gaussian now stored at i (gradients / ids map back through it)."""
    means3d: torch.Tensor
    scales: torch.Tensor
    quats: torch.Tensor
    opacities: torch.Tensor
    features: torch.Tensor
    perm: Optional[torch.Tensor]
    block_bounds: torch.Tensor
    block_size: int

    @property
    def arrays(self):
        return (self.means3d, self.scales, self.quats, self.opacities, self.features)


# id(means3d) -> (weakref to means3d, means version, scales weakref, scales version, block bounds, block size).  Round 6
# (advisor): the entry holds NO strong reference to the scene's arrays -- a PreparedScene the caller has dropped frees its
# Gaussians, and a finalizer on the means takes the entry (and the bounds buffer it keeps) out.
_registry = {}
_registry_lock = __import__("threading").Lock()


def clear_registry():
    with _registry_lock:
        _registry.clear()


def _forget(key):
    with _registry_lock:
        _registry.pop(key, None)


def prepare_scene(means3d, scales, quats, opacities, features, block_size: int = BLOCK_SIZE, reorder: bool = True) -> PreparedScene:
    """Sort the Gaussians along a Morton curve of their means (reorder=False: keep the caller's order, which must then be
    spatially coherent for the bounds to be worth anything -- any order stays CORRECT) and compute the bounds of every
    block of `block_size` of them (ms_scene_prepare).  Once per scene: ~2 ms at 5 M Gaussians.  The scales are the
    LOG-scales render_gaussians takes.  The bounds describe the arrays as they are now: call again after the means or
    scales change (an in-place update is noticed through the tensors' version counters and the bounds are dropped)."""
    import ctypes
    import weakref
    from . import _hip
    _hip.require_cuda(means3d, scales, quats, opacities, features, what="gaussian tensor")
    perm = None
    if reorder:
        perm = morton_permutation(means3d)
        means3d, scales, quats, opacities, features = (t[perm].contiguous() for t in (means3d, scales, quats, opacities, features))
    m, sc = _hip.f32c(means3d), _hip.f32c(scales)
    N = m.shape[0]
    L = _hip.lib()
    nbytes = L.ms_scene_block_bounds_bytes(N, block_size)
    bounds = torch.empty(max(nbytes // 4, 8), dtype=torch.float32, device=m.device)
    with _hip.on_device(m.device):
        _hip.check(L.ms_scene_prepare(N, _hip.ptr(m), _hip.ptr(sc), 1, block_size, _hip.ptr(bounds), _hip.stream(m.device)),
                   "ms_scene_prepare")
    # (the library's buffer: the blocks' bounds, then a 16-byte pre-cull record -- mean, largest linear scale -- per Gaussian;
    # the view below keeps all of it alive and starts where the buffer does)
    nb = -(-N // block_size) if N > 0 else 0
    ps = PreparedScene(m, sc, quats, opacities, features, perm, bounds[:max(nb, 1) * 8].view(-1, 8), block_size)
    with _registry_lock:
        _registry[id(m)] = (weakref.ref(m), m._version, weakref.ref(sc), sc._version, ps.block_bounds, block_size)
    weakref.finalize(m, _forget, id(m))
    return ps


def prepared_bounds(means3d, scales):
    """-> (block_bounds tensor, block_size) if these very tensors -- unmodified since -- belong to a prepared scene, else None."""
    with _registry_lock:
        rec = _registry.get(id(means3d))
    if rec is None:
        return None
    m_ref, m_ver, s_ref, s_ver, bounds, block_size = rec
    if m_ref() is not means3d or s_ref() is not scales or means3d._version != m_ver or scales._version != s_ver:
        return None
    return bounds, block_size
