"""mojosplat_amd -- MI355X (gfx950) backend for the mojosplat 3D-Gaussian-splatting render path.

Public surface = the reference's (mojosplat/render.py, projection.py, binning.py,
rasterization.py, utils.py): ``render_gaussians``, ``Camera``, ``project_gaussians``,
``bin_gaussians_to_tiles``, ``rasterize_gaussians``, each with a ``backend=`` switch that gains
``"hip"``.  Importing this package needs neither a GPU nor the built library.
"""
from .utils import Camera, look_at
from .projection import project_gaussians
from .binning import bin_gaussians_to_tiles
from .rasterization import rasterize_gaussians
from .render import render_gaussians, render_gaussians_batch, TILE_SIZE
from .sh import evaluate_sh


def prepare_scene(*args, **kw):
    """Sort a scene along a Morton curve and compute its block bounds for the multi-GPU band path: scene_order.prepare_scene."""
    from .scene_order import prepare_scene as _prepare
    return _prepare(*args, **kw)


def release_scratch():
    """Give the calling thread's cached render scratch (and the shared lanes') back to the allocator: _fused.release_scratch."""
    from ._fused import release_scratch as _release
    _release()

__all__ = ["Camera", "look_at", "project_gaussians", "bin_gaussians_to_tiles",
           "rasterize_gaussians", "render_gaussians", "render_gaussians_batch", "evaluate_sh", "release_scratch", "prepare_scene", "TILE_SIZE"]
