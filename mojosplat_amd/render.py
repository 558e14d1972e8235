"""Orchestrator: project -> bin/sort -> rasterise.

Drop-in for ``mojosplat.render.render_gaussians`` (reference mojosplat/render.py:11-103): same
signature and checks -- non-CUDA inputs and a background/colour channel mismatch raise
ValueError (render.py:44-46, 57-58), opacities must be (N,) (render.py:60), an empty
intersection list returns a ZEROS image, not the background (render.py:73-76), ``sh_degree``
with 2-D features only slices channels (render.py:82-87).  The reference default backend is
"mojo" (render.py:23); here it is "hip".

Beyond the reference: ``features`` of shape (N, K, 3) together with ``sh_degree`` are
spherical-harmonic coefficients and are evaluated per view (sh.py) -- the TODO at
render.py:83 carried out.
"""
from typing import Optional

import torch

from .binning import bin_gaussians_to_tiles, lds_row_bands
from .projection import project_gaussians
from .rasterization import rasterize_gaussians
from .utils import Camera

TILE_SIZE = 16

# bench.py installs a callable here that returns 4 (already recorded once) torch.cuda.Event
# objects per frame; ms_render_fwd re-records them on the launch stream at: start, after
# projection, after binning, after rasterisation.  None = no events.
_STAGE_HOOK = None


# ---- binning granularity -------------------------------------------------------------------------
# The frame does not depend on the tile size: a pixel blends the same Gaussians in the same order
# whatever grid they were binned on, and the rasteriser works in 16x16 blocks inside any tile
# (measured bit-identical for 16 / 32 / 64 px at every BASELINE config).  What the tile size moves
# is cost: coarser bins mean fewer (Gaussian, tile) pairs to scatter and sort (config 5: 38 M at
# 16 px, 9.9 M at 64 px) but every block then stages a longer, less relevant list.  Dense scenes win
# big from coarse bins (config 5: 1.62 -> 0.74 ms at 64 px, config 4: 0.85 -> 0.62 ms at 32 px), sparse
# ones lose (config 3: 0.29 -> 0.72 ms at 32 px).  So: the first frame of a scene (same N and image
# size) is binned on the caller's tile size; its intersections per tile decide the rest.
_BIN_CHOICE = {}            # (device, N, W, H, tile_size) -> bin size for the following frames
_DENSE_PER_TILE = 800       # (tight) intersections per 16-px tile from which coarse bins pay
_MANY_TILES = 16384         # grids this large have the parallelism to absorb 64-px bins


_REPROBE_EVERY = 256        # a scene drifts (camera moves): one frame in 256 is binned at 16 px again


def _bin_size(means3d, camera, tile_size):
    if tile_size != TILE_SIZE:
        return tile_size    # an explicit non-default tile size is honoured as given
    key = (means3d.device, means3d.shape[0], camera.W, camera.H, tile_size)
    entry = _BIN_CHOICE.get(key)
    if entry is None:
        return tile_size
    entry[1] += 1
    if entry[1] % _REPROBE_EVERY == 0:
        return tile_size
    return entry[0]


def _note_density(means3d, camera, tile_size, bin_size, m):
    if tile_size != TILE_SIZE or bin_size != tile_size:
        return              # only frames binned at the default size are evidence
    tiles = (-(-camera.H // tile_size)) * (-(-camera.W // tile_size))
    choice = tile_size
    if m >= _DENSE_PER_TILE * tiles:
        choice = 64 if tiles >= _MANY_TILES else 32
    key = (means3d.device, means3d.shape[0], camera.W, camera.H, tile_size)
    entry = _BIN_CHOICE.get(key)
    if entry is None:
        _BIN_CHOICE[key] = [choice, 0]
    else:
        entry[0] = choice


@torch.no_grad()
def render_gaussians(
    means3d: torch.Tensor,    # (N, 3) world coordinates
    scales: torch.Tensor,     # (N, 3) log-space scales
    quats: torch.Tensor,      # (N, 4) w, x, y, z
    opacities: torch.Tensor,  # (N,) activated opacities
    features: torch.Tensor,   # (N, C) colours, or (N, K, 3) SH coefficients with sh_degree
    camera: Camera,
    sh_degree: Optional[int] = None,
    background_color: Optional[torch.Tensor] = None,
    tile_size: int = TILE_SIZE,
    backend: str = "hip",
) -> torch.Tensor:
    required = [means3d, scales, quats, opacities, features]
    if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in required):
        raise ValueError("All input gaussian tensors must be CUDA tensors.")

    if features.dim() == 3:
        if sh_degree is None:
            raise ValueError("features of shape (N, K, 3) are SH coefficients: pass sh_degree")
        from .sh import evaluate_sh
        out_dtype = features.dtype
        features = evaluate_sh(means3d, features, camera, sh_degree,
                               backend="hip" if backend == "hip" else "torch").to(out_dtype)
        sh_degree = None

    num_channels = features.shape[-1]
    if background_color is None:
        bg = torch.zeros(num_channels, device=means3d.device, dtype=features.dtype)
    elif not isinstance(background_color, torch.Tensor):
        bg = torch.tensor(background_color, device=means3d.device, dtype=features.dtype)
    else:
        bg = background_color.to(device=means3d.device, dtype=features.dtype)
    if bg.shape[0] != num_channels:
        raise ValueError(f"Background color channels ({bg.shape[0]}) must match gaussian color "
                         f"channels ({num_channels})")
    assert opacities.shape == (means3d.shape[0],)

    if backend == "hip":
        # one library call for the whole frame (ms_render_fwd); same stages, same results as the
        # three calls below, without Python between the kernels
        from ._fused import render_fwd_hip
        colors = features
        if sh_degree is not None and features.shape[-1] > 3:
            colors, bg = features[..., :3], bg[:3]  # the reference's placeholder (render.py:82-87)
        evs = _STAGE_HOOK() if _STAGE_HOOK is not None else None
        bin_size = _bin_size(means3d, camera, tile_size)
        bands = lds_row_bands(camera.H, camera.W, bin_size)
        if len(bands) == 1:
            img, m = render_fwd_hip(means3d, scales, quats, opacities, colors, camera, bg, bin_size,
                                    stage_events=evs)
            _note_density(means3d, camera, tile_size, bin_size, m)
            return img
        # tile grids beyond the binning kernels' LDS budget (> ~40.9k tiles, e.g. 8K x 4K frames) are
        # rendered as consecutive row bands into one framebuffer
        img = torch.empty((camera.H, camera.W, colors.shape[-1]), dtype=torch.float32, device=means3d.device)
        info = {}
        for band in bands:
            render_fwd_hip(means3d, scales, quats, opacities, colors, camera, bg, tile_size,
                           row_range=band, out=img, info=info)
        # zeros when no Gaussian's box touches the grid (band-independent count, render.py:73-76)
        return img if info["on_grid"] > 0 else torch.zeros_like(img)

    means2d, conics, depths, radii = project_gaussians(means3d, scales, quats, opacities, camera,
                                                       backend=backend)
    sorted_ids, tile_ranges = bin_gaussians_to_tiles(means2d, radii, depths, camera.H, camera.W,
                                                     tile_size, backend=backend)
    if sorted_ids.numel() == 0:
        return torch.zeros(camera.H, camera.W, num_channels, device=means3d.device,
                           dtype=features.dtype)

    colors = features
    if sh_degree is not None and features.shape[-1] > 3:
        colors = features[..., :3]  # same placeholder as the reference (render.py:82-87)
        bg = bg[:3]

    return rasterize_gaussians(means2d, conics, colors, opacities, bg, tile_ranges, sorted_ids,
                               camera, tile_size=tile_size, backend=backend)


@torch.no_grad()
def render_gaussians_batch(means3d, scales, quats, opacities, features, cameras, background_color=None,
                           tile_size: int = TILE_SIZE, backend: str = "hip") -> torch.Tensor:
    """Multi-view rendering: the same Gaussians from C cameras -> (C, H, W, channels).

    The reference's kernels carry a camera batch dimension but every wrapper pins C = 1
    (projection.py:431, rasterization.py:175); this is the batched entry point SURVEY.md
    section 8(f) row 4 asks for.  Image i equals ``render_gaussians(..., cameras[i])`` bit for
    bit (including the zeros image for a view with no intersections)."""
    if backend != "hip":
        return torch.stack([render_gaussians(means3d, scales, quats, opacities, features, c,
                                             background_color=background_color, tile_size=tile_size,
                                             backend=backend) for c in cameras])
    required = [means3d, scales, quats, opacities, features]
    if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in required):
        raise ValueError("All input gaussian tensors must be CUDA tensors.")
    C = features.shape[-1]
    if background_color is None:
        bg = torch.zeros(C, device=means3d.device, dtype=features.dtype)
    else:
        bg = torch.as_tensor(background_color, device=means3d.device).to(features.dtype)
    if bg.shape[0] != C:
        raise ValueError(f"Background color channels ({bg.shape[0]}) must match gaussian color "
                         f"channels ({C})")
    assert opacities.shape == (means3d.shape[0],)
    from ._fused import render_batch_hip
    out, _ = render_batch_hip(means3d, scales, quats, opacities, features, list(cameras), bg, tile_size)
    return out
