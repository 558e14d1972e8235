"""Multi-GPU rendering of ONE frame: image tile-row bands over the ranks of a node, one RCCL
all-gather of the framebuffer over xGMI (BASELINE config 5).

The reference has no distributed path at all (no torch.distributed / NCCL call site anywhere),
so this is new design, MI355X-first:
  * every rank holds the full Gaussian set and projects it redundantly -- 76 B/Gaussian of HBM
    streaming is cheaper than moving 32 B/Gaussian of projected data across xGMI;
  * rank r bins, sorts and rasterises only tile rows [r*rows, (r+1)*rows): its pixels are one
    contiguous slab of the HWC framebuffer, so the exchange is a single equal-sized
    all_gather_into_tensor whose input slab aliases its slot of the output (in-place form,
    no staging copy).  Slabs are padded to `rows` tile rows; rows below H are never read;
  * the "no intersections anywhere -> zeros image" rule of the reference (render.py:73-76)
    needs the global intersection count: a 1-float-per-rank all-gather issued right after the
    binning size hand-off, so its latency hides under the rasteriser.

One process per GPU; backend "nccl" is RCCL on ROCm.  `stages` makes the orchestration testable
on CPU with gloo (tests inject CPU stage functions); the default stages are the HIP kernels.
"""
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .utils import Camera


def band_plan(tile_rows: int, world: int) -> Tuple[int, list]:
    """rows per rank (equal, padded) and the [r0, r1) tile-row band of every rank."""
    rows = -(-tile_rows // world)
    bands = [(min(r * rows, tile_rows), min((r + 1) * rows, tile_rows)) for r in range(world)]
    return rows, bands


@dataclass
class Stages:
    project: Callable   # (means3d, scales, quats, opacities, camera) -> means2d, conics, depths, radii
    bin: Callable       # (means2d, radii, depths, tile_size, tw, th, row_range) -> ids, tile_ranges
    raster: Callable    # (means2d, conics, colors, opacities, bg, ranges, ids, camera, tile_size, row_range, out)


def hip_stages() -> Stages:
    from .binning import bin_gaussians_to_tiles_hip
    from .projection import project_gaussians_hip
    from .rasterization import rasterize_gaussians_hip

    def raster(m2, con, col, op, bg, ranges, ids, cam, ts, row_range, out):
        return rasterize_gaussians_hip(m2, con, col, op, bg, ranges, ids, cam, ts, row_range=row_range, out=out)

    return Stages(project=project_gaussians_hip,
                  bin=lambda m2, rad, dep, ts, tw, th, rr: bin_gaussians_to_tiles_hip(m2, rad, dep, ts, tw, th, row_range=rr),
                  raster=raster)


_frames = {}  # (device, H_pad, W, C) -> gather buffer, reused across frames


@torch.no_grad()
def render_gaussians_sharded(means3d, scales, quats, opacities, features, camera: Camera,
                             background_color: Optional[torch.Tensor] = None, tile_size: int = 16,
                             group=None, stages: Optional[Stages] = None) -> torch.Tensor:
    """Every rank returns the full (H, W, C) image.  Inputs must be identical on all ranks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    st = stages or hip_stages()
    dev = means3d.device
    C = features.shape[-1]
    H, W = camera.H, camera.W
    th, tw = -(-H // tile_size), -(-W // tile_size)
    rows, bands = band_plan(th, world)
    r0, r1 = bands[rank]
    bg = torch.zeros(C, device=dev, dtype=torch.float32) if background_color is None else \
        torch.as_tensor(background_color, dtype=torch.float32, device=dev)
    if bg.shape[0] != C:
        raise ValueError(f"Background color channels ({bg.shape[0]}) must match gaussian color channels ({C})")

    means2d, conics, depths, radii = st.project(means3d, scales, quats, opacities, camera)
    ids, ranges = st.bin(means2d, radii, depths, tile_size, tw, th, (r0, r1))

    flags = None
    flag_work = None
    if world > 1:
        flags = torch.empty(world, dtype=torch.float32, device=dev)
        mine = torch.full((1,), float(ids.numel() > 0), dtype=torch.float32, device=dev)
        flag_work = dist.all_gather_into_tensor(flags, mine, group=group, async_op=True)

    slab = rows * tile_size
    H_pad = max(world * slab, H)
    if world == 1:
        full = torch.empty((H_pad, W, C), dtype=torch.float32, device=dev)  # handed to the caller
    else:
        key = (dev, H_pad, W, C)
        full = _frames.get(key)
        if full is None:
            full = torch.zeros((H_pad, W, C), dtype=torch.float32, device=dev)
            _frames[key] = full
    if r1 > r0:
        st.raster(means2d, conics, features, opacities, bg, ranges, ids, camera, tile_size, (r0, r1), full)
    if world > 1:
        mine_slab = full[rank * slab:(rank + 1) * slab]
        dist.all_gather_into_tensor(full[:world * slab], mine_slab, group=group)
        flag_work.wait()
        any_isect = flags.sum() > 0
        img = full[:H]
        # zeros (not background) when no rank found an intersection; stays on device, no sync
        return torch.where(any_isect, img, torch.zeros_like(img))
    if ids.numel() == 0:
        return torch.zeros(H, W, C, device=dev, dtype=torch.float32)
    return full[:H]
