"""Stage 3: tile rasteriser (front-to-back alpha compositing).

Drop-in for ``mojosplat.rasterization.rasterize_gaussians`` (reference
mojosplat/rasterization.py:13-57): same arguments, returns the (H, W, C) image; unknown
backends raise ``ValueError("Invalid backend: ...")`` (pinned by reference
tests/test_rasterization.py:256-266).

Backends: "hip" -- the gfx950 kernel behind ms_rasterize_to_pixels_3dgs_fwd, which takes the
place of both the Mojo op (rasterization.py:169-183) and gsplat.rasterize_to_pixels
(rasterization.py:109-122); no fallback.  "torch" -- the reference has no torch rasteriser
either (its stub forwards to gsplat, rasterization.py:60-78): NotImplementedError here rather
than a hidden hand-over.  "gsplat"/"mojo" -- not shipped, RuntimeError.
The reference default is "mojo" (rasterization.py:23); here it is "hip".
"""
import torch
from torch import Tensor

from . import _hip
from .projection import _FOREIGN, _foreign
from .utils import Camera


def rasterize_gaussians(
    means2d: Tensor,                  # (N, 2)
    conics: Tensor,                   # (N, 3)
    colors: Tensor,                   # (N, C)
    opacities: Tensor,                # (N,) or (N, 1)
    background_color: Tensor,         # (C,)
    tile_ranges: Tensor,              # (th, tw, 2)
    sorted_gaussian_indices: Tensor,  # (M,)
    camera: Camera,
    tile_size: int = 16,
    backend: str = "hip",
) -> Tensor:
    if backend == "hip":
        return rasterize_gaussians_hip(means2d, conics, colors, opacities, background_color,
                                       tile_ranges, sorted_gaussian_indices, camera, tile_size)
    if backend == "torch":
        raise NotImplementedError(
            "there is no PyTorch rasteriser (the reference's is a stub that forwards to gsplat, "
            "rasterization.py:60-78); use backend='hip'")
    if backend in _FOREIGN:
        _foreign(backend)
    raise ValueError(f"Invalid backend: {backend}")


def _squeeze_batch(t: Tensor, nd: int) -> Tensor:
    """The reference wrappers accept tensors with or without a leading camera dim of 1
    (rasterization.py:140-157); normalise to the un-batched form."""
    return t[0] if t.dim() == nd + 1 and t.shape[0] == 1 else t


def rasterize_gaussians_hip(means2d, conics, colors, opacities, background_color, tile_ranges,
                            sorted_gaussian_indices, camera: Camera, tile_size: int = 16,
                            return_aux: bool = False, row_range=None, out=None):
    """-> image (H, W, C) f32 [, alphas (H, W) f32, last_ids (H, W) i32 when return_aux].

    Colours may be fp32 or fp16 (fp32 accumulation either way); the output dtype is fp32,
    the dtype of means2d, as in the reference wrapper (rasterization.py:167).
    row_range=(r0, r1) renders only tile rows [r0, r1) (multi-GPU bands) into `out`."""
    _hip.require_cuda(means2d, conics, colors, opacities, tile_ranges, sorted_gaussian_indices,
                      what="rasteriser input")
    L = _hip.lib()
    dev = means2d.device
    means2d = _hip.f32c(_squeeze_batch(means2d, 2))
    conics = _hip.f32c(_squeeze_batch(conics, 2))
    colors = _squeeze_batch(colors, 2)
    if colors.dtype == torch.float16:
        cdt = 1
        colors = colors.contiguous()
    else:
        cdt = 0
        colors = _hip.f32c(colors)
    N, C = colors.shape
    op = _hip.f32c(opacities.reshape(-1))
    ranges = _squeeze_batch(tile_ranges, 3).to(torch.int32).contiguous()
    ids = sorted_gaussian_indices.reshape(-1).to(torch.int32).contiguous()
    M = ids.numel()
    bg = None
    if background_color is not None:
        bg = _hip.f32c(background_color.reshape(-1).to(dev))
        if bg.numel() != C:
            raise ValueError(f"Background color channels ({bg.numel()}) must match gaussian "
                             f"color channels ({C})")
    H, W = camera.H, camera.W
    th, tw = -(-H // tile_size), -(-W // tile_size)
    if tuple(ranges.shape) != (th, tw, 2):
        raise ValueError(f"tile_ranges shape {tuple(ranges.shape)} != {(th, tw, 2)}")
    assert means2d.shape == (N, 2) and conics.shape == (N, 3) and op.shape == (N,)
    r0, r1 = (0, th) if row_range is None else row_range
    if out is not None:  # caller-owned framebuffer (may be padded below row H), e.g. a gather buffer
        assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= H
        assert tuple(out.shape[1:]) == (W, C) and out.device == dev
        img = out
    else:
        img = torch.empty((H, W, C), dtype=torch.float32, device=dev)
    alphas = torch.empty((H, W), dtype=torch.float32, device=dev) if return_aux else None
    last = torch.empty((H, W), dtype=torch.int32, device=dev) if return_aux else None
    with torch.cuda.device(dev):
        _hip.check(L.ms_rasterize_to_pixels_3dgs_fwd(
            N, M, _hip.ptr(means2d), _hip.ptr(conics), _hip.ptr(colors), cdt, C, _hip.ptr(op),
            _hip.ptr(bg), W, H, tile_size, r0, r1, _hip.ptr(ranges), _hip.ptr(ids), _hip.ptr(img),
            _hip.ptr(alphas), _hip.ptr(last), _hip.stream(dev)),
            "ms_rasterize_to_pixels_3dgs_fwd")
    if return_aux:
        return img, alphas, last
    return img
