"""Stage 2: assign projected Gaussians to screen tiles and depth-sort every tile's list.

Drop-in for ``mojosplat.binning.bin_gaussians_to_tiles`` (reference mojosplat/binning.py:8-37):
same arguments, returns ``(sorted_gaussian_indices (M,) i32, tile_ranges (th, tw, 2))`` with
``tile_ranges[y, x] = [start, end)`` into the sorted list (binning.py:88-100, 258-262).

Ordering contract of every backend here (gsplat's): intersections sorted by
(tile id, float bits of depth, Gaussian index); tile box = floor/ceil of (mean -+ radius)/tile,
min inclusive, max exclusive, clamped to the grid; Gaussians with a radius <= 0 are skipped.

Backends: "hip" (gfx950 kernels via ms_isect_tiles_count / ms_isect_tiles_emit, no fallback),
"torch" (vectorised PyTorch, any device), "gsplat"/"mojo" (not shipped -> RuntimeError).
The reference's default is "gsplat" (binning.py:15); that runtime is not available here, so the
default is "hip".
"""
import ctypes
import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _hip
from .projection import _FOREIGN, _foreign


def bin_gaussians_to_tiles(
    means2d: Tensor,  # (N, 2)
    radii: Tensor,    # (N, 2) int
    depths: Tensor,   # (N,)
    img_height: int,
    img_width: int,
    tile_size: int,
    backend: str = "hip",
) -> Tuple[Tensor, Tensor]:
    n_tiles_h = math.ceil(img_height / tile_size)
    n_tiles_w = math.ceil(img_width / tile_size)
    if backend == "torch":
        return bin_gaussians_to_tiles_torch(means2d, radii, depths, tile_size, n_tiles_w, n_tiles_h)
    if backend == "hip":
        return bin_gaussians_to_tiles_hip(means2d, radii, depths, tile_size, n_tiles_w, n_tiles_h)
    if backend in _FOREIGN:
        _foreign(backend)
    raise ValueError(f"Invalid backend: {backend}")


# --------------------------------------------------------------------------- hip backend
_workspaces = {}   # device -> grow-only uint8 scratch (histograms, work lists)
_pinned = {}       # device -> pinned i64[8] for the one size hand-off


def _workspace(dev, nbytes: int) -> Tensor:
    ws = _workspaces.get(dev)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _workspaces[dev] = ws
    return ws


def _pinned_info(dev) -> Tensor:
    p = _pinned.get(dev)
    if p is None:
        p = torch.empty(8, dtype=torch.int64).pin_memory()
        _pinned[dev] = p
    return p


MAX_LDS_TILES = 39932  # (160 KB of LDS - 4 KB of static blocks - 16 B) / 4 B per tile counter (csrc/binning.hip make_plan)


def lds_row_bands(img_height: int, img_width: int, tile_size: int):
    """Tile-row bands such that each band's tile count fits the binning kernels' LDS histogram
    (one band for anything up to ~40.9k tiles, i.e. every BASELINE config incl. 3840x2160)."""
    th, tw = -(-img_height // tile_size), -(-img_width // tile_size)
    if tw > MAX_LDS_TILES:
        raise ValueError(f"{tw} tiles per row exceed the binning kernels' LDS budget")
    rows = max(1, MAX_LDS_TILES // tw)
    return [(r, min(r + rows, th)) for r in range(0, th, rows)]


def bin_gaussians_to_tiles_hip(means2d, radii, depths, tile_size: int, tile_width: int,
                               tile_height: int, row_range: Optional[Tuple[int, int]] = None,
                               return_isect_ids: bool = False, return_tiles_per_gauss: bool = False):
    if row_range is None and tile_width * tile_height > MAX_LDS_TILES:
        # huge tile grids: bin band by band and splice the lists (bands are disjoint tile-id ranges,
        # so concatenation in band order IS the global (tile, depth, id) order)
        parts, off = [], 0
        ranges = None
        for band in lds_row_bands(tile_height * tile_size, tile_width * tile_size, tile_size):
            out = _bin_hip_band(means2d, radii, depths, tile_size, tile_width, tile_height, band,
                                return_isect_ids, return_tiles_per_gauss)
            r = out[1]
            if ranges is None:
                ranges = torch.zeros_like(r)
            ranges[band[0]:band[1]] = r[band[0]:band[1]] + off
            ranges[band[1]:] = off + out[0].numel()       # tiles after this band start where it ends
            off += out[0].numel()
            parts.append(out)
        res = (torch.cat([p[0] for p in parts]), ranges)
        k = 2
        if return_isect_ids:
            res += (torch.cat([p[k] for p in parts]),)
            k += 1
        if return_tiles_per_gauss:
            res += (sum(p[k] for p in parts),)
        return res
    return _bin_hip_band(means2d, radii, depths, tile_size, tile_width, tile_height, row_range,
                         return_isect_ids, return_tiles_per_gauss)


def _bin_hip_band(means2d, radii, depths, tile_size: int, tile_width: int, tile_height: int,
                  row_range: Optional[Tuple[int, int]] = None, return_isect_ids: bool = False,
                  return_tiles_per_gauss: bool = False):
    """gfx950 binning.  One device->host read (M and the over-sized-tile counts) sits between
    the count and the emit call, where gsplat.isect_tiles has its own (binning.py:73-82).

    row_range=(r0, r1) bins only tile rows [r0, r1) (multi-GPU bands); tile ids stay global."""
    _hip.require_cuda(means2d, radii, depths, what="binning input")
    L = _hip.lib()
    dev = means2d.device
    N = means2d.shape[0]
    means2d, depths = _hip.f32c(means2d), _hip.f32c(depths)
    radii = radii.to(torch.int32).contiguous()
    assert means2d.shape == (N, 2) and radii.shape == (N, 2) and depths.shape == (N,)
    r0, r1 = (0, tile_height) if row_range is None else row_range

    ws_bytes = L.ms_isect_workspace_bytes(N, tile_width, tile_height)
    ws = _workspace(dev, ws_bytes)
    tile_ranges = torch.empty((tile_height, tile_width, 2), dtype=torch.int32, device=dev)
    info = torch.empty(8, dtype=torch.int64, device=dev)
    tpg = torch.empty(N, dtype=torch.int32, device=dev) if return_tiles_per_gauss else None
    with torch.cuda.device(dev):
        st = _hip.stream(dev)
        _hip.check(L.ms_isect_tiles_count(
            N, _hip.ptr(means2d), _hip.ptr(radii), tile_size, tile_width, tile_height, r0, r1,
            _hip.ptr(ws), ws.numel(), _hip.ptr(tpg), _hip.ptr(tile_ranges), _hip.ptr(info), st),
            "ms_isect_tiles_count")
        host = _pinned_info(dev)
        host.copy_(info, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        vals = host.tolist()
        M, n_xl = int(vals[0]), int(vals[4])
        if M > 0x7FFFFFFF:
            raise _hip.HipBackendError(f"{M} intersections do not fit int32 indices")
        flatten_ids = torch.empty(M, dtype=torch.int32, device=dev)
        isect_ids = torch.empty(M, dtype=torch.int64, device=dev) if return_isect_ids else None
        if M > 0:
            keys = torch.empty(M, dtype=torch.int64, device=dev)
            tmp = torch.empty(M, dtype=torch.int64, device=dev) if n_xl > 0 else None
            host_info = (ctypes.c_int64 * 8)(*vals)
            _hip.check(L.ms_isect_tiles_emit(
                N, _hip.ptr(means2d), _hip.ptr(radii), _hip.ptr(depths), tile_size, tile_width,
                tile_height, r0, r1, _hip.ptr(ws), ws.numel(), _hip.ptr(tile_ranges), host_info,
                0, 0, 0.0, 0.0,  # gsplat-exact, fully sorted lists (tight binning / lazy sorting are for ms_render_fwd only)
                _hip.ptr(keys), _hip.ptr(tmp), _hip.ptr(flatten_ids), _hip.ptr(isect_ids), st),
                "ms_isect_tiles_emit")
    out = (flatten_ids, tile_ranges)
    if return_isect_ids:
        out += (isect_ids,)
    if return_tiles_per_gauss:
        out += (tpg,)
    return out


def isect_offset_encode_hip(isect_ids: Tensor, tile_width: int, tile_height: int) -> Tensor:
    """gsplat.isect_offset_encode (reference call site binning.py:84): sorted keys -> per-tile
    start offsets (th, tw) i32."""
    _hip.require_cuda(isect_ids)
    L = _hip.lib()
    ids = isect_ids.to(torch.int64).contiguous()
    out = torch.empty((tile_height, tile_width), dtype=torch.int32, device=ids.device)
    with torch.cuda.device(ids.device):
        _hip.check(L.ms_isect_offset_encode(ids.numel(), _hip.ptr(ids), tile_width, tile_height,
                                            _hip.ptr(out), _hip.stream(ids.device)),
                   "ms_isect_offset_encode")
    return out


# ------------------------------------------------------------------------- torch backend
def bin_gaussians_to_tiles_torch(means2d, radii, depths, tile_size: int, tile_width: int,
                                 tile_height: int, row_range: Optional[Tuple[int, int]] = None):
    """Vectorised PyTorch binning with the contract in the module docstring (no Python loop
    over Gaussians, unlike reference binning.py:172-209)."""
    dev = means2d.device
    N = means2d.shape[0]
    T = tile_width * tile_height
    r0, r1 = (0, tile_height) if row_range is None else row_range
    ts = float(tile_size)
    rad = radii.to(torch.float32)
    tx, ty = means2d[:, 0].float() / ts, means2d[:, 1].float() / ts
    trx, try_ = rad[:, 0] / ts, rad[:, 1] / ts

    def cl(v, hi):
        return torch.nan_to_num(v, nan=0.0).clamp(0, hi).to(torch.int64)

    x0, x1 = cl(torch.floor(tx - trx), tile_width), cl(torch.ceil(tx + trx), tile_width)
    y0, y1 = cl(torch.floor(ty - try_), tile_height), cl(torch.ceil(ty + try_), tile_height)
    y0, y1 = y0.clamp(min=r0), y1.clamp(max=r1)
    y1 = torch.maximum(y1, y0)
    visible = (radii[:, 0] > 0) & (radii[:, 1] > 0)
    w = x1 - x0
    n = torch.where(visible, w * (y1 - y0), torch.zeros_like(w))
    M = int(n.sum())
    gid = torch.repeat_interleave(torch.arange(N, device=dev), n)
    first = torch.cumsum(n, 0) - n
    off = torch.arange(M, device=dev) - first[gid]
    wg = w[gid].clamp(min=1)
    tile = (y0[gid] + off // wg) * tile_width + (x0[gid] + off % wg)
    dbits = depths.float().contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    key = (tile << 32) | dbits[gid]
    order = torch.argsort(key, stable=True)
    flatten_ids = gid[order].to(torch.int32)
    bounds = torch.searchsorted((key[order] >> 32).contiguous(),
                                torch.arange(T + 1, device=dev)).to(torch.int32)
    tile_ranges = torch.stack([bounds[:-1], bounds[1:]], dim=-1).view(tile_height, tile_width, 2)
    return flatten_ids, tile_ranges
