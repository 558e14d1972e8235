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
import math
import os
import threading
import weakref
from typing import Optional

import torch

from .binning import bin_gaussians_to_tiles, lds_row_bands
from .projection import project_gaussians
from .rasterization import rasterize_gaussians
from .utils import Camera, getenv

TILE_SIZE = 16

# bench.py installs a callable here that returns 4 (already recorded once) torch.cuda.Event
# objects per frame; ms_render_fwd re-records them on the launch stream at: start, after
# projection, after binning, after rasterisation.  None = no events.
_STAGE_HOOK = None


# ---- binning granularity -------------------------------------------------------------------------
# The frame does not depend on the grid the Gaussians are binned on: a pixel blends the same Gaussians
# in the same order whatever the bins, and the rasteriser works in 16x16 blocks inside any tile
# (bit-identical for every mode below at every BASELINE config; tests/test_hip_fused.py,
# tests/test_hip_configs.py).  What the grid moves is cost, by up to 2x either way.  Measured with round 2's
# rasteriser (profiles/r02_bin_modes.txt, ms per frame at split / 32 / 64; scripts/scene_scan.py for the rest):
#   16  the library's default for 16-px tiles: 32-px bins whose sorted lists are cut into per-block
#       lists (a "split" frame) -- best only while footprints are small against a BLOCK (1M Gaussians at
#       l = -5, footprint 6 px: 0.22 / 0.32 / 1.05), where a 32-px bin's list would be mostly misses for the
#       8x8 quad that stages it;
#   32  plain 32-px bins, every quad's wave walks its bin's list -- best for footprints of 9-17 px now that
#       staging is three 16-byte gathers per entry (config 3: 0.222 / 0.201 / 0.273; config 2: 0.109 / 0.101 /
#       0.134; config 4: 0.569 / 0.528 / 0.526; round 1's rasteriser, whose staging was seven gathers plus
#       arithmetic, still preferred the split frame there: 0.262 / 0.305);
#   64  plain 64-px bins -- best once footprints span several blocks (100k at l = -3: 0.172 / 0.146 / 0.136;
#       1M at l = -3: 0.286 / 0.238 / 0.192; config 5: 1.09 / 0.91 / 0.68; 1M at l = -3.5, 20 px: 0.245 / 0.206 /
#       0.187): lazily sorted lists are only read ~300-450 entries deep, so binning cost = pairs scattered.
# The choice is a RULE on two statistics every frame's size record already carries -- no timing, no
# synchronisation, same answer for the same sequence of scenes: with M pairs on a g-px grid and n
# Gaussians on the grid, p = M / n pairs per Gaussian give the mean footprint diameter
# d = g (sqrt(p) - 1) px (the same within 2 % whichever grid the frame ran on):
#   d < 9 px: split frame;   9 <= d < 17 px: 32 px;   d >= 17 px: 64 px;
# round 3: from 9 px on also 64 px when the scene is DENSE -- an estimated 1 500 entries or more per 16-px tile
# (n (d / 16 + 1)^2 over the tiles): config 4, see _rule.
# A frame is binned by the rule applied to the PREVIOUS frame of the same (device, N class, image
# size); the first one is a split frame.  Thresholds carry a +-10 % dead band so that a scene sitting
# on one does not flip every frame.  `bin_size=` (or MOJOSPLAT_BIN_PX) overrides the rule;
# `tune_binning()` measures instead, for callers who want that.
_BIN_MODES = (16, 32, 64)
_D_SPLIT, _D_COARSE = 9.0, 17.0
_E_DENSE = 1500.0         # entries per 16-px tile (estimated) from which a whole frame takes 64-px bins whatever its footprints
_DEAD_BAND = 0.10
_MIN_SETTLED = 64
_bin_mode = {}            # (device, N to ~9 %, W, H) -> bin px of the next frame
_bin_left = {}            # same key -> (the grid last switched away from, frames since)
_bin_lock = threading.Lock()


def _rule(d, e, k=1.0, coarse=1.0):
    if d < _D_SPLIT * k:
        return 16
    if e >= _E_DENSE * k and coarse == 1.0:
        # lists several times deeper than a quad's wave ever reads (it stops after ~300-450 entries): the grid moves
        # nothing but the binning cost, which follows the pairs -- and such a scene's frames drop the pairs behind
        # their bins' depth cut-offs (csrc/binning.hip), so the rasteriser is not handed longer lists for it either.
        # Config 4 (6 M Gaussians, 12-px footprints, ~2 700 entries per 16-px tile): 0.291 ms on 32-px bins, 0.265 on 64.
        return 64
    return 32 if d < _D_COARSE * k * coarse else 64


def bin_rule(mode: int, m: int, on_grid: int, W: int, H: int, band: bool = False, grid_px: Optional[int] = None) -> int:
    """Bin size (16 = split frame, 32, 64) the rule above picks after a frame binned at `mode` that
    reported `m` (Gaussian, bin) pairs for `on_grid` Gaussians on a W x H image.
    band=True: the frame was a multi-GPU rank's band of height H whose `on_grid` counts every Gaussian that
    touches the band (csrc/binning.hip, k_band_precull): a Gaussian of diameter d then lies only partly inside,
    on average h / (h + d + g) of its bin rows, and both estimates are corrected for that.
    grid_px: the bin size the pairs were counted on when that is not what `mode` implies."""
    if m <= 0 or on_grid <= 0:
        return mode
    # the grid the pairs were counted on: a frame at mode 16 is a split frame (32-px bins) unless the library
    # kept a thin band on plain 16-px tiles (grid_px says so)
    g = grid_px if grid_px else (32 if mode == 16 else mode)
    p = max(m / on_grid, 1.0)
    bins = math.ceil(W / g) * math.ceil(H / g)
    if p >= 0.8 * bins:
        # footprints as large as the image (every Gaussian in nearly every bin): the estimate below is
        # clamped by the grid and says nothing; such a scene wants the coarsest bins
        return 64
    d = g * (math.sqrt(p) - 1.0)
    inside = 1.0
    if band:
        for _ in range(8):   # fixed point of p (h + d + g) / h = (d / g + 1)^2
            d = g * (math.sqrt(p * (H + d + g) / H) - 1.0)
        inside = H / (H + d + 16.0)
    e = on_grid * inside * (d / 16.0 + 1.0) ** 2 / (math.ceil(W / 16) * math.ceil(H / 16))
    # a rank's band pays the fixed costs of a frame for a fraction of its pixels: 64-px bins only win from ~26 px there
    # (config 3 cut 4 ways, edge bands of 21-22 px footprints: 115 us at 32 px, 138 at 64; config 5 cut 4 or 8 ways,
    # 29 px and up: 64 px wins on every band, by up to 12 %: scripts/band_bench.py)
    coarse = 1.5 if band else 1.0
    lo, hi = _rule(d, e, 1.0 - _DEAD_BAND, coarse), _rule(d, e, 1.0 + _DEAD_BAND, coarse)
    if lo == hi:
        return lo
    # inside a dead band: stay -- unless the current grid is neither of the two the band lies between (a first frame
    # on the default grid): then the coarser of them
    return mode if mode in (lo, hi) else lo


def _settle(key, mode, nxt):
    """Record `nxt` as the grid of the next frame behind `key` -- but never straight back to the grid just left
    (estimates taken on different grids can disagree at the margins: no frame-by-frame flip-flop); after
    _MIN_SETTLED frames it may."""
    with _bin_lock:
        if len(_bin_mode) >= 256 and key not in _bin_mode:
            old = next(iter(_bin_mode))     # the same key leaves both tables
            _bin_mode.pop(old)
            _bin_left.pop(old, None)
        left, age = _bin_left.get(key, (None, 0))
        if nxt != mode and nxt == left and age < _MIN_SETTLED:
            nxt = mode
        _bin_left[key] = (mode, 0) if nxt != mode else (left, age + 1)
        _bin_mode[key] = nxt


_bg_cache = {}


def _background_as(background, device, dtype, fp32_after):
    """The background in the colours' dtype (the reference's order of conversions, render.py:55: with fp16 colours 0.1
    is 0.09998 on every path) -- and, for the HIP path, back in fp32, which is what the library takes.  Two one-element
    kernels per frame otherwise (each a ~4 us bubble in the frame's stream: 2-3 % of a 0.25 ms frame with fp16 colours),
    so the converted tensor is kept per (storage, version, dtypes)."""
    if background.device == device and background.dtype == dtype and (dtype == torch.float32 or not fp32_after):
        return background
    # (keyed by the tensor OBJECT -- a weak reference, checked for identity -- and its version counter: a new tensor at
    # a recycled address, or an in-place update of this one, is converted afresh)
    # (round 4, advisor: `_version` does not move when the storage is written through `.data`, `set_()` or a numpy / DLPack
    # alias -- so views and tensors that share storage with anything else are converted afresh every time, and the key
    # carries the storage address; the converted tensor belongs to the stream that made it: other streams wait for it)
    if background._base is not None or not background.is_contiguous() or background.requires_grad:
        out = background.to(device=device, dtype=dtype)
        return out.to(torch.float32) if fp32_after else out
    key = (id(background), background.data_ptr(), str(device), dtype, fp32_after)
    hit = _bg_cache.get(key)
    if hit is not None and hit[0]() is background and hit[1] == background._version:
        if device.type == "cuda":
            cur = torch.cuda.current_stream(device)
            if cur != hit[3]:
                cur.wait_event(hit[4])
        return hit[2]
    out = background.to(device=device, dtype=dtype)
    if fp32_after:
        out = out.to(torch.float32)
    if len(_bg_cache) >= 64:
        _bg_cache.clear()
    stream = ev = None
    if device.type == "cuda":
        stream = torch.cuda.current_stream(device)
        ev = torch.cuda.Event()
        ev.record(stream)
    _bg_cache[key] = (weakref.ref(background), background._version, out, stream, ev)
    return out


def _bin_key(means3d, camera):
    n = means3d.shape[0]
    return (means3d.device, round(math.log2(n) * 8) if n > 0 else -1, camera.W, camera.H)


def _env_bin_px():
    v = getenv(b"MOJOSPLAT_BIN_PX")
    if not v:
        return None
    if int(v) not in _BIN_MODES:
        raise ValueError(f"MOJOSPLAT_BIN_PX must be one of {_BIN_MODES}")
    return int(v)


def tune_binning(means3d, scales, quats, opacities, features, camera, background_color=None,
                 frames: int = 4):
    """Measure instead of estimate: renders the scene `frames` times per binning mode between stream
    events (synchronises -- not for a latency-critical loop) -> (best bin size, {mode: seconds}).
    Pass the result as `render_gaussians(..., bin_size=best)`."""
    times = {}
    for mode in _BIN_MODES:
        best = None
        for k in range(frames + 1):   # the first frame of a mode warms its scratch up
            start, end = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            torch.cuda.synchronize(means3d.device)
            start.record()
            render_gaussians(means3d, scales, quats, opacities, features, camera,
                             background_color=background_color, bin_size=mode)
            end.record()
            end.synchronize()
            if k:
                t = start.elapsed_time(end) * 1e-3
                best = t if best is None else min(best, t)
        times[mode] = best
    return min(times, key=times.get), times


def _begin_async(means3d, scales, quats, opacities, colors, camera, bg, mode, evs, info, learn):
    """render_gaussians(async_op=True): the frame on a lane stream (ms_render_fwd BEGIN, no host wait), as the sharded
    entry point runs a rank's band (distributed.py) -- streams addressed by handle and ordered with the lanes' persistent
    events; torch's current stream is never switched."""
    from ._fused import _Frame, _frame_lock, _lane_streams
    from .distributed import PendingFrame, _current_stream, _lane_events, _turn
    dev = means3d.device
    cur = _current_stream(dev)
    lanes = _lane_streams(dev)
    # (round 5, advisor: the shared lanes are one-thread-at-a-time under _frame_lock -- the lane pick, the frame's claim on
    # the lane's scratch and begin() happen under it, so two host threads cannot both pass the "lane is free" check)
    with _frame_lock:
        lane = _turn.get(dev, 0)
        _turn[dev] = 1 - lane
        s = lanes[lane]
        # marshal first (copies of non-fp32 / strided inputs are enqueued on the CURRENT stream), then order the lane behind it
        frame = _Frame(means3d, scales, quats, opacities, colors, camera, bg, mode, evs, None, None, 1 + lane, s.cuda_stream)
        ev_in, ev_out = _lane_events(dev, lane)
        ev_in.record(cur)
        s.wait_event(ev_in)
        frame.begin()
    seen = set()
    for t in (means3d, scales, quats, opacities, colors, bg, frame.img) + tuple(x for x in frame.keep[:-1] if x is not None):
        if id(t) not in seen:     # (the caching allocator must not hand their memory out again before the lane is done)
            seen.add(id(t))
            t.record_stream(s)

    def finalize():
        try:
            img, m = frame.finish(info=info)      # size-record check (+ exact redo on the lane stream if it failed)
        finally:
            frame.st["busy"] = False              # (a frame whose finish raised must not hold the lane for ever)
        learn(m)
        ev_out.record(s)
        _current_stream(dev).wait_event(ev_out)
        return img
    # (a PendingFrame dropped unwaited gives the lane back too: its kernels are ordered on the lane's stream before whatever
    # the next frame on that lane enqueues)
    return PendingFrame(finalize=finalize, on_drop=lambda st=frame.st: st.__setitem__("busy", False))


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
    bin_size: Optional[int] = None,
    async_op: bool = False,
) -> torch.Tensor:
    """The reference's entry point (render.py:11-103), same signature.  The reference decorates it `@torch.no_grad()`
    (render.py:11: it has no backward, README.md:145); here the decorator is GONE, as SURVEY.md section 8(f) row 1 asks: a
    call made with autograd enabled on inputs that require gradients (backend "hip", CUDA tensors) is the differentiable
    frame -- autograd.render_gaussians_trainable: one forward library call that keeps its alphas, the quad-wave backward --
    and every other call is the inference frame below, run under no_grad exactly as the reference runs.  (bin_size and
    async_op are inference-frame arguments; a differentiable call ignores bin_size -- the pixels and the gradients do not
    depend on the binning grid -- and refuses async_op.)"""
    # (round 6, advisor: the differentiable frame is for CUDA tensors only -- CPU inputs fall through to the reference's
    # "must be CUDA tensors" ValueError below instead of failing inside the autograd wrapper)
    if backend == "hip" and torch.is_grad_enabled() and all(
            isinstance(t, torch.Tensor) and t.is_cuda for t in (means3d, scales, quats, opacities, features)) and any(
            isinstance(t, torch.Tensor) and t.requires_grad for t in (means3d, scales, quats, opacities, features, background_color)):
        if async_op:
            raise ValueError("async_op: a differentiable frame is rendered by the blocking call")
        from .autograd import render_gaussians_trainable
        feats = features
        if feats.dim() == 2 and sh_degree is not None and feats.shape[-1] > 3:   # the reference's placeholder (render.py:82-87)
            feats = feats[..., :3]
            if background_color is not None:
                background_color = torch.as_tensor(background_color)[:3]
        return render_gaussians_trainable(means3d, scales, quats, opacities, feats, camera, background_color=background_color,
                                          tile_size=tile_size, sh_degree=sh_degree if feats.dim() == 3 else None)
    return _render_gaussians_nograd(means3d, scales, quats, opacities, features, camera, sh_degree, background_color, tile_size,
                                    backend, bin_size, async_op)


@torch.no_grad()
def _render_gaussians_nograd(
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
    bin_size: Optional[int] = None,   # hip backend, tile_size 16: 16 (split frame) | 32 | 64; None = the rule above
    async_op: bool = False,           # hip backend: -> a PendingFrame (see below); use it one frame ahead
) -> torch.Tensor:
    """The reference's entry point (render.py:20-103).  async_op=True (hip backend, round 4) enqueues the frame on one of
    two lane streams -- own scratch each -- without any host wait and returns a `PendingFrame`; `.wait()` completes it on the
    host side (size-record check), makes the CURRENT stream wait for it and hands out the image.  Meant to be used one
    frame ahead -- `nxt = render_gaussians(..., async_op=True); img = cur.wait(); cur = nxt` -- so that the kernels of two
    frames overlap (one frame's latency-bound binning under the other's rasteriser): 0.147 ms a frame against 0.166 for
    back-to-back blocking calls at config 3 (`scripts/two_lane_probe.py`).  At most two frames may be pending; the
    inputs must stay unmodified until `.wait()` returns."""
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
        bg = _background_as(background_color, means3d.device, features.dtype, backend == "hip")
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
        bands = lds_row_bands(camera.H, camera.W, tile_size)
        if len(bands) == 1:
            # an explicit non-default tile size is honoured as given; so is an explicit bin size
            explicit = bin_size if bin_size is not None else _env_bin_px()
            if explicit is not None and explicit not in _BIN_MODES:
                raise ValueError(f"bin_size must be one of {_BIN_MODES}")
            key = None
            if tile_size != TILE_SIZE:
                mode = tile_size
            elif explicit is not None:
                mode = explicit
            else:
                key = _bin_key(means3d, camera)
                with _bin_lock:
                    mode = _bin_mode.get(key, TILE_SIZE)
            info = {}

            def learn(m):
                if key is not None:
                    # (a frame at mode 16 is a split frame -- 32-px bins, flag bit 3 -- unless its lane has fallen back
                    # to fully sorted 16-px tiles: then the pairs were counted on those)
                    grid = (32 if info["flags"] & 8 else 16) if mode == TILE_SIZE else mode
                    _settle(key, mode, bin_rule(mode, m, info["on_grid"], camera.W, camera.H, grid_px=grid))
            if async_op:
                return _begin_async(means3d, scales, quats, opacities, colors, camera, bg, mode, evs, info, learn)
            img, m = render_fwd_hip(means3d, scales, quats, opacities, colors, camera, bg, mode,
                                    stage_events=evs, info=info)
            learn(m)
            return img
        # tile grids beyond the binning kernels' LDS budget (> ~40.9k tiles, e.g. 8K x 4K frames) are
        # rendered as consecutive row bands into one framebuffer
        if async_op:
            raise ValueError("async_op: frames of more than one row band are rendered by the blocking call")
        img = torch.empty((camera.H, camera.W, colors.shape[-1]), dtype=torch.float32, device=means3d.device)
        info = {}
        for band in bands:
            render_fwd_hip(means3d, scales, quats, opacities, colors, camera, bg, tile_size,
                           row_range=band, out=img, info=info)
        # zeros when no Gaussian's box touches the grid (band-independent count, render.py:73-76)
        return img if info["on_grid"] > 0 else torch.zeros_like(img)

    if async_op:
        raise ValueError("async_op needs backend='hip'")
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
                           tile_size: int = TILE_SIZE, backend: str = "hip", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Multi-view rendering: the same Gaussians from C cameras -> (C, H, W, channels).

    The reference's kernels carry a camera batch dimension but every wrapper pins C = 1
    (projection.py:431, rasterization.py:175); this is the batched entry point SURVEY.md
    section 8(f) row 4 asks for.  Image i equals ``render_gaussians(..., cameras[i])`` bit for
    bit (including the zeros image for a view with no intersections)."""
    if backend != "hip":
        imgs = torch.stack([render_gaussians(means3d, scales, quats, opacities, features, c,
                                             background_color=background_color, tile_size=tile_size,
                                             backend=backend) for c in cameras])
        if out is not None:
            out.copy_(imgs)
            return out
        return imgs
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
    cameras = list(cameras)
    # binning granularity: the same rule as render_gaussians (pixels do not depend on it), learnt per scene / image
    # size from the batch's last view and shared with the single-view entry point
    key, mode = None, tile_size
    if tile_size == TILE_SIZE and cameras:
        explicit = _env_bin_px()
        if explicit is not None:
            mode = explicit
        else:
            key = _bin_key(means3d, cameras[0])
            with _bin_lock:
                mode = _bin_mode.get(key, TILE_SIZE)
    out, _, rec = render_batch_hip(means3d, scales, quats, opacities, features, cameras, bg, mode, out=out)
    if key is not None and rec is not None:
        grid = (32 if rec["flags"] & 8 else 16) if mode == TILE_SIZE else mode
        _settle(key, mode, bin_rule(mode, rec["m"], rec["on_grid"], cameras[0].W, cameras[0].H, grid_px=grid))
    return out
