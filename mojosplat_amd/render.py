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
# The frame does not depend on the grid the Gaussians are binned on: a pixel blends the same Gaussians
# in the same order whatever the bins, and the rasteriser works in 16x16 blocks inside any tile
# (bit-identical for every mode below at every BASELINE config; tests/test_hip_fused.py).  What the
# grid moves is cost, by up to 2x either way:
#   16  the library's default for 16-px tiles: 32-px bins whose sorted lists are cut into per-block
#       lists (a "split" frame) -- best while Gaussians are small against a tile (config 3: 0.26 ms
#       against 0.31 at plain 32 px and 0.62 at 64);
#   32 / 64  plain coarse bins, every block walks its bin's whole list -- best for dense scenes of
#       larger footprints (config 5: 1.18 / 1.08 / 0.77 ms at 16 / 32 / 64; config 4: 0.69 / 0.64 / 0.66).
# No density rule separates these cases reliably, so the first frames of a scene (same device, N and
# image size) are a measurement: each mode renders one warm-up frame and _TIMED_FRAMES timed ones (a
# pair of events on the launch stream around a synchronised call; the fastest counts), the fastest
# mode is kept, and the race is run again every _REPROBE_EVERY frames, or as soon as a frame's
# intersection count has moved by a quarter, because scenes drift.  64 px is only tried when 32 px did
# not already lose clearly.
_BIN_CHOICE = {}            # (device, N to ~9 %, W, H) -> _BinTuner
_BIN_MODES = (16, 32, 64)
_REPROBE_EVERY = 1024
_TIMED_FRAMES = 4
_MIN_SETTLED = 64           # frames a verdict stands before a drifting count may start the next race
_SKIP_COARSER = 1.10        # 32 px slower than 16 px by this factor: do not try 64


class _BinTuner:
    def __init__(self):
        self.choice = None
        self.frames = 0
        self._start_race()

    def _start_race(self):
        self.times, self.counts, self.settled = {}, {}, 0
        self.queue = [(m, k > 0) for m in _BIN_MODES for k in range(1 + _TIMED_FRAMES)]   # (mode, timed?)

    def next(self):
        """-> (bin size of this frame, whether to time it)."""
        if not self.queue:
            self.frames += 1
            if self.frames % _REPROBE_EVERY == 0:
                self._start_race()
            else:
                return self.choice, False
        return self.queue[0]

    def done(self, mode, seconds, m=None):
        """The frame `next()` announced has been rendered (seconds: None when it was not timed; m: its
        (intersection count, frame kind)).  -> True when the scene has drifted and a new race starts."""
        if not self.queue:
            # a different scene behind the same (device, N, image size): the count of the chosen mode has
            # moved by more than a quarter since it was first seen -> race again from the next frame on.
            # (`m` = (count, kind of frame): a lane that falls back from lazily sorted 32-px bins to fully
            # sorted 16-px tiles counts different things; like is compared with like.)
            self.settled += 1
            if mode == self.choice and m is not None:
                ref = self.counts.setdefault((mode, m[1]), m[0])
                if ref and abs(m[0] - ref) > 0.25 * ref:
                    if self.settled < _MIN_SETTLED:   # two scenes taking turns behind one key: do not race on
                        self.counts[(mode, m[1])] = m[0]   # every switch; follow the count instead
                        return False
                    self._start_race()
                    return True
            return False
        if self.queue[0][0] != mode:
            return False
        self.queue.pop(0)
        if seconds is not None:
            self.times[mode] = min(seconds, self.times.get(mode, seconds))
        if 16 in self.times and 32 in self.times and self.times[32] > _SKIP_COARSER * self.times[16]:
            self.queue = [q for q in self.queue if q[0] != 64]
        if not self.queue:
            self.choice = min(self.times, key=self.times.get)
        return False


def _tuner(means3d, camera, tile_size):
    if tile_size != TILE_SIZE:
        return None         # an explicit non-default tile size is honoured as given
    # (scene sizes within ~9 % of each other share a race: a stream of frames whose N creeps must not race
    # anew on every frame; the drift detector takes care of real changes)
    n = means3d.shape[0]
    key = (means3d.device, round(math.log2(n) * 8) if n > 0 else -1, camera.W, camera.H)
    t = _BIN_CHOICE.get(key)
    if t is None:
        if len(_BIN_CHOICE) >= 64:
            _BIN_CHOICE.pop(next(iter(_BIN_CHOICE)))
        t = _BIN_CHOICE[key] = _BinTuner()
        # a scene that has grown or shrunk into the next size class starts from its neighbour's verdict
        # (the next scheduled race, or the drift detector, revisits it)
        near = [o for k, o in _BIN_CHOICE.items() if o is not t and o.choice is not None and not o.queue
                and (k[0], k[2], k[3]) == (key[0], key[2], key[3]) and abs(k[1] - key[1]) <= 3]
        if near:
            t.choice, t.times, t.queue = near[-1].choice, dict(near[-1].times), []
    return t


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
        bands = lds_row_bands(camera.H, camera.W, tile_size)
        if len(bands) == 1:
            tuner = _tuner(means3d, camera, tile_size)
            bin_size, timed = (tile_size, False) if tuner is None else tuner.next()
            if timed:
                torch.cuda.synchronize(means3d.device)
                start, end = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                start.record()
            info = {}
            img, m = render_fwd_hip(means3d, scales, quats, opacities, colors, camera, bg, bin_size,
                                    stage_events=evs, info=info)
            if tuner is not None:
                seconds = None
                if timed:
                    end.record()
                    end.synchronize()
                    seconds = start.elapsed_time(end) * 1e-3
                if tuner.done(bin_size, seconds, (m, info["flags"] & 8)):
                    from ._fused import forget_learning
                    forget_learning(means3d.device)   # a new scene: the lanes' sorting modes are re-learnt too
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
