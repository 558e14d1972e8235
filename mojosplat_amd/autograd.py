"""Differentiable wrappers around the HIP stages.

The reference pipeline is inference-only (``@torch.no_grad()`` at mojosplat/render.py:11,
"backward pass" listed as future work at README.md:145).  These ``torch.autograd.Function``s
expose the gfx950 backward kernels so the same three stages can sit inside a training loop:

    means2d, conics, depths, radii = project_gaussians_autograd(...)
    ids, ranges = bin_gaussians_to_tiles(means2d.detach(), radii, depths.detach(), ...)
    image = rasterize_gaussians_autograd(means2d, conics, colors, opacities, bg, ranges, ids, cam)

Binning is index work and carries no gradient.
"""
import ctypes
import os

import torch

from . import _hip
from .binning import bin_gaussians_to_tiles_hip
from .projection import EPS2D, project_gaussians_hip
from .rasterization import rasterize_gaussians_hip
from .utils import Camera


class _ProjectHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3d, scales, quats, opacities, camera):
        means3d, scales, quats = _hip.f32c(means3d), _hip.f32c(scales), _hip.f32c(quats)
        out = project_gaussians_hip(means3d, scales, quats, opacities, camera)
        means2d, conics, depths, radii = out
        ctx.camera = camera
        ctx.save_for_backward(means3d, scales, quats, radii)
        ctx.mark_non_differentiable(radii)
        return means2d, conics, depths, radii

    @staticmethod
    def backward(ctx, v_means2d, v_conics, v_depths, _v_radii):
        means3d, scales, quats, radii = ctx.saved_tensors
        cam = ctx.camera
        L = _hip.lib()
        N, dev = means3d.shape[0], means3d.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        v_means2d = _hip.f32c(v_means2d) if v_means2d is not None else z(N, 2)
        v_conics = _hip.f32c(v_conics) if v_conics is not None else z(N, 3)
        v_depths = _hip.f32c(v_depths) if v_depths is not None else None
        v_means3d = torch.empty((N, 3), dtype=torch.float32, device=dev)
        v_scales = torch.empty((N, 3), dtype=torch.float32, device=dev)
        v_quats = torch.empty((N, 4), dtype=torch.float32, device=dev)
        vm = cam._viewmat_f32().to(dev)
        with torch.cuda.device(dev):
            _hip.check(L.ms_project_gaussians_bwd(
                N, _hip.ptr(means3d), _hip.ptr(scales), 1, _hip.ptr(quats), _hip.ptr(vm), cam.fx,
                cam.fy, cam.cx, cam.cy, cam.W, cam.H, EPS2D, _hip.ptr(radii), _hip.ptr(v_means2d),
                _hip.ptr(v_conics), _hip.ptr(v_depths), _hip.ptr(v_means3d), _hip.ptr(v_scales),
                _hip.ptr(v_quats), _hip.stream(dev)), "ms_project_gaussians_bwd")
        return v_means3d, v_scales, v_quats, None, None


class _RasterizeHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, background, tile_ranges, flatten_ids,
                camera, tile_size):
        means2d, conics = _hip.f32c(means2d), _hip.f32c(conics)
        colors, opacities = _hip.f32c(colors), _hip.f32c(opacities.reshape(-1))
        bg = None if background is None else _hip.f32c(background.reshape(-1))
        ranges = tile_ranges.to(torch.int32).contiguous()
        ids = flatten_ids.reshape(-1).to(torch.int32).contiguous()
        img, alphas, last = rasterize_gaussians_hip(means2d, conics, colors, opacities, bg, ranges,
                                                    ids, camera, tile_size, return_aux=True)
        ctx.camera, ctx.tile_size = camera, tile_size
        ctx.save_for_backward(means2d, conics, colors, opacities, bg, ranges, ids, alphas, last)
        return img

    @staticmethod
    def backward(ctx, v_img):
        means2d, conics, colors, opacities, bg, ranges, ids, alphas, last = ctx.saved_tensors
        cam, ts = ctx.camera, ctx.tile_size
        L = _hip.lib()
        N, C = colors.shape
        dev = means2d.device
        v_img = _hip.f32c(v_img)
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # overwritten by the library
        v_means2d, v_conics, v_colors, v_opac = z(N, 2), z(N, 3), z(N, C), z(N)
        ws_bytes = L.ms_rasterize_bwd_workspace_bytes(N, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        with torch.cuda.device(dev):
            _hip.check(L.ms_rasterize_to_pixels_3dgs_bwd(
                N, ids.numel(), _hip.ptr(means2d), _hip.ptr(conics), _hip.ptr(colors), C,
                _hip.ptr(opacities), _hip.ptr(bg), cam.W, cam.H, ts, _hip.ptr(ranges), _hip.ptr(ids),
                _hip.ptr(alphas), _hip.ptr(last), _hip.ptr(v_img), None, _hip.ptr(v_means2d),
                _hip.ptr(v_conics), _hip.ptr(v_colors), _hip.ptr(v_opac), _hip.ptr(ws), ws_bytes, 1,
                _hip.stream(dev)), "ms_rasterize_to_pixels_3dgs_bwd")
        v_bg = None
        if bg is not None and ctx.needs_input_grad[4]:
            v_bg = ((1.0 - alphas)[..., None] * v_img).sum(dim=(0, 1))
        return v_means2d, v_conics, v_colors, v_opac, v_bg, None, None, None, None


# bench.py installs a callable here that returns 3 (already recorded once) torch.cuda.Event objects per backward:
# they are re-recorded on the launch stream before the backward rasteriser, between it and the backward projection,
# and after that.  None = no events.
_BWD_HOOK = None


class _RenderFusedHip(torch.autograd.Function):
    """The whole differentiable frame around ONE forward library call (ms_render_fwd with the
    backward's per-pixel records): fused projection + counting, tight binning, sync-free emit and
    rasterise -- the inference path's forward -- with the frame's scratch kept alive for backward."""

    @staticmethod
    def forward(ctx, means3d, scales, quats, opacities, colors, background, camera, tile_size):
        from ._fused import WHOLE, _Frame
        from . import render as _render   # bench.py's in-situ stage timing hook (None otherwise)
        evs = _render._STAGE_HOOK() if _render._STAGE_HOOK is not None else None
        # three float32 channels: the frame keeps its alphas only and runs the inference frame's binning (lazily sorted
        # fronts on the grid the binning rule picks: the image and the gradients do not depend on it); its backward is
        # the quad-wave rasteriser.  Anything else keeps last_ids and fully sorted lists for the older kernel.
        # (round 5, advisor: the quad-wave kernel works in 16x16 blocks -- ms_render_bwd only takes it when the tile size is a
        # multiple of 16; a frame at tile_size 8 / 24 / ... keeps last_ids and fully sorted lists for the older kernel)
        lean = colors.shape[1] == 3 and colors.dtype == torch.float32 and tile_size % 16 == 0 and \
            os.environ.get("MOJOSPLAT_BWD_QUADS", "1") != "0"
        key = None
        if lean and tile_size == _render.TILE_SIZE and "MOJOSPLAT_TRAIN_BIN_PX" not in os.environ:
            explicit = _render._env_bin_px()
            if explicit is not None:
                tile_size = max(explicit, 16)
            else:
                key = _render._bin_key(means3d, camera)
                with _render._bin_lock:
                    tile_size = _render._bin_mode.get(key, 32)
        frame = _Frame(means3d.detach(), scales.detach(), quats.detach(), opacities.detach(), colors.detach(),
                       camera, background, tile_size, evs, None, None, 0, own=True, own_last=not lean)
        info = {}
        img, M = frame.finish(WHOLE, info)
        if key is not None:
            _render._settle(key, tile_size, _render.bin_rule(tile_size, M, info["on_grid"], camera.W, camera.H, grid_px=tile_size))
        ctx.empty = info["on_grid"] == 0
        ctx.camera, ctx.tile_size = camera, tile_size
        m3, sc, qu, op, col, bg = frame.keep[:6]
        if ctx.empty:
            ctx.save_for_backward(m3, sc, qu, op, col, bg)
            return img
        # the frame's scratch (projected arrays, records, sorted lists) and its size record stay alive for backward:
        # ms_render_bwd reads them where the forward call left them
        import numpy as np
        ctx.scratch = (frame.ws, frame.isect, np.array(frame.st["host_np"], dtype=np.int64, copy=True))
        # (round 5: a lazily sorted differentiable frame reports its clean-up count behind its backward: _fused.own_report)
        h = ctx.scratch[2]
        ctx.own_learn = (frame.st, frame.shape, frame.mode, frame.level, int(h[2]) + int(h[3]) + int(h[4]), frame.grid) if lean else None
        # (the image itself is saved too: the quad-wave backward rasteriser takes what lies behind an entry from it)
        ctx.save_for_backward(m3, sc, qu, op, col, bg, frame.alphas, frame.last, img)
        return img

    @staticmethod
    def backward(ctx, v_img):
        cam, ts = ctx.camera, ctx.tile_size
        if ctx.empty:
            m3, sc, qu, op, col, bg = ctx.saved_tensors
            return (torch.zeros_like(m3), torch.zeros_like(sc), torch.zeros_like(qu), torch.zeros_like(op),
                    torch.zeros_like(col), None if bg is None else torch.zeros_like(bg), None, None)
        m3, sc, qu, op, col, bg, alphas, last, img = ctx.saved_tensors
        ws, isect, host = ctx.scratch
        L = _hip.lib()
        N, C = col.shape
        dev = m3.device
        v_img = _hip.f32c(v_img)
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # overwritten by the library
        v_means3d, v_scales, v_quats, v_opac, v_colors = z(N, 3), z(N, 3), z(N, 4), z(N), z(N, C)
        lean_own = ctx.own_learn is not None and C == 3 and last is None
        # (round 6: a lean differentiable frame's rasteriser zeroed the rows of raw sums inside the frame's own workspace -- bit 15 of
        # its flag word; the backward then needs no scratch of its own and the library skips its 10-us memset)
        rows_in_ws = lean_own and bool(int(host[7]) & 32768)
        bws_bytes = L.ms_render_bwd_workspace_bytes(N, C)
        bws = None if rows_in_ws else torch.empty(bws_bytes, dtype=torch.uint8, device=dev)
        vm = cam._viewmat_f32().to(dev)
        bev = _BWD_HOOK() if _BWD_HOOK is not None else None
        with _hip.on_device(dev):
            if bev:
                bev[0].record()
            if lean_own:
                # (round 5: the two halves of ms_render_bwd, so that a lazily sorted frame's redo launch can leave the frame's
                # clean-up counts in this thread's pinned words on its way -- what the next differentiable frame learns from)
                from . import _fused
                st, shape, mode, level, heavy, grid = ctx.own_learn
                fronts = bool(int(host[7]) & 512)
                mirror = _fused.own_mirror(st) if fronts else None
                th_ = -(-cam.H // ts)
                rows_bytes = L.ms_render_bwd_rows_bytes(N)
                if rows_in_ws:   # (the workspace's last region: include/mojosplat_hip.h, ms_render_bwd_rows)
                    off = L.ms_render_workspace_bytes(N, -(-cam.W // ts), th_) - rows_bytes
                    rows = ws[off:off + rows_bytes].view(torch.float32)
                else:
                    rows = bws[:rows_bytes].view(torch.float32)
                _hip.check(L.ms_render_bwd_rows(N, 3, cam.W, cam.H, ts, 0, th_, _hip.ptr(bg), _hip.ptr(ws), ws.numel(), _hip.ptr(isect),
                                                0 if isect is None else isect.numel(), host.ctypes.data, _hip.ptr(img), _hip.ptr(alphas),
                                                _hip.ptr(v_img), None, _hip.ptr(rows),
                                                None if mirror is None else ctypes.c_void_p(mirror.data_ptr()), _hip.stream(dev)),
                           "ms_render_bwd_rows")
                if rows_in_ws:
                    # (the rows now hold this backward's sums: a second backward through the same graph -- retain_graph=True --
                    # must not add to them: the frame's record no longer vouches for zeroed rows, the next call zeroes its own)
                    host[7] = int(host[7]) & ~32768
                if bev:
                    bev[1].record()
                _hip.check(L.ms_render_bwd_finish(N, _hip.ptr(m3), _hip.ptr(sc), 1, _hip.ptr(qu), _hip.ptr(op), 3, _hip.ptr(vm), cam.fx,
                                                  cam.fy, cam.cx, cam.cy, cam.W, cam.H, EPS2D, _hip.ptr(rows), _hip.ptr(v_means3d),
                                                  _hip.ptr(v_scales), _hip.ptr(v_quats), _hip.ptr(v_opac), _hip.ptr(v_colors),
                                                  _hip.stream(dev)), "ms_render_bwd_finish")
                if bev:
                    bev[2].record()
                if fronts:
                    _fused.own_report(st, shape, mode, level, heavy)
            # one library call: the backward rasteriser (staging from the frame's ready-made records) and the backward
            # projection, on the scratch the forward call left behind
            else:
              _hip.check(L.ms_render_bwd(
                N, _hip.ptr(m3), _hip.ptr(sc), 1, _hip.ptr(qu), _hip.ptr(op), _hip.ptr(col), C, _hip.ptr(vm), cam.fx, cam.fy,
                cam.cx, cam.cy, cam.W, cam.H, EPS2D, ts, _hip.ptr(bg), _hip.ptr(ws), ws.numel(), _hip.ptr(isect),
                0 if isect is None else isect.numel(), host.ctypes.data, _hip.ptr(img) if C == 3 else None, _hip.ptr(alphas),
                _hip.ptr(last), _hip.ptr(v_img),
                None, _hip.ptr(v_means3d), _hip.ptr(v_scales), _hip.ptr(v_quats), _hip.ptr(v_opac), _hip.ptr(v_colors),
                _hip.ptr(bws), bws_bytes, ctypes.c_void_p(bev[1].cuda_event) if bev else None, _hip.stream(dev)),
                "ms_render_bwd")
              if bev:
                bev[2].record()
        v_bg = None
        if bg is not None and ctx.needs_input_grad[5]:
            v_bg = ((1.0 - alphas)[..., None] * v_img).sum(dim=(0, 1))
        return v_means3d, v_scales, v_quats, v_opac, v_colors, v_bg, None, None


def project_gaussians_autograd(means3d, scales, quats, opacities, camera: Camera):
    return _ProjectHip.apply(means3d, scales, quats, opacities, camera)


def rasterize_gaussians_autograd(means2d, conics, colors, opacities, background, tile_ranges,
                                 flatten_ids, camera: Camera, tile_size: int = 16):
    return _RasterizeHip.apply(means2d, conics, colors, opacities, background, tile_ranges,
                               flatten_ids, camera, tile_size)


def render_gaussians_trainable(means3d, scales, quats, opacities, features, camera: Camera,
                               background_color=None, tile_size: int = 16, sh_degree=None,
                               stagewise: bool = False):
    """Differentiable twin of ``render_gaussians(backend="hip")``: grads for means3d, scales
    (log-space), quats, opacities and colours (BASELINE config 3).  With ``sh_degree`` and
    features of shape (N, K, 3) the colours are view-dependent SH (sh.py): gradients then reach
    the coefficients and, through the viewing direction, the means.

    The forward is the inference path's single library call (fused projection + counting, tight
    binning, sync-free emit + rasterise) with the backward's per-pixel records switched on;
    ``stagewise=True`` runs the three per-stage autograd functions instead (gsplat-exact lists,
    one host sync) -- same image, same gradients."""
    _hip.require_cuda(means3d, scales, quats, opacities, features, what="gaussian tensor")
    dev = means3d.device
    if features.dim() == 3:
        if sh_degree is None:
            raise ValueError("features of shape (N, K, 3) are SH coefficients: pass sh_degree")
        from .sh import evaluate_sh_hip
        features = evaluate_sh_hip(means3d, features, camera, sh_degree)
    C = features.shape[-1]
    bg = torch.zeros(C, device=dev) if background_color is None else \
        torch.as_tensor(background_color, dtype=torch.float32, device=dev)
    if stagewise or features.dtype != torch.float32:
        means2d, conics, depths, radii = project_gaussians_autograd(means3d, scales, quats, opacities, camera)
        th, tw = -(-camera.H // tile_size), -(-camera.W // tile_size)
        with torch.no_grad():
            ids, ranges = bin_gaussians_to_tiles_hip(means2d, radii, depths, tile_size, tw, th)
        if ids.numel() == 0:
            # same zeros image as the inference path (reference render.py:73-76), grad-connected
            return (means3d.sum() + features.sum()) * 0 + torch.zeros(camera.H, camera.W, C, device=dev)
        return rasterize_gaussians_autograd(means2d, conics, features, opacities, bg, ranges, ids, camera, tile_size)
    # The differentiable frame's binning grid is free, like the inference frame's (the image and the gradients are
    # sums over the same (pixel, Gaussian) pairs whatever the bins): MOJOSPLAT_TRAIN_BIN_PX = 16 | 32 | 64 picks it for
    # 16-px tiles (measurements; default: the tile size as given).
    bin_px = tile_size
    if tile_size == 16:
        v = os.environ.get("MOJOSPLAT_TRAIN_BIN_PX")
        if v and int(v) in (16, 32, 64):
            bin_px = int(v)
    return _RenderFusedHip.apply(means3d, scales, quats, opacities.reshape(-1), features, bg, camera, bin_px)
