"""One-call forward path used by ``render_gaussians(backend="hip")``: ms_render_fwd.

The three stage calls of the reference orchestrator (mojosplat/render.py:63-101) collapse into
one library call; the projected tensors, tile ranges and the sorted intersection list stay in
caller-owned scratch that is cached per device and reused across frames (288 GB of HBM: the
scratch for 1M Gaussians at 1080p is ~100 MB and is never freed or re-sized unless a frame
needs more).  The per-stage entry points stay available for callers that want the
intermediates (project_gaussians / bin_gaussians_to_tiles / rasterize_gaussians).
"""
import ctypes

import torch

from . import _hip
from .projection import EPS2D

_state = {}  # (device, lane) -> dict(ws, isect, host, ev); lane 0 = the plain single-frame path


def _dev_state(dev, lane=0):
    st = _state.get((dev, lane))
    if st is None:
        ev = torch.cuda.Event()
        with torch.cuda.device(dev):
            ev.record()  # materialises the hipEvent_t the library re-records for its size hand-off
        st = dict(ws=None, isect=None, host=torch.zeros(8, dtype=torch.int64).pin_memory(), ev=ev)
        _state[(dev, lane)] = st
    return st


def _grow(st, key, nbytes, dev, slack=1.0):
    buf = st[key]
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * slack) + 256, dtype=torch.uint8, device=dev)
        st[key] = buf
    return buf


def render_fwd_hip(means3d, scales, quats, opacities, colors, camera, background, tile_size,
                   stage_events=None, row_range=None, out=None, lane=0):
    """-> (image (H,W,C) f32, M).  `background` may be None.  stage_events: None or a list of 4
    torch.cuda.Event that have been recorded once (so their handles exist).
    row_range=(r0, r1) renders only tile rows [r0, r1) into `out` (a caller-owned framebuffer of
    at least H rows: the multi-GPU gather buffer); M is then the band's intersection count.
    `lane` selects an independent set of scratch buffers (multi-view batches keep two frames in
    flight on two streams, each with its own lane)."""
    L = _hip.lib()
    dev = means3d.device
    N = means3d.shape[0]
    means3d, scales, quats = _hip.f32c(means3d), _hip.f32c(scales), _hip.f32c(quats)
    op = _hip.f32c(opacities.reshape(-1))
    if colors.dtype == torch.float16:
        cdt, colors = 1, colors.contiguous()
    else:
        cdt, colors = 0, _hip.f32c(colors)
    C = colors.shape[1]
    assert means3d.shape == (N, 3) and scales.shape == (N, 3) and quats.shape == (N, 4)
    assert op.shape == (N,) and colors.shape == (N, C)
    bg = None if background is None else _hip.f32c(background.reshape(-1))
    H, W = camera.H, camera.W
    th, tw = -(-H // tile_size), -(-W // tile_size)
    vm = camera._viewmat_f32()
    if vm.device != dev:
        vm = vm.to(dev)

    st = _dev_state(dev, lane)
    ws = _grow(st, "ws", L.ms_render_workspace_bytes(N, tw, th), dev)
    isect = st["isect"]
    host = st["host"]
    r0, r1 = (0, th) if row_range is None else row_range
    if out is not None:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.device == dev
        assert out.shape[0] >= H and tuple(out.shape[1:]) == (W, C)
        img = out
    else:
        img = torch.empty((H, W, C), dtype=torch.float32, device=dev)
    evs = None
    if stage_events is not None:
        evs = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(e.cuda_event) for e in stage_events])

    def call(resume):
        return L.ms_render_fwd(
            N, _hip.ptr(means3d), _hip.ptr(scales), 1, _hip.ptr(quats), _hip.ptr(op), _hip.ptr(colors), cdt, C,
            _hip.ptr(vm), camera.fx, camera.fy, camera.cx, camera.cy, W, H, EPS2D, camera.near, camera.far,
            tile_size, r0, r1, _hip.ptr(bg), _hip.ptr(ws), ws.numel(), _hip.ptr(isect),
            0 if isect is None else isect.numel(), ctypes.c_void_p(host.data_ptr()), resume, _hip.ptr(img),
            evs, ctypes.c_void_p(st["ev"].cuda_event) if st.get("speculate", True) else None,
            _hip.stream(dev))

    with torch.cuda.device(dev):
        rc = call(0)
        if rc == 2:  # MS_ERR_WORKSPACE: the intersection buffer is too small for this frame's M
            need = int(host[5])
            if need > 0 and (isect is None or isect.numel() < need):
                isect = _grow(st, "isect", need, dev, slack=1.25)
                rc = call(1)
        _hip.check(rc, "ms_render_fwd")
    # a frame whose tiles need the merge-fallback sort cannot run sync-free (the library redoes it
    # on the exact path); do not speculate on the next frame of such a scene
    st["speculate"] = int(host[4]) == 0
    return img, int(host[0])


@torch.no_grad()
def render_batch_hip(means3d, scales, quats, opacities, colors, cameras, background, tile_size):
    """Render the same Gaussians from several cameras -> (C, H, W, channels) f32.

    Views are independent, so two are kept in flight: view i+1's projection / counting (memory and
    latency bound) runs on a second stream beside view i's rasteriser (VALU bound).  Each lane has
    its own scratch; the caller's stream waits for both lanes before the batch is handed back."""
    dev = means3d.device
    H, W = cameras[0].H, cameras[0].W
    assert all(c.H == H and c.W == W for c in cameras), "all cameras of a batch share one image size"
    C = colors.shape[1]
    out = torch.empty((len(cameras), H, W, C), dtype=torch.float32, device=dev)
    cur = torch.cuda.current_stream(dev)
    lanes = _lane_streams(dev)
    ready = torch.cuda.Event()
    ready.record(cur)
    counts = []
    for i, cam in enumerate(cameras):
        lane = 1 + (i % len(lanes))
        s = lanes[lane - 1]
        s.wait_event(ready)
        with torch.cuda.stream(s):
            _, m = render_fwd_hip(means3d, scales, quats, opacities, colors, cam, background, tile_size,
                                  out=out[i], lane=lane)
        counts.append(m)
    for s in lanes:
        cur.wait_stream(s)
    for t in (out, means3d, scales, quats, opacities, colors):
        for s in lanes:
            t.record_stream(s)
    return out, counts


_lanes = {}


def _lane_streams(dev):
    ls = _lanes.get(dev)
    if ls is None:
        ls = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        _lanes[dev] = ls
    return ls
