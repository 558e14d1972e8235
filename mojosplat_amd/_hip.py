"""ctypes binding of libmojosplat_hip.so (the C ABI declared in include/mojosplat_hip.h).

This replaces the reference's plugin mechanism -- ``CustomOpLibrary(kernels_dir)`` created at
import time (reference mojosplat/projection.py:12-13, rasterization.py:9-10).  Differences by
design: the library is loaded lazily on the first ``backend="hip"`` call (so importing the
package works on a box with no GPU), it is a prebuilt gfx950 shared object rather than a
JIT-specialised kernel per (N, W, H, M) (projection.py:429-436, rasterization.py:169-179),
and a missing library or GPU is a hard error: there is no CPU fallback behind ``"hip"``.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import torch

# MOJOSPLAT_HIP_LIB: another build of the same library (the -DMS_DIAG one of scripts/raster_waves.py)
_LIB_PATH = os.environ.get("MOJOSPLAT_HIP_LIB") or \
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmojosplat_hip.so")
_lib = None

ABI_VERSION = 3

# name -> (restype, argtypes); mirrors include/mojosplat_hip.h one to one
_SIGNATURES = {
    "ms_version": (c_int, []),
    "ms_last_error_string": (ctypes.c_char_p, []),
    "ms_project_gaussians_fwd": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                         c_void_p, c_float, c_float, c_float, c_float, c_int, c_int,
                                         c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "ms_isect_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "ms_isect_tiles_count": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ms_isect_tiles_emit": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ms_project_isect_count": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                       c_float, c_float, c_float, c_int, c_int, c_float, c_float, c_float,
                                       c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ms_isect_offset_encode": (c_int, [c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ms_rasterize_to_pixels_3dgs_fwd": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                                c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "ms_rasterize_to_pixels_3dgs_bwd": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                                c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_size_t, c_int, c_void_p]),
    "ms_rasterize_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ms_render_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "ms_render_isect_bytes": (c_size_t, [c_int64, c_int]),
    "ms_render_fwd": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_float, c_float, c_float, c_float, c_int, c_int, c_float, c_float,
                              c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                              c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ms_render_workspace_layout": (c_int, [c_int64, c_int, c_int, c_void_p]),
    "ms_render_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ms_config_depth_cut": (c_int, [c_int, ctypes.c_longlong]),
    "ms_render_redo_counts": (c_int, [c_void_p, c_size_t, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "ms_render_bwd": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                              c_float, c_float, c_float, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_size_t,
                              c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ms_isect_tiles_emit_speculative": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int64,
                                                c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                                c_void_p]),
    "ms_spherical_harmonics_fwd": (c_int, [c_int64, c_int, c_int, c_void_p, c_float, c_float, c_float, c_void_p,
                                           c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ms_spherical_harmonics_bwd": (c_int, [c_int64, c_int, c_int, c_void_p, c_float, c_float, c_float, c_void_p,
                                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ms_project_gaussians_bwd": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                         c_float, c_float, c_float, c_float, c_int, c_int, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
}



class ViewLane(ctypes.Structure):
    """ms_view_lane (include/mojosplat_hip.h): one scratch set of ms_render_fwd_batch."""
    _fields_ = [("workspace", c_void_p), ("workspace_bytes", c_size_t), ("isect_buf", c_void_p),
                ("isect_bytes", c_size_t), ("host_info", c_void_p), ("sync_event", c_void_p), ("stream", c_void_p)]


_SIGNATURES["ms_render_fwd_batch"] = (c_int, [c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                              c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float,
                                              c_int, c_void_p, c_int, ctypes.POINTER(ViewLane), c_int, c_void_p,
                                              c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_size_t),
                                              ctypes.POINTER(c_int)])

class Scene(ctypes.Structure):
    """ms_scene (include/mojosplat_hip.h): the Gaussians, marshalled once per scene (+ a prepared scene's block bounds)."""
    _fields_ = [("N", c_int64), ("means3d", c_void_p), ("scales", c_void_p), ("scales_are_log", c_int), ("quats", c_void_p),
                ("opacities", c_void_p), ("colors", c_void_p), ("color_dtype", c_int), ("CDIM", c_int),
                ("block_bounds", c_void_p), ("block_size", c_int), ("n_blocks", c_int64)]


class BandLane(ctypes.Structure):
    """ms_band_lane: one scratch set + the stream a band runs on + the two events that order it against the caller's."""
    _fields_ = [("workspace", c_void_p), ("workspace_bytes", c_size_t), ("isect_buf", c_void_p), ("isect_bytes", c_size_t),
                ("host_info", c_void_p), ("sync_event", c_void_p), ("stream", c_void_p), ("in_event", c_void_p),
                ("out_event", c_void_p)]


class BandFrame(ctypes.Structure):
    """ms_band_frame: one rank's band of one frame."""
    _fields_ = [("scene", ctypes.POINTER(Scene)), ("viewmat", c_void_p), ("fx", c_float), ("fy", c_float), ("cx", c_float),
                ("cy", c_float), ("W", c_int), ("H", c_int), ("eps2d", c_float), ("near_plane", c_float), ("far_plane", c_float),
                ("tile_size", c_int), ("row_begin", c_int), ("row_end", c_int), ("flags", c_int), ("backgrounds", c_void_p),
                ("render_colors", c_void_p), ("stage_events", c_void_p)]


_SIGNATURES["ms_render_bwd_rows_bytes"] = (c_size_t, [c_int64])
_SIGNATURES["ms_render_bwd_rows"] = (c_int, [c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                             c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p])
_SIGNATURES["ms_render_bwd_finish"] = (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                               c_float, c_float, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p])
_SIGNATURES["ms_scene_block_bounds_bytes"] = (c_size_t, [c_int64, c_int])
_SIGNATURES["ms_scene_prepare"] = (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p])
_SIGNATURES["ms_render_band_begin"] = (c_int, [ctypes.POINTER(BandFrame), ctypes.POINTER(BandLane), c_void_p])
_SIGNATURES["ms_render_band_finish"] = (c_int, [ctypes.POINTER(BandFrame), ctypes.POINTER(BandLane), c_void_p, c_int,
                                               ctypes.POINTER(c_int64)])

# entry points added after ABI v1's first cut; bound when present
_OPTIONAL = {}

EXPORTS = tuple(_SIGNATURES)


class HipBackendError(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load and type the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipBackendError(
            f"{_LIB_PATH} not found: build it with `python -m mojosplat_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950). backend='hip' has no fallback.")
    L = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the ABI is incomplete
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in _OPTIONAL.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
    if L.ms_version() != ABI_VERSION:
        raise HipBackendError(f"ABI mismatch: library {L.ms_version()} != binding {ABI_VERSION}")
    _lib = L
    return L


def lib():
    """The library, for launching work: additionally requires a visible GPU."""
    global _gpu_ok
    L = load()
    if not _gpu_ok:
        if not torch.cuda.is_available():
            raise HipBackendError("backend='hip' needs a ROCm GPU (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback for this backend")
        _gpu_ok = True
    return L


_gpu_ok = False


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ms_last_error_string().decode("utf-8", "replace")
        raise HipBackendError(f"{what or 'libmojosplat_hip'} failed (status {rc}): {msg}")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device=None):
    """hipStream_t of torch's current stream on `device` (what every ms_* call is launched on)."""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else device
        return c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class on_device:
    """`with on_device(dev):` -- torch.cuda.device(dev) only when dev is not already current (the
    context manager costs several microseconds per frame otherwise)."""

    def __init__(self, dev):
        idx = dev.index if isinstance(dev, torch.device) else dev
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def require_cuda(*tensors, what="input"):
    for t in tensors:
        if t is not None and not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise ValueError(f"backend='hip': every {what} must be a CUDA/ROCm tensor")


def f32c(t):
    """contiguous fp32 view/copy (no copy when already so)"""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def config_depth_cut(mode=None, min_pairs=None):
    """Depth cut-offs of the fused frame (include/mojosplat_hip.h, ms_config_depth_cut): mode 0 never / 1 from
    `min_pairs` pairs on (the default: 6 M) / 2 on every frame that can.  Process-wide; None leaves a setting alone.
    The library reads MOJOSPLAT_DEPTH_CUT / MOJOSPLAT_DEPTH_CUT_MIN_PAIRS once -- tests and measurements that switch
    inside one process call this instead of writing os.environ."""
    check(lib().ms_config_depth_cut(-1 if mode is None else int(mode), -1 if min_pairs is None else int(min_pairs)),
          "ms_config_depth_cut")
