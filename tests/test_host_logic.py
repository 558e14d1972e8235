"""CPU: host-side logic -- the API mirror, the torch backends, error behaviour, and that the
C-ABI library loads and exports every symbol include/mojosplat_hip.h declares (no compute)."""
import os
import re

import numpy as np
import pytest
import torch

import mojosplat_amd as ms
import oracle
from helpers import (camera_from_golden, golden_files, load_golden, np_, oracle_project, proj_scene,
                     raster_scene, simple_camera)
from mojosplat_amd import _hip
from mojosplat_amd.scenes import randscene_v1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_matches_reference_layout():
    R = torch.tensor([[0.0, 1, 0], [1, 0, 0], [0, 0, -1]])
    T = torch.tensor([1.0, 2.0, 3.0])
    cam = ms.Camera(R=R, T=T, H=48, W=64, fx=10.0, fy=11.0, cx=32.0, cy=24.0)
    assert cam.near == 0.1 and cam.far == 100.0
    assert torch.equal(cam.view_matrix[:3, :3], R) and torch.equal(cam.view_matrix[:3, 3], T)
    assert torch.equal(cam.view_matrix[3], torch.tensor([0.0, 0, 0, 1]))
    assert torch.equal(cam.Ks, torch.tensor([[10.0, 0, 32], [0, 11, 24], [0, 0, 1]]))


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_torch_projection_backend_matches_reference_goldens(path):
    d, c = load_golden(path)
    cam = camera_from_golden(d, c)
    t = lambda k: torch.from_numpy(d[k])
    m2, con, dep, rad = ms.project_gaussians(t("means3d"), t("scales"), t("quats"), t("opacities"),
                                             cam, backend="torch")
    assert m2.dtype == torch.float32 and rad.dtype == torch.int32
    assert m2.shape == (len(d["means3d"]), 2) and con.shape[1] == 3 and rad.shape[1] == 2
    vis = d["vis_index"]
    assert np.array_equal(np_(rad), d["ref_radii"])
    np.testing.assert_allclose(np_(m2)[vis], d["ref_means2d"][vis], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(np_(dep)[vis], d["ref_depths"][vis], rtol=1e-5, atol=1e-5)
    scale = np.abs(d["ref_conics"][vis]).max(axis=1, keepdims=True)
    assert np.max(np.abs(np_(con)[vis] - d["ref_conics"][vis]) / scale) < 1e-4


def test_config1_plumbing_cpu_torch_backend():
    """BASELINE config 1: 1k Gaussians, 256x256, backend='torch' on CPU tensors."""
    sc, cam = randscene_v1(1000, 256, 256, ell=-2.0)
    m2, con, dep, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"],
                                             sc["opacities"], cam, backend="torch")
    ids, ranges = ms.bin_gaussians_to_tiles(m2, rad, dep, 256, 256, 16, backend="torch")
    assert ids.dtype == torch.int32 and ranges.shape == (16, 16, 2) and ids.numel() > 0
    oi, orng = oracle.bin_tiles(np_(m2), np_(rad), np_(dep), 256, 256, 16)
    assert np.array_equal(oi, np_(ids)) and np.array_equal(orng, np_(ranges))


@pytest.mark.parametrize("ts", [8, 16, 32])
def test_torch_binning_equals_oracle_with_ties_and_culled(ts):
    means3d, scales, quats, opac = proj_scene(400, seed=5)
    cam = simple_camera(T=(0, 0, 5.0), H=96, W=80)
    m2, con, dep, rad = oracle_project(oracle, means3d, scales, quats, opac, cam)
    dep = np.round(dep, 1)  # force depth ties
    ids, ranges = ms.bin_gaussians_to_tiles(torch.from_numpy(m2), torch.from_numpy(rad),
                                            torch.from_numpy(dep), 96, 80, ts, backend="torch")
    oi, orng = oracle.bin_tiles(m2, rad, dep, 96, 80, ts)
    assert np.array_equal(oi, np_(ids)) and np.array_equal(orng, np_(ranges))


def test_torch_binning_row_band():
    from mojosplat_amd.binning import bin_gaussians_to_tiles_torch
    means3d, scales, quats, opac = proj_scene(300, seed=9)
    cam = simple_camera(T=(0, 0, 5.0), H=128, W=64)
    m2, con, dep, rad = oracle_project(oracle, means3d, scales, quats, opac, cam)
    ids, ranges = bin_gaussians_to_tiles_torch(torch.from_numpy(m2), torch.from_numpy(rad),
                                               torch.from_numpy(dep), 16, 4, 8, row_range=(2, 5))
    oi, orng = oracle.bin_tiles(m2, rad, dep, 128, 64, 16, row_begin=2, row_end=5)
    assert np.array_equal(oi, np_(ids)) and np.array_equal(orng, np_(ranges))


def test_invalid_backend_errors():
    sc, cam = randscene_v1(8, 64, 64)
    with pytest.raises(ValueError, match="Invalid backend"):
        ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="nope")
    with pytest.raises(ValueError, match="Invalid backend"):
        ms.bin_gaussians_to_tiles(torch.zeros(1, 2), torch.ones(1, 2, dtype=torch.int32),
                                  torch.ones(1), 64, 64, 16, backend="nope")
    with pytest.raises(ValueError, match="Invalid backend"):
        ms.rasterize_gaussians(torch.zeros(1, 2), torch.zeros(1, 3), torch.zeros(1, 3), torch.zeros(1),
                               torch.zeros(3), torch.zeros(4, 4, 2), torch.zeros(0), cam, backend="nope")
    for b in ("gsplat", "mojo"):
        with pytest.raises(RuntimeError, match="third-party"):
            ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend=b)


def test_render_rejects_cpu_tensors_like_the_reference():
    sc, cam = randscene_v1(8, 64, 64)
    with pytest.raises(ValueError, match="CUDA"):
        ms.render_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"], cam)


def test_hip_backend_never_falls_back_on_cpu():
    """backend='hip' with CPU tensors / no GPU must fail loudly, not route to torch."""
    sc, cam = randscene_v1(8, 64, 64)
    with pytest.raises((ValueError, _hip.HipBackendError)):
        ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
    with pytest.raises((ValueError, _hip.HipBackendError)):
        ms.bin_gaussians_to_tiles(torch.zeros(1, 2), torch.ones(1, 2, dtype=torch.int32), torch.ones(1),
                                  64, 64, 16, backend="hip")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mojosplat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "gsplat_oracle" not in src, f


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mojosplat_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", header))
    assert {"ms_project_gaussians_fwd", "ms_isect_tiles_count", "ms_isect_tiles_emit",
            "ms_rasterize_to_pixels_3dgs_fwd", "ms_version"} <= declared
    if not os.path.exists(_hip.library_path()):
        from mojosplat_amd.csrc import build
        build.build()
    import ctypes
    L = ctypes.CDLL(_hip.library_path())
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    Lt = _hip.load()
    assert Lt.ms_version() == _hip.ABI_VERSION
    assert Lt.ms_last_error_string() is not None
    assert Lt.ms_isect_workspace_bytes(1000, 120, 68) > 0


def test_c_abi_argument_validation_returns_status_codes():
    """Every entry point validates its arguments BEFORE touching the device and reports through the
    status code + ms_last_error_string (no exception crosses the ABI) -- checkable
    without a GPU.  Pointers are fake but non-null; nothing is dereferenced on these paths."""
    import ctypes
    L = _hip.load()
    P = ctypes.c_void_p(0x1000)           # "some pointer": validation never dereferences it
    N = ctypes.c_int64(10)
    OK, INVALID, WORKSPACE, TOO_LARGE = 0, 1, 2, 3

    def err():
        return L.ms_last_error_string().decode()

    # projection
    assert L.ms_project_gaussians_fwd(ctypes.c_int64(-1), P, P, 1, P, P, P, 1., 1., 0., 0., 8, 8, .3, .1, 10., 0.,
                                      P, P, P, P, None) == INVALID and "N < 0" in err()
    assert L.ms_project_gaussians_fwd(N, None, P, 1, P, P, P, 1., 1., 0., 0., 8, 8, .3, .1, 10., 0.,
                                      P, P, P, P, None) == INVALID and "null" in err()
    assert L.ms_project_gaussians_fwd(N, P, P, 1, P, P, P, 0., 1., 0., 0., 8, 8, .3, .1, 10., 0.,
                                      P, P, P, P, None) == INVALID and "camera" in err()
    assert L.ms_project_gaussians_fwd(ctypes.c_int64(0), None, None, 1, None, None, None, 1., 1., 0., 0., 8, 8, .3,
                                      .1, 10., 0., None, None, None, None, None) == OK   # N == 0 is a no-op
    # binning: grid / band / workspace checks
    ws_bytes = L.ms_isect_workspace_bytes(N, 4, 4)
    assert ws_bytes > 0 and L.ms_isect_workspace_bytes(N, 0, 4) == 0
    assert L.ms_isect_tiles_count(N, P, P, 16, 4, 4, 3, 2, P, ws_bytes, None, P, P, None) == INVALID \
        and "row band" in err()
    assert L.ms_isect_tiles_count(N, P, P, 16, 4, 4, 0, 4, P, 16, None, P, P, None) == WORKSPACE \
        and "workspace" in err()
    assert L.ms_isect_tiles_count(N, P, P, 16, 100000, 100000, 0, 1, P, ws_bytes, None, P, P, None) == TOO_LARGE
    big = L.ms_isect_workspace_bytes(N, 512, 512)   # 262 144 tiles: more than one LDS histogram holds
    assert L.ms_isect_tiles_count(N, P, P, 16, 512, 512, 0, 512, P, big, None, P, P, None) == TOO_LARGE \
        and "LDS" in err()
    # rasteriser
    def rast(cdim=3, dtype=0, r0=0, r1=1, M=5):
        return L.ms_rasterize_to_pixels_3dgs_fwd(N, ctypes.c_int64(M), P, P, P, dtype, cdim, P, None, 16, 16, 16,
                                                 r0, r1, P, P, P, None, None, None)
    assert rast(cdim=33) == INVALID and "CDIM" in err()
    assert rast(dtype=7) == INVALID and "dtype" in err()
    assert rast(r0=1, r1=3) == INVALID and "row band" in err()
    assert rast(r0=1, r1=1) == OK                                    # empty band: nothing to launch
    # spherical harmonics
    assert L.ms_spherical_harmonics_fwd(N, 9, 5, P, 0., 0., 0., P, None, 1, 0, P, None) == INVALID and "degree" in err()
    assert L.ms_spherical_harmonics_fwd(N, 4, 2, P, 0., 0., 0., P, None, 1, 0, P, None) == INVALID \
        and "coefficients" in err()
    # fused frame
    host = (ctypes.c_int64 * 8)()
    def frame(phase=0, r0=0, r1=1, wsb=1 << 30):
        return L.ms_render_fwd(N, P, P, 1, P, P, P, 0, 3, P, 1., 1., 0., 0., 16, 16, .3, .1, 10., 16, r0, r1, None,
                               P, wsb, None, 0, host, phase, P, None, None, None, None, None)
    assert frame(phase=9) == INVALID and "phase" in err()
    assert frame(r0=1, r1=0) == INVALID and "band" in err()
    assert frame(wsb=8) == WORKSPACE and "workspace" in err()
    assert L.ms_render_workspace_bytes(N, 1, 1) >= L.ms_isect_workspace_bytes(N, 1, 1)
    # backward of a differentiable frame
    host[0], host[6], host[7] = 100, 5, 4          # 100 pairs, 5 Gaussians on the grid, exact layout
    bws = L.ms_render_bwd_workspace_bytes(N, 3)
    assert bws >= L.ms_rasterize_bwd_workspace_bytes(N, 3) + 10 * 20

    def bwd(n=N, cdim=3, wsb=1 << 30, isb=1 << 20, bwsb=None, hinfo=host, ws=P):
        return L.ms_render_bwd(n, P, P, 1, P, P, P, cdim, P, 1., 1., 0., 0., 16, 16, .3, 16, None, ws, wsb, P, isb, hinfo,
                               P, P, P, P, None, P, P, P, P, P, P, bws if bwsb is None else bwsb, None, None)
    assert bwd(cdim=40) == INVALID and "sizes" in err()
    assert bwd(hinfo=None) == INVALID and "null" in err()
    assert bwd(ws=None) == INVALID and "null" in err()
    assert bwd(wsb=8) == WORKSPACE and "workspace" in err()
    assert bwd(bwsb=8) == WORKSPACE and "backward workspace" in err()
    assert bwd(isb=64) == WORKSPACE and "ids" in err()
    host[7] = 4 | 8
    assert bwd() == INVALID and "split" in err()
    assert bwd(n=ctypes.c_int64(0)) == OK          # nothing to differentiate
    for k in range(8):
        host[k] = 0


def test_c_abi_argument_validation_of_the_round5_entry_points():
    """The band pair, the prepared-scene call and the two halves of the backward validate before they touch the device."""
    import ctypes
    L = _hip.load()
    P = ctypes.c_void_p(0x1000)
    N = ctypes.c_int64(10)
    OK, INVALID, WORKSPACE = 0, 1, 2
    err = lambda: L.ms_last_error_string().decode()
    # prepared scenes
    # (4 blocks' bounds of 32 bytes, then a 16-byte pre-cull record per Gaussian)
    assert L.ms_scene_block_bounds_bytes(ctypes.c_int64(1000), 256) == 4 * 32 + 1000 * 16 and L.ms_scene_block_bounds_bytes(ctypes.c_int64(1000), 100) == 0
    assert L.ms_scene_block_bounds_bytes(ctypes.c_int64(1000), 32) == 0 and L.ms_scene_block_bounds_bytes(ctypes.c_int64(0), 256) == 0
    assert L.ms_scene_prepare(N, P, P, 1, 100, P, None) == INVALID and "power of two" in err()
    assert L.ms_scene_prepare(N, None, P, 1, 256, P, None) == INVALID and "null" in err()
    assert L.ms_scene_prepare(ctypes.c_int64(0), None, None, 1, 256, None, None) == OK
    # the band pair: structs as _band.py builds them
    scene = _hip.Scene(10, 0x1000, 0x1000, 1, 0x1000, 0x1000, 0x1000, 0, 3, None, 0, 0)
    lane = _hip.BandLane(0x1000, 1 << 20, None, 0, 0x1000, None, None, 0x1000, 0x1000)
    frame = _hip.BandFrame()
    frame.scene = ctypes.pointer(scene)
    frame.viewmat, frame.render_colors = 0x1000, 0x1000
    frame.W, frame.H, frame.tile_size, frame.row_begin, frame.row_end = 64, 64, 16, 0, 4
    status = (ctypes.c_int64 * 4)()
    assert L.ms_render_band_begin(None, ctypes.byref(lane), None) == INVALID and "null frame" in err()
    assert L.ms_render_band_finish(ctypes.byref(frame), None, None, 0, status) == INVALID
    bad = _hip.BandLane(None, 0, None, 0, 0x1000, None, None, 0x1000, 0x1000)
    assert L.ms_render_band_begin(ctypes.byref(frame), ctypes.byref(bad), None) == INVALID and "scratch" in err()
    bad = _hip.BandLane(0x1000, 1 << 20, None, 0, 0x1000, None, None, None, 0x1000)
    assert L.ms_render_band_begin(ctypes.byref(frame), ctypes.byref(bad), None) == INVALID and "event" in err()
    frame.render_colors = None
    assert L.ms_render_band_begin(ctypes.byref(frame), ctypes.byref(lane), None) == INVALID and "image" in err()
    frame.render_colors, frame.row_begin = 0x1000, 5
    assert L.ms_render_band_begin(ctypes.byref(frame), ctypes.byref(lane), None) == INVALID and "row band" in err()
    frame.row_begin = 0
    assert L.ms_render_band_finish(ctypes.byref(frame), ctypes.byref(lane), None, 0, None) == INVALID and "status" in err()
    # the backward's two halves
    host = (ctypes.c_int64 * 8)()
    rows_bytes = L.ms_render_bwd_rows_bytes(N)
    assert rows_bytes >= 10 * 64 and rows_bytes % 256 == 0

    def rows(n=N, hinfo=host, out=P, cdim=3, ts=16, wsb=1 << 30):
        return L.ms_render_bwd_rows(n, cdim, 64, 64, ts, 0, 4, None, P, wsb, P, 1 << 20, hinfo, P, P, P, None, out, None, None)
    assert rows(hinfo=None) == INVALID and rows(out=None) == INVALID
    assert rows(n=ctypes.c_int64(0)) == OK
    host[0], host[6], host[7] = 100, 5, 4
    assert rows(cdim=4) == INVALID and "3 channels" in err()
    assert rows(ts=24) == INVALID
    assert rows(wsb=8) == WORKSPACE and "workspace" in err()
    host[7] = 4 | 8
    assert rows() == INVALID and "split" in err()

    def finish(n=N, cdim=3, r=P, op=P, vm=P):
        return L.ms_render_bwd_finish(n, P, P, 1, P, op, cdim, P, 1., 1., 0., 0., 64, 64, .3, r, vm, P, P, P, P, None)
    assert finish(cdim=4) == INVALID and finish(vm=None) == INVALID
    assert finish(r=None) == INVALID and "null" in err()
    assert finish(n=ctypes.c_int64(0)) == OK
    counts = (ctypes.c_int32 * 2)()
    assert L.ms_render_redo_counts(P, 8, N, 4, 4, counts, None) == WORKSPACE
    assert L.ms_render_redo_counts(None, 1 << 30, N, 4, 4, counts, None) == INVALID


def test_bin_rule_on_the_baseline_configs():
    """render.py's binning-granularity rule: the footprint diameter estimated from a frame's size record
    (pairs M on the grid it ran on, Gaussians on the grid) picks split / 32 / 64 px.  The records below are
    measured ones (profiles/r02_bin_modes.txt: M per mode; on-grid counts from the scenes), with the mode that
    was fastest on the GPU with round 2's rasteriser."""
    from mojosplat_amd import render as R
    W, H = 1920, 1080
    cases = {   # name: (on_grid, {mode: M}, W, H, fastest)
        "cfg2": (95_000, {16: 195_694, 32: 195_858, 64: 140_298}, W, H, 32),            # 0.109 / 0.101 / 0.134 ms
        "cfg3": (950_000, {16: 1_966_222, 32: 1_967_590, 64: 1_406_830}, W, H, 32),     # 0.222 / 0.201 / 0.273
        # (round 3: a tie then, 0.291 against 0.265 ms now that such frames drop the pairs behind their depth cut-offs:
        # dense scenes -- ~2 700 entries per 16-px tile here -- take 64-px bins whatever their footprints)
        "cfg4": (5_700_000, {16: 10_933_155, 32: 10_942_475, 64: 8_128_726}, 1600, 1063, 64),   # 0.569 / 0.528 / 0.526
        "cfg2-heavy": (95_000, {16: 472_816, 32: 473_006, 64: 245_133}, W, H, 64),      # 0.172 / 0.146 / 0.136
        "cfg3-heavy": (950_000, {16: 4_763_542, 32: 4_765_457, 64: 2_463_575}, W, H, 64),   # 0.286 / 0.238 / 0.192
        "cfg5": (4_700_000, {16: 17_106_998, 32: 17_119_671, 64: 9_899_012}, 3840, 2160, 64),   # 1.09 / 0.91 / 0.68
    }
    for name, (n, ms_, w, h, best) in cases.items():
        for mode, m in ms_.items():       # whatever grid the previous frame ran on, the verdict is the same
            assert R.bin_rule(mode, m, n, w, h) == best, (name, mode)
    # tiny footprints (1M Gaussians at l = -5: 6 px): the split frame (0.22 / 0.32 / 1.05 ms)
    n = 900_000
    p_at = lambda d, g: (d / g + 1.0) ** 2
    assert R.bin_rule(16, int(n * p_at(6.0, 32)), n, W, H) == 16 and R.bin_rule(32, int(n * p_at(6.0, 32)), n, W, H) == 16
    # dead band: a scene sitting on a threshold keeps the mode it has
    for thr, modes in ((R._D_SPLIT, (16, 32)), (R._D_COARSE, (32, 64))):
        for mode in modes:
            g = 32 if mode == 16 else mode
            assert R.bin_rule(mode, int(n * p_at(thr * 1.02, g)), n, W, H) == mode
            assert R.bin_rule(mode, int(n * p_at(thr * 0.98, g)), n, W, H) == mode
    assert R.bin_rule(16, 0, 0, W, H) == 16 and R.bin_rule(64, 0, 5, W, H) == 64    # empty frames decide nothing
    # a rank's band (multi-GPU): its Gaussians lie only partly inside; the band-aware estimate undoes that
    # (config 5, 17-row band of 272 px: footprints of 29 px -> 64-px bins, as for the whole frame)
    h, d, g = 272.0, 29.0, 32
    n_band = 600_000
    m_band = int(n_band * p_at(d, g) * h / (h + d + g))
    assert R.bin_rule(16, m_band, n_band, 3840, int(h), band=True) == 64
    assert R.bin_rule(16, m_band, n_band, 3840, int(h), band=True, grid_px=32) == 64


def test_render_fwd_batch_argument_checks_without_a_gpu():
    """ms_render_fwd_batch (the camera-batch entry point, reference kernels/projection.mojo:32-37) validates its
    host-side arguments before it touches a device: callable on a box with no GPU."""
    import ctypes
    L = _hip.load()
    done, need, lane = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_int(0)
    lanes = (_hip.ViewLane * 2)()
    args = lambda C, n_lanes, lanes_p, d: (C, 10, None, None, 1, None, None, None, 0, 3, None, None, 64, 64, 0.3, 0.1, 100.0,
                                           16, None, n_lanes, lanes_p, 0, None, None, ctypes.byref(d), ctypes.byref(need),
                                           ctypes.byref(lane))
    err = lambda: L.ms_last_error_string().decode()
    assert L.ms_render_fwd_batch(*args(2, 0, lanes, done)) == 1 and "lane" in err()          # n_lanes out of 1..4
    assert L.ms_render_fwd_batch(*args(2, 2, lanes, done)) == 1 and "null pointer" in err()   # no view matrices
    bad = ctypes.c_int(5)
    assert L.ms_render_fwd_batch(*args(0, 1, lanes, bad)) == 1 and "views_done" in err()
    ok = ctypes.c_int(0)
    one = (_hip.ViewLane * 1)()
    assert L.ms_render_fwd_batch(*args(0, 1, one, ok)) == 1 and "lacks scratch" in err()      # lanes without buffers


def test_lane_pair_choice_from_a_calibration():
    """_fused._pick_lane_pair on records of the kind profiles/r04_two_frames_in_flight.jsonl holds: never a pair on one
    hardware queue (ratio ~1.9) while an independent one is free, never a stream on the caller's queue, the most independent
    pair otherwise, and a choice even when every pair is bad."""
    from mojosplat_amd._fused import _pick_lane_pair
    with_cur = [1.26, 1.25, 1.26, 1.26, 1.94, 1.25]
    pairs = {(0, 1): 1.94, (0, 2): 1.26, (0, 3): 1.12, (0, 4): 1.25, (0, 5): 1.92, (1, 2): 1.26, (1, 3): 1.10, (1, 4): 1.25,
             (1, 5): 1.92, (2, 3): 1.12, (2, 4): 1.12, (2, 5): 1.23, (3, 4): 1.24, (3, 5): 1.23, (4, 5): 1.23}
    assert _pick_lane_pair(pairs, with_cur) == (1, 3)
    # the only independent pairs sit on the caller's queue: a serialised pair runs at the blocking rate, a lane behind the
    # caller's waits was measured well below it -- so not stream 4
    only4 = {k: (1.10 if 4 in k else 1.9) for k in pairs}
    assert 4 not in _pick_lane_pair(only4, with_cur)
    shared = {k: 1.9 for k in pairs}
    shared[(2, 5)] = 1.3
    assert _pick_lane_pair(shared, [1.2] * 6) == (2, 5)
    assert _pick_lane_pair({(0, 1): 1.95}, [1.9, 1.9]) == (0, 1)


def test_scene_cache_key_tells_views_apart_and_registry_holds_no_scene():
    """Round-6 advisor findings, the host logic that runs without a GPU: (1) the band path's ms_scene cache key includes
    extent and strides -- a prefix view shares its base's data pointer and version counter; (2) the prepared-scene registry
    keeps weak references only, and its entry goes when the means tensor does."""
    import gc
    import weakref

    import torch
    from mojosplat_amd import _band, scene_order
    m = torch.zeros(100, 3)
    assert m[:10].data_ptr() == m.data_ptr() and m[:10]._version == m._version
    assert _band._tkey(m) != _band._tkey(m[:10])
    assert _band._tkey(m) != _band._tkey(m.double())
    assert _band._tkey(m) == _band._tkey(m)
    v0 = _band._tkey(m)
    m.add_(1.0)
    assert _band._tkey(m) != v0
    # the registry: an entry built by hand the way prepare_scene builds it
    sc = torch.zeros(100, 3)
    bounds = torch.zeros(1, 8)
    with scene_order._registry_lock:
        scene_order._registry[id(m)] = (weakref.ref(m), m._version, weakref.ref(sc), sc._version, bounds, 256)
    weakref.finalize(m, scene_order._forget, id(m))
    assert scene_order.prepared_bounds(m, sc) == (bounds, 256) or scene_order.prepared_bounds(m, sc)[1] == 256
    assert scene_order.prepared_bounds(m[:10], sc) is None      # (a view is another object)
    key = id(m)
    sc.add_(1.0)
    assert scene_order.prepared_bounds(m, sc) is None           # scales changed in place: bounds void
    del m
    gc.collect()
    assert key not in scene_order._registry
    scene_order.clear_registry()


def test_band_scene_cache_on_cpu_tensors_prefix_views_and_the_fast_path():
    """The band path's ms_scene cache (mojosplat_amd/_band.py) is host logic: the struct is built from data pointers and
    shapes, no GPU call.  Round-6 advisor scenario on CPU tensors: the full scene, then a prefix view (same data pointer, same
    version counter), then the full scene again -- every struct carries ITS tensors' N; the one-entry fast path returns the
    cached struct for the very same tensor objects and notices an in-place update."""
    import torch
    from mojosplat_amd import _band
    _band.clear_scenes()
    n = 300
    g = (torch.randn(n, 3), torch.randn(n, 3), torch.randn(n, 4), torch.rand(n), torch.rand(n, 3))
    S_full = _band.scene_struct(*g)
    assert S_full.N == n and S_full.CDIM == 3
    assert _band.scene_struct(*g) is S_full                      # fast path: same objects, same versions
    gk = tuple(t[:40] for t in g)
    S_k = _band.scene_struct(*gk)
    assert S_k is not S_full and S_k.N == 40
    assert S_k.means3d == S_full.means3d                         # (the very aliasing the key must see through)
    assert _band.scene_struct(*g) is S_full and _band.scene_struct(*tuple(t[:40] for t in g)).N == 40
    g[0].add_(1.0)                                               # in-place update: a new version -> a new struct, the old one dropped
    S2 = _band.scene_struct(*g)
    assert S2 is not S_full and S2.N == n
    assert all(v[0] is not S_full for v in _band._scenes.values())
    # a float64 / strided scene is marshalled: the struct points at the copies, which the entry keeps alive
    gd = (g[0].double(), g[1][:, [2, 1, 0]], g[2], g[3], g[4])
    S3 = _band.scene_struct(*gd)
    assert S3.N == n and S3.means3d != gd[0].data_ptr() and any(t.dtype == torch.float32 and t.data_ptr() == S3.means3d for t in S3._keep)
    _band.clear_scenes()
    assert not _band._scenes and _band._last is None
