"""CPU: the oracle against the reference-generated golden vectors, plus the reference tests'
known-answer cases restated on the oracle.  No GPU needed."""
import numpy as np
import pytest

import oracle
from helpers import golden_files, load_golden


def _close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_projection_matches_reference_torch_backend(path):
    """oracle (reference-torch semantics) == reference project_gaussians(backend='torch').
    Reference's own cross-backend tolerances: means2d 1e-3, depths 1e-4, conics 1e-2
    (tests/test_projection_mojo.py:119-161); held here to 1e-4 rel / tighter abs."""
    d, cam = load_golden(path)
    m2, con, dep, rad = oracle.project_fwd(
        d["means3d"], d["scales"], d["quats"], d["opacities"], d["viewmat"], cam["fx"], cam["fy"],
        cam["cx"], cam["cy"], cam["W"], cam["H"], near=cam["near"], far=cam["far"],
        semantics=oracle.SEM_TORCH)
    vis = d["vis_index"]
    assert np.array_equal(rad, d["ref_radii"])  # integer radii: bit-exact
    _close(m2[vis], d["ref_means2d"][vis], rtol=2e-4, atol=2e-3)
    _close(dep[vis], d["ref_depths"][vis], rtol=1e-5, atol=1e-5)
    scale = np.abs(d["ref_conics"][vis]).max(axis=1, keepdims=True)
    assert np.max(np.abs(con[vis] - d["ref_conics"][vis]) / scale) < 1e-4


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_binning_matches_reference_torch_backend(path):
    """oracle binning == reference bin_gaussians_to_tiles(backend='torch') on the visible set
    (the configurations where gsplat's and the torch backend's rules coincide): bit-exact."""
    d, cam = load_golden(path)
    vis = d["vis_index"]
    for ts in (8, 16, 32):
        if f"bin{ts}_ids" not in d.files:
            continue
        ids, ranges = oracle.bin_tiles(d["ref_means2d"][vis], d["ref_radii"][vis],
                                       d["ref_depths"][vis], cam["H"], cam["W"], ts)
        assert np.array_equal(ids, d[f"bin{ts}_ids"])
        assert np.array_equal(ranges, d[f"bin{ts}_ranges"])


def test_gsplat_semantics_vs_torch_semantics():
    """gsplat mode differs from torch mode only in culling/radius rules: values agree where both
    keep the Gaussian; gsplat radii <= torch radii (opacity-aware extent is tighter)."""
    d, cam = load_golden(golden_files()[0])
    args = (d["means3d"], d["scales"], d["quats"], d["opacities"], d["viewmat"], cam["fx"],
            cam["fy"], cam["cx"], cam["cy"], cam["W"], cam["H"])
    kw = dict(near=cam["near"], far=cam["far"])
    mt, ct, dt, rt = oracle.project_fwd(*args, semantics=oracle.SEM_TORCH, **kw)
    mg, cg, dg, rg = oracle.project_fwd(*args, semantics=oracle.SEM_GSPLAT, **kw)
    keep = (rg > 0).all(1)
    assert keep.sum() > 0
    assert ((rt > 0).all(1) | ~keep).all()          # gsplat-visible => torch-visible
    assert (rg[keep] <= rt[keep]).all()
    np.testing.assert_allclose(mg[keep], mt[keep], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(dg[keep], dt[keep], rtol=1e-6)
    np.testing.assert_allclose(cg[keep], ct[keep], rtol=1e-4, atol=1e-6)
    assert (mg[~keep] == 0).all() and (cg[~keep] == 0).all() and (dg[~keep] == 0).all()


# ---- known-answer cases of reference tests/test_projection_mojo.py:203-258 --------------
def _one(mean, scale=0.1, opacity=1.0, quat=(1, 0, 0, 0), T=(0, 0, 0)):
    vm = np.eye(4, dtype=np.float32)
    vm[:3, 3] = T
    return oracle.project_fwd(np.array([mean], np.float32), np.log(np.full((1, 3), scale, np.float32)),
                              np.array([quat], np.float32), np.array([opacity], np.float32), vm,
                              100.0, 100.0, 32.0, 32.0, 64, 64, near=0.1, far=100.0)


def test_on_axis_projects_to_centre_and_depth_is_z():
    m2, con, dep, rad = _one([0, 0, 2.0])
    assert abs(m2[0, 0] - 32) < 2 and abs(m2[0, 1] - 32) < 2
    assert dep[0] == pytest.approx(2.0, abs=1e-6)
    assert (rad[0] > 0).all()


def test_low_opacity_and_behind_camera_are_culled():
    assert (_one([0, 0, 2.0], opacity=0.001)[3] == 0).all()
    assert (_one([0, 0, -1.0])[3] == 0).all()
    assert (_one([0, 0, 200.0])[3] == 0).all()  # beyond far plane


# ---- raster known answers (reference tests/test_rasterization.py:154-248) ---------------
def _render(means3d, colors, opac, scale=0.1, bg=(0, 0, 0)):
    N = len(means3d)
    img, aux = oracle.render_fwd(np.array(means3d, np.float32), np.log(np.full((N, 3), scale, np.float32)),
                                 np.tile(np.array([[1, 0, 0, 0]], np.float32), (N, 1)),
                                 np.array(opac, np.float32), np.array(colors, np.float32),
                                 np.eye(4, dtype=np.float32), 100.0, 100.0, 32.0, 32.0, 64, 64,
                                 background=np.array(bg, np.float32))
    return img, aux


def test_centre_pixel_red_and_corners_background():
    img, _ = _render([[0, 0, 2.0]], [[1, 0, 0]], [0.9], bg=(0.1, 0.2, 0.3))
    c = img[32, 32]
    assert c[0] > 0.1 and c[0] > c[1] and c[0] > c[2]
    np.testing.assert_allclose(img[0, 0], [0.1, 0.2, 0.3], atol=1e-6)


def test_brightness_monotone_in_opacity():
    vals = [_render([[0, 0, 2.0]], [[1, 1, 1]], [o])[0][32, 32, 0] for o in (0.2, 0.5, 0.9)]
    assert vals[0] < vals[1] < vals[2]


def test_nearer_gaussian_dominates():
    img, _ = _render([[0, 0, 2.0], [0, 0, 4.0]], [[1, 0, 0], [0, 1, 0]], [0.9, 0.9])
    assert img[32, 32, 0] > img[32, 32, 1]


def test_empty_ranges_give_background():
    ranges = np.zeros((4, 4, 2), np.int32)
    img, alphas, last = oracle.rasterize_fwd(np.zeros((1, 2)), np.ones((1, 3)), np.ones((1, 3)),
                                             np.ones(1), np.array([0.2, 0.4, 0.6]), ranges,
                                             np.zeros(0, np.int32), 64, 64, 16)
    np.testing.assert_allclose(img, np.broadcast_to([0.2, 0.4, 0.6], (64, 64, 3)), atol=1e-6)
    assert (alphas == 0).all()


def test_fp32_compositor_close_to_float64_second_opinion():
    d, cam = load_golden([p for p in golden_files() if "raster_scene_n200" in p][0])
    img, aux = oracle.render_fwd(d["means3d"], d["scales"], d["quats"], d["opacities"], d["colors"],
                                 d["viewmat"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["W"],
                                 cam["H"], background=np.array([0.1, 0.1, 0.1], np.float32))
    assert aux["M"] > 0
    img64 = oracle.rasterize_fwd(aux["means2d"], aux["conics"], d["colors"], d["opacities"],
                                 np.array([0.1, 0.1, 0.1]), aux["ranges"], aux["ids"], cam["H"],
                                 cam["W"], 16, f64=True)
    # fp32 accumulation error only, except pixels where a 1/255 or 1e-4 branch flips
    diff = np.abs(img - img64)
    assert np.mean(diff > 1e-4) < 2e-3
    assert np.median(diff) < 1e-6


def test_binning_structure_and_edge_cases():
    """reference tests/test_binning.py:78-100,134-194,358-373 restated on the oracle."""
    m2 = np.array([[15.5, 15.5], [300.0, 300.0], [-50.0, 10.0]], np.float32)
    rad = np.array([[8, 8], [5, 5], [4, 4]], np.int32)
    dep = np.array([1.0, 2.0, 3.0], np.float32)
    for ts in (8, 16, 32):
        ids, ranges = oracle.bin_tiles(m2, rad, dep, 64, 64, ts)
        assert ids.dtype == np.int32 and ranges.dtype == np.int32
        assert (ranges[..., 0] <= ranges[..., 1]).all() and ranges.max() <= ids.size
        assert ((ids >= 0) & (ids < 3)).all()
        assert set(ids.tolist()) == {0}          # off-image Gaussians contribute nothing
    ids, ranges = oracle.bin_tiles(m2[:1], rad[:1], dep[:1], 64, 64, 16)
    assert ids.size == 4                          # (15.5,15.5) r=8 touches 2x2 tiles
    ids, ranges = oracle.bin_tiles(np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0), 64, 64, 16)
    assert ids.size == 0 and (ranges[..., 0] == ranges[..., 1]).all()
    # radius 0 is skipped; ties in depth resolve to ascending Gaussian index
    m2 = np.array([[8.0, 8.0]] * 4, np.float32)
    rad = np.array([[3, 3], [0, 3], [3, 3], [3, 3]], np.int32)
    dep = np.array([2.0, 1.0, 2.0, 1.5], np.float32)
    ids, _ = oracle.bin_tiles(m2, rad, dep, 16, 16, 16)
    assert ids.tolist() == [3, 0, 2]


def test_differentiable_restatement_matches_c_oracle():
    """oracle/torch_oracle.py (gradient oracle) reproduces the C oracle's forward."""
    import torch
    from oracle import torch_oracle
    d, cam = load_golden([p for p in golden_files() if "raster_scene_n200" in p][0])
    m2, con, dep, rad = oracle.project_fwd(d["means3d"], d["scales"], d["quats"], d["opacities"], d["viewmat"],
                                           cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["W"], cam["H"],
                                           near=cam["near"], far=cam["far"])
    t = lambda a: torch.from_numpy(np.asarray(a)).double()
    tm2, tcon, tdep = torch_oracle.project(t(d["means3d"]), t(d["scales"]), t(d["quats"]), t(d["viewmat"]),
                                           cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["W"], cam["H"])
    vis = (rad > 0).all(1)
    np.testing.assert_allclose(tm2.numpy()[vis], m2[vis], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(tdep.numpy()[vis], dep[vis], rtol=1e-6)
    scale = np.abs(con[vis]).max(axis=1, keepdims=True)
    assert np.max(np.abs(tcon.numpy()[vis] - con[vis]) / scale) < 2e-5
    ids, ranges = oracle.bin_tiles(m2, rad, dep, cam["H"], cam["W"], 16)
    bg = np.array([0.2, 0.3, 0.4], np.float32)
    img, alphas, _ = oracle.rasterize_fwd(m2, con, d["colors"], d["opacities"], bg, ranges, ids, cam["H"], cam["W"], 16)
    timg, talpha = torch_oracle.rasterize(t(m2), t(con), t(d["colors"]), t(d["opacities"]), t(bg),
                                          torch.from_numpy(ranges), torch.from_numpy(ids), cam["H"], cam["W"], 16)
    diff = np.abs(timg.numpy() - img).max(-1)
    assert np.mean(diff > 1e-4) < 2e-3 and np.median(diff) < 1e-6
    assert np.abs(talpha.numpy() - alphas).max() < 5e-3
