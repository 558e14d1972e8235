"""GPU: the BASELINE configs at FULL size against the C oracle (configs 2, 3, 4, 5; config 2's per-stage kernels on the
oracle's inputs also live in test_hip_parity.py), through the calls a user makes.

Per config:
  * the frame of `render_gaussians(backend="hip")` -- the fused path: tight binning, lazily sorted lists,
    split frames / coarse bins by the binning rule, sync-free -- equals the per-stage path (HIP
    project -> bin -> rasterise: gsplat-exact, fully sorted lists) BIT FOR BIT, run to run;
  * HIP binning of the oracle's projected inputs equals the oracle's lists bit for bit; HIP rasteriser on
    those lists against the oracle's rasteriser under the strict bar of helpers.check_image_strict
    (<= 1e-4 abs per pixel fp32 except where a branch of the walk sits within 1e-5 of its threshold;
    zero unexplained pixels), `last_ids` equal wherever no branch is close;
  * the fused frame end to end (its own projection) against oracle.render_fwd under the same rule with the
    margin widened to 2e-5 (the two projections differ by an ulp in exp(scale): conics move by 1e-7
    relative, which can only matter where a branch is within that of its threshold);
  * config 5 additionally as the 8 tile-row bands of the multi-GPU decomposition
    (render_gaussians_sharded(rehearse=(r, 8))): assembled bands == the single-GPU frame bit for bit;
  * config 3 backward: finite, repeatable within atomic noise, fused == per-stage autograd; a 20k-Gaussian
    crop against float64 autograd of the restatement (oracle/torch_oracle.py).

Reference bar: tests/test_rasterization.py:94-146 (atol = rtol = 1e-4 mojo vs gsplat).
"""
import numpy as np
import pytest
import torch

import mojosplat_amd as ms
import oracle
from helpers import assert_grad_close, check_image_strict, np_
from mojosplat_amd import _fused
from mojosplat_amd.binning import bin_gaussians_to_tiles_hip
from mojosplat_amd.distributed import band_plan, render_gaussians_sharded
from mojosplat_amd.rasterization import rasterize_gaussians_hip
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (N, W, H, ell, fp16 colours)
    "cfg2": (100_000, 1920, 1080, -4.0, False),   # (round 4: the fused "light frame" path -- short sorts only, no clean-up launch)
    "cfg3": (1_000_000, 1920, 1080, -4.0, False),
    "cfg4": (6_000_000, 1600, 1063, -4.0, True),
    "cfg5": (5_000_000, 3840, 2160, -4.0, False),
}


def _scene(name, device):
    N, W, H, ell, fp16 = CONFIGS[name]
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42, device=device)
    if fp16:
        sc["features"] = sc["features"].half()
    return sc, cam, (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"])


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_forward_full_size_vs_oracle(device, name):
    N, W, H, ell, fp16 = CONFIGS[name]
    sc, cam, g = _scene(name, device)
    # render_gaussians casts the background to the colours' dtype (reference render.py:55): with fp16
    # colours 0.1 becomes 0.09998; every path below gets that same value
    bg = torch.tensor(BACKGROUND_V1, device=device).to(sc["features"].dtype)
    bgn = np_(bg.float())
    th, tw = -(-H // 16), -(-W // 16)

    # ---- the user's call (fused path, rule-chosen grid), three frames: first frame split, then the rule's
    _fused._state.clear()
    ms.render._bin_mode.clear(); ms.render._bin_left.clear()
    _fused.FRAME_STATS = stats = {}
    try:
        # (a fresh lane's first frame is a split frame on the exact path, the second moves to the rule's grid, the third
        # finds a clean-up count left by another grid's layout and takes no cut, the fourth leaves cut-offs, the FIFTH
        # is the first that can take them: scripts/cut_when.py)
        frames = [ms.render_gaussians(*g, cam, background_color=bg, backend="hip") for _ in range(5)]
        before_last = dict(stats)
        frames.append(ms.render_gaussians(*g, cam, background_color=bg, backend="hip"))
    finally:
        _fused.FRAME_STATS = None
    chosen = next(iter(ms.render._bin_mode.values()))
    # which shortcuts the frames that are compared below really took (the comparison is only worth what ran):
    # config 2's later frames bet on "no heavy tile" and launch the short sorts alone; configs 4 / 5 hold > 6 M pairs,
    # so the LAST frame drops the pairs behind its bins' depth cut-offs (and no speculation was lost on the way)
    last_cut = stats.get("depth_cut", 0) - before_last.get("depth_cut", 0)
    print(f"{name}: frame stats {stats}, last frame depth-cut: {last_cut}")
    assert stats.get("speculated", 0) >= 4 and stats.get("overflow", 0) <= 1
    if name in ("cfg4", "cfg5"):
        assert last_cut == 1, f"{name}: the frame compared with the oracle did not take the depth cut ({stats})"
    else:
        assert stats.get("depth_cut", 0) == 0
    # ---- per-stage HIP path
    m2, con, dep, rad = ms.project_gaussians(*g[:4], cam, backend="hip")
    ids, ranges = bin_gaussians_to_tiles_hip(m2, rad, dep, 16, tw, th)
    st = rasterize_gaussians_hip(m2, con, g[4], g[3], bg, ranges, ids, cam)
    for k, f in enumerate(frames):
        assert torch.equal(f, st), f"{name}: fused frame {k} differs from the per-stage path"
    for b in (16, 32, 64):     # every grid, explicitly: the same pixels
        assert torch.equal(ms.render_gaussians(*g, cam, background_color=bg, bin_size=b), st), (name, b)
    # every tile's list sorted by (depth bits, id)
    cnt = (ranges[..., 1] - ranges[..., 0]).flatten().long()
    key = (dep.view(torch.int32).to(torch.int64)[ids.long()] << 32) | ids.long()
    tile_of = torch.repeat_interleave(torch.arange(th * tw, device=device), cnt)
    assert bool(((tile_of[1:] > tile_of[:-1]) | ((tile_of[1:] == tile_of[:-1]) & (key[1:] > key[:-1]))).all())
    M_hip = int(ids.numel())
    del key, tile_of, cnt, ids, ranges, frames

    # ---- oracle, full size
    cpu = {k: np_(v.float()) for k, v in sc.items()}
    ref, aux = oracle.render_fwd(cpu["means3d"], cpu["scales"], cpu["quats"], cpu["opacities"], cpu["features"],
                                 np_(cam.view_matrix), cam.fx, cam.fy, cam.cx, cam.cy, W, H, background=bgn,
                                 margin=True)
    f64 = oracle.rasterize_fwd(aux["means2d"], aux["conics"], cpu["features"], cpu["opacities"], bgn, aux["ranges"],
                               aux["ids"], H, W, 16, f64=True)
    print(f"{name}: N={N} M_oracle={aux['M']} M_hip={M_hip} bin rule chose {chosen} px")
    assert abs(M_hip - aux["M"]) <= max(50, aux["M"] // 100_000)   # a handful of +-1 radius flips (expf/logf ulps)

    # ---- stage-wise: HIP binning + rasteriser on the ORACLE's projected inputs
    to = lambda a: torch.from_numpy(a).to(device)
    oids, oranges = bin_gaussians_to_tiles_hip(to(aux["means2d"]), to(aux["radii"]), to(aux["depths"]), 16, tw, th)
    assert np.array_equal(np_(oids), aux["ids"]) and np.array_equal(np_(oranges), aux["ranges"])
    img, alphas, last = rasterize_gaussians_hip(to(aux["means2d"]), to(aux["conics"]), g[4], g[3], bg, oranges, oids,
                                                cam, return_aux=True)
    check_image_strict(img, ref, aux["margin"], tag=f"{name} rasteriser on oracle inputs", eps=1e-5, f64=f64)
    calm = aux["margin"] >= 1e-5
    assert np.array_equal(np_(last)[calm], aux["last_ids"][calm])
    assert np.abs(np_(alphas) - aux["alphas"])[calm].max() <= 1e-5
    del img, alphas, last, oids, oranges

    # ---- end to end: the fused frame (GPU projection) against the oracle's frame
    check_image_strict(st, ref, aux["margin"], tag=f"{name} fused frame end to end", eps=2e-5)


def test_config5_as_eight_bands_equals_the_single_gpu_frame(device):
    """BASELINE config 5 is the 8-GPU config: each rank's band through the sharded entry point (rehearse =
    act as that rank without a process group) lands in its slab; the assembled frame is the single-GPU one."""
    sc, cam, g = _scene("cfg5", device)
    bg = torch.tensor(BACKGROUND_V1, device=device)
    H, W = cam.H, cam.W
    full = ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
    th = -(-H // 16)
    for world in (8, 3):
        rows, bands = band_plan(th, world)
        out = torch.zeros_like(full)
        for r, (r0, r1) in enumerate(bands):
            img = render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(r, world))
            y0, y1 = min(r0 * 16, H), min(r1 * 16, H)
            out[y0:y1] = img[y0:y1]
        assert torch.equal(out, full), world


def test_config5_centre_band_takes_the_depth_cut_and_stays_exact(device):
    """Round 4: a rank's band of the 8-GPU config keeps depth cut-offs of its own (threshold scaled by the band's share of
    the rows).  The centre band of eight, eight frames through the sharded entry point with the default switches: the later
    frames take the cut, and every frame's rows equal the single-GPU frame's."""
    sc, cam, g = _scene("cfg5", device)
    bg = torch.tensor(BACKGROUND_V1, device=device)
    H = cam.H
    full = ms.render_gaussians(*g, cam, background_color=bg, backend="hip")
    _, bands = band_plan(-(-H // 16), 8)
    r0, r1 = bands[3]
    y0, y1 = r0 * 16, min(r1 * 16, H)
    _fused._state.clear()
    _fused.FRAME_STATS = stats = {}
    try:
        for k in range(8):
            img = render_gaussians_sharded(*g, cam, background_color=bg, rehearse=(3, 8))
            assert torch.equal(img[y0:y1], full[y0:y1]), k
    finally:
        _fused.FRAME_STATS = None
        _fused._state.clear()
    assert stats.get("depth_cut", 0) >= 2, stats
    assert stats.get("cut_redo_tiles", 0) == 0 and stats.get("regen_mismatch", 0) == 0, stats


def test_config3_backward_full_size(device):
    """Config 3 forward + backward at full size: gradients for means / scales / quats / opacities / colours
    finite, non-zero, repeatable within the order of the float atomics, and equal to the per-stage
    autograd functions (the fused differentiable frame runs tight binning + sync-free)."""
    from mojosplat_amd.autograd import render_gaussians_trainable
    sc, cam, g = _scene("cfg3", device)
    bg = torch.tensor(BACKGROUND_V1, device=device)
    names = ("means3d", "scales", "quats", "opacities", "features")
    v_img = torch.rand(cam.H, cam.W, 3, generator=torch.Generator().manual_seed(43)).to(device)
    res = []
    for stagewise in (False, False, True):
        leaves = [sc[k].clone().requires_grad_(True) for k in names]
        img = render_gaussians_trainable(*leaves, cam, background_color=bg, stagewise=stagewise)
        (img * v_img).sum().backward()
        res.append((img.detach(), [l.grad for l in leaves]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
    assert torch.equal(res[0][0], ms.render_gaussians(*g, cam, background_color=bg, backend="hip"))
    for name, a, b, c in zip(names, res[0][1], res[1][1], res[2][1]):
        assert torch.isfinite(a).all() and a.abs().sum() > 0, name
        assert_grad_close(name + " (repeat)", a, b, rel=1e-4)
        # (round 5: ... and element by element -- every element of at least 1e-3 of the tensor's max to 1e-3 of itself)
        st = assert_grad_close(name + " (fused vs per-stage)", a, c, rel=1e-4, elem_rel=1e-3, elem_p999=5e-4)
        print("GRAD", name, st)


@pytest.mark.parametrize("x0,y0", [(640, 348), (960, 348), (640, 540), (960, 540)])
def test_config3_crop_backward_vs_float64_autograd(device, x0, y0):
    """Four disjoint 20k-Gaussian crops of config 3's scene (the Gaussians nearest the centre of a 320x192 window of
    the full frame at (x0, y0): same camera, principal point moved): HIP gradients against float64 autograd of the
    restatement, 5e-3 of each tensor's max."""
    from mojosplat_amd.autograd import project_gaussians_autograd, render_gaussians_trainable
    from mojosplat_amd.utils import Camera
    from oracle import torch_oracle
    sc, cam0, _ = _scene("cfg3", device)
    W, H = 320, 192
    cam = Camera(R=cam0.R, T=cam0.T, H=H, W=W, fx=cam0.fx, fy=cam0.fy, cx=cam0.cx - x0, cy=cam0.cy - y0, near=cam0.near,
                 far=cam0.far)
    with torch.no_grad():
        m2, _, _, rad = ms.project_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam, backend="hip")
        vis = (rad > 0).all(1)
        d2 = ((m2 - torch.tensor([W / 2.0, H / 2.0], device=device)) ** 2).sum(1)
        d2[~vis] = float("inf")
        keep = torch.argsort(d2)[:20_000]
    names = ("means3d", "scales", "quats", "opacities", "features")
    leaves = [sc[k][keep].clone().requires_grad_(True) for k in names]
    bg = torch.tensor(BACKGROUND_V1, device=device)
    v_img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(43)).to(device)
    img = render_gaussians_trainable(*leaves, cam, background_color=bg)
    img.backward(v_img)

    with torch.no_grad():
        m2h, conh, deph, radh = project_gaussians_autograd(*[l.detach() for l in leaves[:4]], cam)
    ids, ranges = bin_gaussians_to_tiles_hip(m2h, radh, deph, 16, W // 16, H // 16)
    rl = [l.detach().double().cpu().requires_grad_(True) for l in leaves]
    rm2, rcon, _ = torch_oracle.project(rl[0], rl[1], rl[2], cam.view_matrix.double().cpu(), cam.fx, cam.fy, cam.cx,
                                        cam.cy, W, H)
    rimg, _ = torch_oracle.rasterize(rm2, rcon, rl[4], rl[3], bg.double().cpu(), ranges.cpu(), ids.cpu(), H, W, 16)
    d = (img.detach().double().cpu() - rimg.detach()).abs()
    assert float((d > 1e-4).float().mean()) <= 1e-4 and float(d.max()) <= 1e-2   # branch flips only
    (rimg * v_img.double().cpu()).sum().backward()
    for name, a, b in zip(names, leaves, rl):
        st = assert_grad_close(name, a.grad, b.grad, rel=5e-3, elem_rel=2e-3)   # (round 5: the per-element bar, against float64)
        print("GRAD", name, st)


def test_scene_swap_at_config4_is_exact_and_bounded(device):
    """The worst frame the lazy machinery can meet (scripts/cut_miss_cost.py): config 4 on 64-px bins with depth cut-offs,
    then the same Gaussians with the near half all but transparent -- every bin outlives both its cut-off and its sorted
    front.  Each frame on the way must be bit-identical to the per-stage path of ITS scene.  Bounds: the swapped scene's
    frames within 15 ms each (round 2: 14 s; round 3 and the first half of round 4: 72-85 ms, one workgroup per stranded
    bin walking its sixteen blocks in turn; since the two-launch clean-up -- rasterize.hip, k_redo_sort -- 4.8-5.9 ms
    measured, against 3.9 ms for the same scene on the fully sorted path), full sorts from its third frame on at the latest
    (<= 8 ms), and the original scene again within two frames of the swap back."""
    import time
    sc, cam, g = _scene("cfg4", device)
    bg = torch.tensor(BACKGROUND_V1, device=device).to(sc["features"].dtype)
    depth = (sc["means3d"] @ cam.R.T + cam.T)[:, 2]
    other = dict(sc)
    other["opacities"] = torch.where(depth < depth.median(), sc["opacities"] * 0.02, sc["opacities"])
    tup = lambda s_: (s_["means3d"], s_["scales"], s_["quats"], s_["opacities"], s_["features"])
    th, tw = -(-cam.H // 16), -(-cam.W // 16)

    def stagewise_frame(s_):
        m2, con, dep, rad = ms.project_gaussians(*tup(s_)[:4], cam, backend="hip")
        ids, ranges = bin_gaussians_to_tiles_hip(m2, rad, dep, 16, tw, th)
        return rasterize_gaussians_hip(m2, con, s_["features"], s_["opacities"], bg, ranges, ids, cam)
    ref = {id(sc): stagewise_frame(sc), id(other): stagewise_frame(other)}
    _fused._state.clear()
    _fused.FRAME_STATS = stats = {}
    times = []
    try:
        for k in range(12):
            s_ = sc if k < 5 or k >= 9 else other
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img = ms.render_gaussians(*tup(s_), cam, background_color=bg, backend="hip", bin_size=64)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            assert torch.equal(img, ref[id(s_)]), f"frame {k} differs from the per-stage path"
    finally:
        _fused.FRAME_STATS = None
        _fused._state.clear()
    print("scene swap at config 4, ms per synchronised frame:", [round(t, 2) for t in times], stats)
    assert stats.get("depth_cut", 0) >= 2            # the steady frames before the swap were cut
    assert stats.get("redo_tiles", 0) + stats.get("cut_redo_tiles", 0) > 100   # ... and the swap did strand the bins
    assert max(times[5:9]) <= 15.0
    assert times[7] <= 8.0 and times[8] <= 8.0       # full sorts by the swapped scene's third frame
    assert max(times[10:]) <= 3.0


def test_training_step_on_stranded_bins_is_exact_and_bounded(device):
    """Round 5: the worst STEP the lazy machinery can meet -- config 3's scene with the near half all but transparent, through
    render_gaussians_trainable: the forward's lazily sorted fronts run out in most heavy bins, its two-launch clean-up pass
    sorts those bins whole and redoes them (alphas included), and the backward's redo launch walks the sorted ids, a wave
    per (bin, block, quad) (round 4: one workgroup per bin re-sorting its keys in global memory, 64 workgroups in all,
    unbounded).  Image bit for bit and gradients within the suite's bars of the per-stage functions; every step <= 20 ms;
    and the differentiable frames LEARN (advisor, round 4): once a step's clean-up count has reached the host, later steps
    run deeper fronts or full sorts and stop paying the clean-up."""
    import time
    from mojosplat_amd.autograd import render_gaussians_trainable
    sc, cam, _ = _scene("cfg3", device)
    depth = (sc["means3d"] @ cam.R.T + cam.T)[:, 2]
    sc = dict(sc)
    sc["opacities"] = torch.where(depth < depth.median(), sc["opacities"] * 0.02, sc["opacities"])
    names = ("means3d", "scales", "quats", "opacities", "features")
    bg = torch.tensor(BACKGROUND_V1, device=device)
    v_img = torch.rand(cam.H, cam.W, 3, generator=torch.Generator().manual_seed(43)).to(device)

    def step(stagewise):
        leaves = [sc[k].clone().requires_grad_(True) for k in names]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = render_gaussians_trainable(*leaves, cam, background_color=bg, stagewise=stagewise)
        img.backward(v_img)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, img.detach(), [l.grad for l in leaves]
    _, ref_img, ref_grads = step(True)
    _fused._state.clear()
    _fused.FRAME_STATS = stats = {}
    times = []
    try:
        for k in range(8):
            ms_, img, grads = step(False)
            times.append(ms_)
            assert torch.equal(img, ref_img), f"step {k}: the image differs from the per-stage path"
            for name, a, b in zip(names, grads, ref_grads):
                assert_grad_close(f"step {k} {name}", a, b, rel=2e-3, elem_rel=5e-2, elem_p999=5e-3)
    finally:
        _fused.FRAME_STATS = None
        _fused._state.clear()
    print("training steps on stranded bins, ms per synchronised step:", [round(t, 2) for t in times], stats)
    assert stats.get("own_redo_tiles", 0) > 50, "the scene did not strand its bins"
    assert max(times[1:]) <= 20.0          # (the first step includes buffer growth / kernel loading)
    assert stats.get("own_full_sort_on", 0) + stats.get("own_front_level_up", 0) >= 1, "the differentiable frames never learnt"
    assert times[-1] <= 0.75 * max(times[1:4]) or times[-1] <= 2.0, "later steps still pay the clean-up pass"
