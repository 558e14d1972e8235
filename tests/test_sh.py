"""Spherical-harmonic colours (the TODO at reference render.py:82-87).

CPU: the oracle's basis is pinned to the mathematics (scipy's spherical harmonics, orthonormality)
because gsplat -- whose convention it restates -- is not importable in any container of this build;
the product's torch backend is checked against the oracle.  GPU: the HIP kernels against the
oracle (forward, every degree, padded K, culled rows, fp16 output) and against float64 autograd of
the oracle's torch restatement (backward), plus the render paths that consume it.
Tolerances: colours 5e-6 abs (fp32 kernel vs double oracle; sums of up to 25 O(1) terms), gradients 1e-5 relative to the
tensor's largest gradient.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_oracle
from mojosplat_amd import evaluate_sh
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1
from mojosplat_amd.sh import camera_position, sh_basis_torch


def _dirs(n, seed=0):
    d = np.random.default_rng(seed).normal(size=(n, 3)).astype(np.float32)
    return d


def test_oracle_basis_is_the_real_sh_basis_with_3dgs_signs():
    from scipy.special import sph_harm_y
    d = _dirs(3000)
    _, B = oracle.sh_fwd(d, np.zeros(3), np.zeros((len(d), 25, 3), np.float32), 4, want_basis=True)
    dn = d.astype(np.float64)
    dn /= np.linalg.norm(dn, axis=1, keepdims=True)
    theta, phi = np.arccos(dn[:, 2]), np.arctan2(dn[:, 1], dn[:, 0])
    for l in range(5):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)            # complex, Condon-Shortley phase included
            real = Y.real if m == 0 else np.sqrt(2) * (-1) ** m * (Y.real if m > 0 else Y.imag)
            want = (-1) ** m * real                          # 3DGS / gsplat sign convention
            assert np.abs(B[:, l * (l + 1) + m] - want).max() < 1e-12, (l, m)
    # known values of the convention: degree 1 is (-y, z, -x) * 0.4886
    assert np.allclose(B[:, 1:4], 0.4886025119029199 * np.stack([-dn[:, 1], dn[:, 2], -dn[:, 0]], 1), atol=1e-12)


def test_oracle_basis_is_orthonormal_on_the_sphere():
    # Gauss-Legendre in cos(theta) x uniform in phi integrates degree-8 polynomials exactly
    xs, ws = np.polynomial.legendre.leggauss(12)
    phis = np.arange(24) * (2 * np.pi / 24)
    ct, ph = np.meshgrid(xs, phis, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    d = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = (ws[:, None] * np.full_like(ph, 2 * np.pi / 24)).reshape(-1)
    # feed exact unit vectors through the means (campos 0); float32 rounding of the inputs ~1e-7
    _, B = oracle.sh_fwd(d.astype(np.float32), np.zeros(3), np.zeros((len(d), 25, 3), np.float32), 4,
                         want_basis=True)
    G = (B * w[:, None]).T @ B
    assert np.abs(G - np.eye(25)).max() < 5e-6


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
def test_torch_backend_matches_oracle(degree):
    sc, cam = randscene_v1(500, 64, 64, ell=-2.0, seed=3)
    K = 25
    coeffs = torch.randn(500, K, 3, generator=torch.Generator().manual_seed(1)) * 0.4
    radii = torch.randint(0, 3, (500, 2), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    cp = camera_position(cam).numpy()
    for r in (None, radii):
        for clamp in (True, False):
            got = evaluate_sh(sc["means3d"], coeffs, cam, degree, radii=r, clamp=clamp, backend="torch")
            want = oracle.sh_fwd(sc["means3d"].numpy(), cp, coeffs.numpy(), degree,
                                 radii=None if r is None else r.numpy(), clamp=clamp)
            assert np.abs(got.numpy() - want).max() < 2e-6
    # the product's polynomial table against the oracle's, at full precision
    d = torch.from_numpy(_dirs(200, 5)).double()
    d = d / d.norm(dim=-1, keepdim=True)
    _, B = oracle.sh_fwd(d.numpy().astype(np.float32), np.zeros(3), np.zeros((200, 25, 3), np.float32), 4,
                         want_basis=True)
    d32 = torch.from_numpy(d.numpy().astype(np.float32)).double()
    d32 = d32 / d32.norm(dim=-1, keepdim=True)
    assert (sh_basis_torch(degree, d32).numpy() - B[:, :(degree + 1) ** 2]).__abs__().max() < 1e-12


def test_argument_checks():
    sc, cam = randscene_v1(10, 32, 32, ell=-2.0, seed=3)
    with pytest.raises(ValueError, match="sh_degree"):
        evaluate_sh(sc["means3d"], torch.zeros(10, 25, 3), cam, 5, backend="torch")
    with pytest.raises(ValueError, match="coefficients"):
        evaluate_sh(sc["means3d"], torch.zeros(10, 9, 3), cam, 3, backend="torch")
    with pytest.raises(ValueError, match="Invalid backend"):
        evaluate_sh(sc["means3d"], torch.zeros(10, 9, 3), cam, 2, backend="nope")
    with pytest.raises(RuntimeError):
        evaluate_sh(sc["means3d"], torch.zeros(10, 9, 3), cam, 2, backend="gsplat")


# ---------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("degree,K", [(0, 1), (1, 4), (2, 9), (3, 16), (4, 25), (1, 16), (3, 25), (0, 16)])
def test_hip_forward_matches_oracle(device, degree, K):
    N = 10_007  # not a multiple of 64: ragged last wave
    sc, cam = randscene_v1(N, 640, 360, ell=-3.0, seed=7, device=device)
    coeffs = (torch.randn(N, K, 3, generator=torch.Generator().manual_seed(degree)) * 0.5).to(device)
    radii = torch.randint(0, 4, (N, 2), generator=torch.Generator().manual_seed(9), dtype=torch.int32).to(device)
    cp = camera_position(cam).cpu().numpy()
    for r in (None, radii):
        for clamp in (True, False):
            got = evaluate_sh(sc["means3d"], coeffs, cam, degree, radii=r, clamp=clamp, backend="hip")
            want = oracle.sh_fwd(sc["means3d"].cpu().numpy(), cp, coeffs.cpu().numpy(), degree,
                                 radii=None if r is None else r.cpu().numpy(), clamp=clamp)
            assert got.shape == (N, 3) and got.dtype == torch.float32
            assert np.abs(got.cpu().numpy() - want).max() < 5e-6, (degree, K, clamp)
    # an unaligned view of the coefficients takes the scalar staging path: same numbers
    big = torch.zeros(N * K * 3 + 1, device=device)
    big[1:] = coeffs.reshape(-1)
    view = big[1:].view(N, K, 3)
    assert view.data_ptr() % 16 != 0
    assert torch.equal(evaluate_sh(sc["means3d"], view, cam, degree), evaluate_sh(sc["means3d"], coeffs, cam, degree))


@pytest.mark.gpu
@pytest.mark.parametrize("degree,K,clamp", [(0, 1, True), (1, 4, True), (2, 9, False), (3, 16, True), (4, 25, True),
                                            (2, 16, True)])
def test_hip_backward_matches_float64_autograd(device, degree, K, clamp):
    N = 3001
    sc, cam = randscene_v1(N, 320, 200, ell=-3.0, seed=11, device=device)
    g = torch.Generator().manual_seed(100 + degree)
    coeffs = (torch.randn(N, K, 3, generator=g) * 0.5).to(device).requires_grad_()
    means = sc["means3d"].clone().requires_grad_()
    radii = torch.randint(0, 4, (N, 2), generator=g, dtype=torch.int32).to(device)
    v = torch.randn(N, 3, generator=g).to(device)
    from mojosplat_amd.sh import evaluate_sh_hip
    col = evaluate_sh_hip(means, coeffs, cam, degree, radii=radii, clamp=clamp)
    (col * v).sum().backward()

    m64 = sc["means3d"].double().cpu().requires_grad_()
    c64 = coeffs.detach().double().cpu().requires_grad_()
    ref = torch_oracle.sh_colors(m64, camera_position(cam).double().cpu(), c64, degree, clamp=clamp,
                                 radii=radii.cpu())
    assert (col.detach().cpu().double() - ref.detach()).abs().max() < 5e-6
    (ref * v.double().cpu()).sum().backward()
    for name, got, want in (("v_coeffs", coeffs.grad, c64.grad), ("v_means3d", means.grad, m64.grad)):
        got = got.cpu().double()
        want = torch.zeros_like(got) if want is None else want  # degree 0 does not depend on the direction
        assert (got - want).abs().max() <= 1e-5 * max(want.abs().max().item(), 1e-3), name
    assert (coeffs.grad[:, (degree + 1) ** 2:] == 0).all()
    off = ~((radii[:, 0] > 0) & (radii[:, 1] > 0))
    assert (coeffs.grad[off] == 0).all() and (means.grad[off] == 0).all()
    # coefficients only (no v_means3d requested) takes the kernel's other branch
    c2 = coeffs.detach().clone().requires_grad_()
    (evaluate_sh_hip(sc["means3d"], c2, cam, degree, radii=radii, clamp=clamp) * v).sum().backward()
    assert torch.equal(c2.grad, coeffs.grad)


@pytest.mark.gpu
def test_render_with_sh_coefficients(device):
    """render_gaussians(features=(N,K,3), sh_degree=d) == render_gaussians(colours evaluated first);
    the trainable path back-propagates into the coefficients."""
    import mojosplat_amd as ms
    from mojosplat_amd.autograd import render_gaussians_trainable
    N = 5000
    sc, cam = randscene_v1(N, 320, 200, ell=-2.5, seed=13, device=device)
    bg = torch.tensor(BACKGROUND_V1, device=device)
    coeffs = (torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(4)) * 0.3).to(device)
    g = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"])
    cols = evaluate_sh(sc["means3d"], coeffs, cam, 3)
    want = ms.render_gaussians(*g, cols, cam, background_color=bg)
    got = ms.render_gaussians(*g, coeffs, cam, sh_degree=3, background_color=bg)
    assert torch.equal(got, want)
    with pytest.raises(ValueError, match="sh_degree"):
        ms.render_gaussians(*g, coeffs, cam, background_color=bg)
    # degree 0 over constant coefficients = flat colour 0.2821*c + 0.5
    flat = torch.zeros(N, 16, 3, device=device)
    flat[:, 0] = torch.tensor([0.5, -0.25, 1.0], device=device)
    c0 = evaluate_sh(sc["means3d"], flat, cam, 0)
    assert torch.allclose(c0, (0.28209479177387814 * flat[:, 0] + 0.5).clamp_min(0), atol=1e-6)

    c = coeffs.clone().requires_grad_()
    img = render_gaussians_trainable(*g, c, cam, background_color=bg, sh_degree=3)
    assert (img.detach() - want).abs().max() < 1e-5
    img.square().sum().backward()
    assert c.grad is not None and torch.isfinite(c.grad).all() and c.grad.abs().max() > 0
