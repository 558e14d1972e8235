"""SH colour kernel timing (forward / backward) at 1M Gaussians; HBM bytes = 12*K*N + 24*N."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mojosplat_amd.scenes import randscene_v1  # noqa: E402
from mojosplat_amd.sh import evaluate_sh_hip  # noqa: E402

dev = torch.device("cuda", 0)
N = 1_000_000
sc, cam = randscene_v1(N, 1920, 1080, ell=-4.0, seed=42, device=dev)
for degree in (0, 1, 2, 3, 4):
    K = (degree + 1) ** 2
    coeffs = torch.randn(N, K, 3, device=dev) * 0.3
    for _ in range(5):
        evaluate_sh_hip(sc["means3d"], coeffs, cam, degree)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        evaluate_sh_hip(sc["means3d"], coeffs, cam, degree)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / 50
    from mojosplat_amd import _hip
    L = _hip.lib()
    col = evaluate_sh_hip(sc["means3d"], coeffs, cam, degree)
    v = torch.randn(N, 3, device=dev)
    vc, vm = torch.empty_like(coeffs), torch.empty_like(sc["means3d"])
    cp = cam._campos()

    def bwd(want_means):
        _hip.check(L.ms_spherical_harmonics_bwd(N, K, degree, _hip.ptr(sc["means3d"]), cp[0], cp[1], cp[2],
                                                _hip.ptr(coeffs), None, 1, _hip.ptr(col), _hip.ptr(v), _hip.ptr(vc),
                                                _hip.ptr(vm) if want_means else None, _hip.stream(dev)))
    res = []
    for wm in (False, True):
        for _ in range(3):
            bwd(wm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            bwd(wm)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 50)
    bytes_fwd = (12 * K + 24) * N
    print(f"degree {degree} K={K}: fwd {fwd*1e6:.1f} us = {bytes_fwd/fwd/1e12:.2f} TB/s; "
          f"bwd(coeffs) {res[0]*1e6:.1f} us = {(12*K+36)*N/res[0]/1e12:.2f} TB/s; "
          f"bwd(coeffs+means) {res[1]*1e6:.1f} us = {(24*K+48)*N/res[1]/1e12:.2f} TB/s", flush=True)
