"""Projection-only sweep, the shape of the reference's examples/benchmark_proj.py (N = 1k .. 5M at
1920x1080, 3 warm-ups, wall clock bracketed by synchronize): backends 'hip' (GPU) and 'torch' (CPU).

    python examples/benchmark_proj.py [--sizes 1000 10000 ...] [--backends hip torch] [--runs 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojosplat_amd import project_gaussians  # noqa: E402
from mojosplat_amd.scenes import randscene_v1  # noqa: E402


def bench(backend, N, runs):
    dev = "cuda:0" if backend == "hip" else "cpu"
    sc, cam = randscene_v1(N, 1920, 1080, ell=-3.0, seed=42, device=dev)
    args = (sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], cam)
    sync = torch.cuda.synchronize if backend == "hip" else (lambda: None)
    for _ in range(3):
        project_gaussians(*args, backend=backend)
    sync()
    ts = []
    for _ in range(runs):
        sync()
        t0 = time.perf_counter()
        out = project_gaussians(*args, backend=backend)
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    vis = int((out[3] > 0).all(1).sum())
    return ts[len(ts) // 2] * 1e3, vis


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[1_000, 10_000, 100_000, 1_000_000, 5_000_000])
    ap.add_argument("--backends", nargs="+", default=["hip", "torch"])
    ap.add_argument("--runs", type=int, default=10)
    a = ap.parse_args()
    if "hip" in a.backends and not torch.cuda.is_available():
        raise SystemExit("backend 'hip' needs a ROCm GPU")
    print(f"{'N':>10} {'backend':>8} {'median ms':>11} {'MGauss/s':>10} {'GB/s (76 B/G)':>14} {'visible':>9}")
    for N in a.sizes:
        for b in a.backends:
            if b == "torch" and N > 1_000_000:
                continue  # CPU path: keep the sweep short
            ms, vis = bench(b, N, a.runs)
            print(f"{N:>10} {b:>8} {ms:>11.4f} {N / ms / 1e3:>10.1f} {76 * N / ms / 1e6:>14.1f} {vis:>9}")


if __name__ == "__main__":
    main()
