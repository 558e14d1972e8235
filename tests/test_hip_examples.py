"""GPU: the sample / benchmark callers run as a user runs them -- a child
process each -- and what they produce is checked against the oracle.

Reference counterparts: render_sample.py:115-135 (render 10k random Gaussians, save a PNG),
examples/benchmark_proj.py:279-284 (projection sweep over N), README.md:129-130 (the full-pipeline
benchmark the reference names but does not ship).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from helpers import check_image_strict, np_
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return r.stdout


def test_render_sample_writes_the_oracles_image(tmp_path):
    W, H, N, ell = 320, 192, 2000, -2.0
    out, raw = str(tmp_path / "sample.png"), str(tmp_path / "sample.npy")
    stdout = _run([os.path.join(ROOT, "examples", "render_sample.py"), "--gaussians", str(N), "--width", str(W),
                   "--height", str(H), "--ell", str(ell), "--out", out, "--raw", raw])
    assert f"rendered ({H}, {W}, 3)" in stdout and "saved" in stdout
    saved = stdout.split("saved", 1)[1].strip().splitlines()[0].strip()
    assert os.path.exists(saved) and os.path.getsize(saved) > 1000
    sc, cam = randscene_v1(N, W, H, ell=ell, seed=42)
    cpu = {k: np_(v) for k, v in sc.items()}
    ref, aux = oracle.render_fwd(cpu["means3d"], cpu["scales"], cpu["quats"], cpu["opacities"], cpu["features"],
                                 np_(cam.view_matrix), cam.fx, cam.fy, cam.cx, cam.cy, W, H,
                                 background=np.array(BACKGROUND_V1, np.float32), margin=True)
    img = np.load(raw)
    check_image_strict(img, ref, aux["margin"], tag="examples/render_sample.py 2000 Gaussians 320x192", eps=2e-5)
    # and the file on disk is that frame, quantised
    if saved.endswith(".png"):
        from PIL import Image
        u8 = np.asarray(Image.open(saved))
    else:
        with open(saved, "rb") as f:
            assert f.readline() == b"P6\n"
            w, h = (int(v) for v in f.readline().split())
            assert f.readline() == b"255\n"
            u8 = np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)
    want = (np.clip(ref, 0, 1) * 255).astype(np.uint8)
    assert u8.shape == want.shape and np.abs(u8.astype(int) - want.astype(int)).max() <= 1


def test_projection_sweep_and_pipeline_benchmark_run():
    stdout = _run([os.path.join(ROOT, "examples", "benchmark_proj.py"), "--sizes", "1000", "20000", "--backends", "hip",
                   "torch", "--runs", "2"])
    rows = [l.split() for l in stdout.splitlines() if l.strip() and l.split()[0].isdigit()]
    assert [(r[0], r[1]) for r in rows] == [("1000", "hip"), ("1000", "torch"), ("20000", "hip"), ("20000", "torch")]
    assert all(float(r[2]) > 0 for r in rows)
    for N in (1000, 20000):     # the visible count the sweep prints is the oracle's
        sc, cam = randscene_v1(N, 1920, 1080, ell=-3.0, seed=42)
        cpu = {k: np_(v) for k, v in sc.items()}
        rad = oracle.project_fwd(cpu["means3d"], cpu["scales"], cpu["quats"], cpu["opacities"], np_(cam.view_matrix),
                                 cam.fx, cam.fy, cam.cx, cam.cy, 1920, 1080)[3]
        hip = [r for r in rows if r[0] == str(N) and r[1] == "hip"][0]
        assert abs(int(hip[-1]) - int((rad > 0).all(1).sum())) <= 1
    stdout = _run([os.path.join(ROOT, "examples", "benchmark.py"), "--sizes", "20000", "--width", "640", "--height",
                   "360", "--ell", "-3.0", "--iters", "3"])
    rec = json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])
    sc, cam = randscene_v1(20000, 640, 360, ell=-3.0, seed=42)
    cpu = {k: np_(v) for k, v in sc.items()}
    _, aux = oracle.render_fwd(cpu["means3d"], cpu["scales"], cpu["quats"], cpu["opacities"], cpu["features"],
                               np_(cam.view_matrix), cam.fx, cam.fy, cam.cx, cam.cy, 640, 360)
    assert rec["N"] == 20000 and rec["T"] == 40 * 23 and abs(rec["M"] - aux["M"]) <= 2
    assert rec["fps"] > 0 and all(rec[k]["median"] > 0 for k in ("project_us", "bin_us", "raster_us", "render_us"))
