"""Generate tests/golden/*.npz from the REFERENCE's own backend="torch" path.

Runs ONLY in the build container (needs /root/reference).  The reference never travels to
the GPU box: the fixtures below are plain data (inputs + the reference's outputs).

The reference's projection module imports the third-party `max.torch.CustomOpLibrary` at
import time (reference mojosplat/projection.py:9,13); that package is absent here, so an
inert placeholder module is registered under that name before the import.  Only the
reference's pure-PyTorch functions are executed (project_gaussians(backend="torch"),
bin_gaussians_to_tiles(backend="torch")); no Mojo / gsplat code path is touched.

Usage: python scripts/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)


def import_reference():
    if "max" not in sys.modules:
        mx = types.ModuleType("max")
        mxt = types.ModuleType("max.torch")

        class CustomOpLibrary:  # placeholder: never called by the torch backend
            def __init__(self, *a, **k):
                pass

        mxt.CustomOpLibrary = CustomOpLibrary
        mx.torch = mxt
        sys.modules["max"] = mx
        sys.modules["max.torch"] = mxt
    sys.path.insert(0, REF)
    import mojosplat.projection as rp
    import mojosplat.binning as rb
    import mojosplat.utils as ru
    return rp, rb, ru


def scene_proj_tests(N, seed=42):
    """Scene of reference tests/test_projection_mojo.py:34-46, CPU generator."""
    gen = torch.Generator().manual_seed(seed)
    means3d = torch.randn(N, 3, generator=gen) * 2.0
    means3d[:, 2] = means3d[:, 2].abs() + 1.0
    scales = torch.log(torch.rand(N, 3, generator=gen) * 0.3 + 0.05)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=gen), p=2, dim=-1)
    opac = torch.sigmoid(torch.randn(N, 1, generator=gen))
    return means3d, scales, quats, opac.view(-1)


def scene_raster_tests(N, seed):
    """Scene of reference tests/test_rasterization.py:24-36, CPU generator."""
    gen = torch.Generator().manual_seed(seed)
    means3d = torch.randn(N, 3, generator=gen) * 1.0
    means3d[:, 2] = torch.rand(N, generator=gen) * 3.5 + 1.5
    ls = torch.ones(N, 3) * -2.0 + torch.randn(N, 3, generator=gen) * 0.1
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=gen), dim=1)
    opac = torch.rand(N, generator=gen) * 0.45 + 0.5
    colors = torch.rand(N, 3, generator=gen)
    return means3d, ls, quats, opac, colors


def main():
    from mojosplat_amd.scenes import randscene_v1

    rp, rb, ru = import_reference()
    os.makedirs(OUT, exist_ok=True)

    def cam_simple(T, H=64, W=64, f=100.0):
        return ru.Camera(R=torch.eye(3), T=torch.tensor(T, dtype=torch.float32), H=H, W=W,
                         fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, near=0.1, far=100.0)

    cases = {}
    m, s, q, o = scene_proj_tests(100)
    cases["proj_identity_n100"] = (m, s, q, o, None, cam_simple([0.0, 0.0, 0.0]))
    m, s, q, o = scene_proj_tests(500)
    cases["proj_offset_n500"] = (m, s, q, o, None, cam_simple([0.0, 0.0, 5.0]))
    m, s, q, o, c = scene_raster_tests(200, 200)
    cases["raster_scene_n200"] = (m, s, q, o, c, cam_simple([0.0, 0.0, 0.0]))
    m, s, q, o, c = scene_raster_tests(100, 3)
    cases["raster_scene_128_n100"] = (m, s, q, o, c, cam_simple([0.0, 0.0, 0.0], 128, 128, 200.0))
    sc, cam = randscene_v1(1000, 256, 256, ell=-2.0, seed=42)
    rcam = ru.Camera(R=cam.R, T=cam.T, H=cam.H, W=cam.W, fx=cam.fx, fy=cam.fy, cx=cam.cx,
                     cy=cam.cy, near=cam.near, far=cam.far)
    cases["cfg1_randscene_n1000_256"] = (sc["means3d"], sc["scales"], sc["quats"],
                                         sc["opacities"], sc["features"], rcam)
    sc, cam = randscene_v1(5000, 640, 360, ell=-3.0, seed=7)
    rcam = ru.Camera(R=cam.R, T=cam.T, H=cam.H, W=cam.W, fx=cam.fx, fy=cam.fy, cx=cam.cx,
                     cy=cam.cy, near=cam.near, far=cam.far)
    cases["randscene_n5000_640x360"] = (sc["means3d"], sc["scales"], sc["quats"],
                                        sc["opacities"], sc["features"], rcam)

    for name, (m, s, q, o, c, cam) in cases.items():
        means2d, conics, depths, radii = rp.project_gaussians(m, s, q, o.view(-1, 1), cam,
                                                              backend="torch")
        d = dict(means3d=m.numpy(), scales=s.numpy(), quats=q.numpy(), opacities=o.numpy(),
                 viewmat=cam.view_matrix.numpy(), intr=np.array([cam.fx, cam.fy, cam.cx, cam.cy],
                                                                np.float32),
                 HW=np.array([cam.H, cam.W], np.int32),
                 nearfar=np.array([cam.near, cam.far], np.float32),
                 ref_means2d=means2d.numpy(), ref_conics=conics.numpy(),
                 ref_depths=depths.numpy(), ref_radii=radii.numpy())
        if c is not None:
            d["colors"] = c.numpy()

        # Binning golden: reference torch binning on the visible subset.  Its semantics and
        # gsplat's coincide when radii > 0, depths are distinct and no bbox edge sits exactly
        # on a tile boundary -- asserted here.
        vis = (radii[:, 0] > 0) & (radii[:, 1] > 0)
        m2v, rv, dv = means2d[vis].contiguous(), radii[vis].contiguous(), depths[vis].contiguous()
        for ts in (16,) if cam.W > 128 else (8, 16, 32):
            if m2v.shape[0] == 0:
                continue
            assert torch.unique(dv).numel() == dv.numel(), "depth ties"
            lo = (m2v - rv.float()) / ts
            hi = (m2v + rv.float()) / ts
            assert not ((lo == lo.floor()) | (hi == hi.floor())).any(), "bbox edge on a tile boundary"
            ids, ranges = rb.bin_gaussians_to_tiles(m2v, rv, dv, cam.H, cam.W, ts, backend="torch")
            d[f"bin{ts}_ids"] = ids.numpy().astype(np.int32)
            d[f"bin{ts}_ranges"] = ranges.numpy().astype(np.int32)
        d["vis_index"] = torch.nonzero(vis).view(-1).numpy().astype(np.int32)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "N", m.shape[0], "visible", int(vis.sum()),
              {k: v.shape for k, v in d.items() if k.startswith("bin")})


if __name__ == "__main__":
    main()
