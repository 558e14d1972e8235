"""CPU oracle for the mojosplat hot path -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front end of oracle/gsplat_oracle.c (see that file's header for which
reference lines each function follows and how parity is pinned).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (mojosplat_amd) never does.
"""
import ctypes
import numpy as np

from . import build as _build

SEM_GSPLAT = 0
SEM_TORCH = 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build()
        L = ctypes.CDLL(path)
        L.orc_isect_count.restype = ctypes.c_int64
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def project_fwd(means3d, scales, quats, opacities, viewmat, fx, fy, cx, cy, W, H, *,
                scales_are_log=True, eps2d=0.3, near=0.1, far=100.0, radius_clip=0.0,
                semantics=SEM_GSPLAT):
    """-> means2d (N,2) f32, conics (N,3) f32, depths (N,) f32, radii (N,2) i32."""
    means3d, scales, quats = _f32(means3d), _f32(scales), _f32(quats)
    N = means3d.shape[0]
    op = _f32(opacities).reshape(-1) if opacities is not None else None
    V = _f32(viewmat).reshape(16)
    means2d = np.empty((N, 2), np.float32)
    conics = np.empty((N, 3), np.float32)
    depths = np.empty((N,), np.float32)
    radii = np.empty((N, 2), np.int32)
    rc = lib().orc_project_fwd(
        ctypes.c_int64(N), _p(means3d), _p(scales), ctypes.c_int(int(scales_are_log)), _p(quats),
        _p(op), _p(V), ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx),
        ctypes.c_float(cy), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(eps2d),
        ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(radius_clip),
        ctypes.c_int(semantics), _p(means2d), _p(conics), _p(depths), _p(radii))
    assert rc == 0
    return means2d, conics, depths, radii


def bin_tiles(means2d, radii, depths, H, W, tile_size, *, row_begin=0, row_end=None,
              return_keys=False):
    """gsplat-semantics binning -> flatten_ids (M,) i32, tile_ranges (th,tw,2) i32
    [, isect_ids (M,) i64, tiles_per_gauss (N,) i32]."""
    means2d, radii, depths = _f32(means2d), _i32(radii), _f32(depths)
    N = means2d.shape[0]
    th = -(-H // tile_size)
    tw = -(-W // tile_size)
    if row_end is None:
        row_end = th
    tpg = np.empty((N,), np.int32)
    L = lib()
    M = L.orc_isect_count(ctypes.c_int64(N), _p(means2d), _p(radii), ctypes.c_int(tile_size),
                          ctypes.c_int(tw), ctypes.c_int(th), ctypes.c_int(row_begin),
                          ctypes.c_int(row_end), _p(tpg))
    keys = np.empty((M,), np.int64)
    ids = np.empty((M,), np.int32)
    ranges = np.zeros((th, tw, 2), np.int32)
    rc = L.orc_isect_sorted(ctypes.c_int64(N), _p(means2d), _p(radii), _p(depths),
                            ctypes.c_int(tile_size), ctypes.c_int(tw), ctypes.c_int(th),
                            ctypes.c_int(row_begin), ctypes.c_int(row_end), ctypes.c_int64(M),
                            _p(keys), _p(ids), _p(ranges))
    assert rc == 0, rc
    if return_keys:
        return ids, ranges, keys, tpg
    return ids, ranges


def _row_chunks(H, threads):
    n = max(1, min(threads, H))
    step = -(-H // (n * 4)) if n > 1 else H     # 4 chunks per thread: rows are not equally heavy
    return [(r, min(r + step, H)) for r in range(0, H, step)]


def default_threads():
    """Host threads the rasteriser legs use for big frames (row chunks; ctypes drops the GIL)."""
    import os
    return max(1, min(16, (os.cpu_count() or 1)))


def rasterize_fwd(means2d, conics, colors, opacities, background, tile_ranges, flatten_ids,
                  H, W, tile_size, *, f64=False, margin=False, threads=None):
    """-> (colors (H,W,C) f32, alphas (H,W) f32, last_ids (H,W) i32[, margin (H,W) f32]);
    f64=True -> colors f64 only.  margin: see orc_rasterize_fwd_rows.  threads: row chunks run on
    that many host threads (default: 1 for small frames; the result does not depend on it)."""
    means2d, conics, colors = _f32(means2d), _f32(conics), _f32(colors)
    op = _f32(opacities).reshape(-1)
    N, CD = colors.shape
    bg = _f32(background).reshape(-1) if background is not None else None
    ranges = _i32(tile_ranges)
    ids = _i32(flatten_ids).reshape(-1)
    M = ids.shape[0]
    L = lib()
    if threads is None:
        threads = default_threads() if H * W >= 512 * 512 else 1
    args = (ctypes.c_int64(N), ctypes.c_int64(M), _p(means2d), _p(conics), _p(colors),
            ctypes.c_int(CD), _p(op), _p(bg), ctypes.c_int(W), ctypes.c_int(H),
            ctypes.c_int(tile_size), _p(ranges), _p(ids))

    def run(fn, tail):
        chunks = _row_chunks(H, threads)
        call = lambda rr: fn(*args, ctypes.c_int(rr[0]), ctypes.c_int(rr[1]), *tail)
        if len(chunks) == 1:
            rcs = [call(chunks[0])]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(threads) as ex:
                rcs = list(ex.map(call, chunks))
        assert all(rc == 0 for rc in rcs), rcs

    if f64:
        out = np.empty((H, W, CD), np.float64)
        run(L.orc_rasterize_fwd_f64_rows, (_p(out),))
        return out
    out = np.empty((H, W, CD), np.float32)
    alphas = np.empty((H, W), np.float32)
    last = np.empty((H, W), np.int32)
    mg = np.empty((H, W), np.float32) if margin else None
    run(L.orc_rasterize_fwd_rows, (_p(out), _p(alphas), _p(last), _p(mg)))
    return (out, alphas, last, mg) if margin else (out, alphas, last)


def render_fwd(means3d, scales, quats, opacities, colors, viewmat, fx, fy, cx, cy, W, H, *,
               background=None, tile_size=16, near=0.1, far=100.0, margin=False, threads=None):
    """Whole forward path with gsplat semantics; mirrors render.py:63-101 incl. the
    zeros-image-when-no-intersections rule (render.py:73-76).  margin=True adds aux['margin']."""
    m2, con, dep, rad = project_fwd(means3d, scales, quats, opacities, viewmat, fx, fy, cx, cy,
                                    W, H, near=near, far=far)
    ids, ranges = bin_tiles(m2, rad, dep, H, W, tile_size)
    C = np.asarray(colors).shape[1]
    if ids.size == 0:
        aux = dict(M=0)
        if margin:   # no branch of any walk anywhere: nothing excuses a pixel
            aux["margin"] = np.full((H, W), np.inf, np.float32)
        return np.zeros((H, W, C), np.float32), aux
    bg = np.zeros((C,), np.float32) if background is None else background
    res = rasterize_fwd(m2, con, colors, opacities, bg, ranges, ids, H, W, tile_size, margin=margin,
                        threads=threads)
    img, alphas, last = res[:3]
    aux = dict(M=int(ids.size), means2d=m2, conics=con, depths=dep, radii=rad, ids=ids,
               ranges=ranges, alphas=alphas, last_ids=last)
    if margin:
        aux["margin"] = res[3]
    return img, aux


def sh_fwd(means3d, campos, coeffs, degree, *, radii=None, clamp=True, want_basis=False):
    """-> colors f32[N,3] (and the f64[N,25] basis when want_basis).  gsplat convention, see
    gsplat_oracle.c."""
    m, c = _f32(means3d), _f32(coeffs)
    N, K = c.shape[0], c.shape[1]
    cp = _f32(campos)
    r = _i32(radii) if radii is not None else None
    out = np.empty((N, 3), np.float32)
    basis = np.empty((N, 25), np.float64) if want_basis else None
    rc = lib().orc_sh_fwd(ctypes.c_int64(N), K, int(degree), _p(m), _p(cp), _p(c), _p(r), int(clamp), _p(out),
                          _p(basis))
    if rc:
        raise ValueError(f"orc_sh_fwd failed ({rc})")
    return (out, basis) if want_basis else out
