"""Differentiable PyTorch restatement of projection + rasteriser -- TEST INFRASTRUCTURE ONLY.

Supplies gradient oracles through autograd (the reference has no backward at all:
mojosplat/render.py:11, README.md:145, so backward parity is "unpinned" against the reference
and pinned only against this restatement + finite differences).  Forward follows the same
reference lines as oracle/gsplat_oracle.c: projection mojosplat/kernels/projection.mojo:50-257,
rasteriser mojosplat/kernels/rasterization.mojo:75-162.  float64 throughout by default so
that it doubles as a high-precision check of the fp32 kernels' gradients.
"""
import torch

ALPHA_THRESHOLD = 1.0 / 255.0


def project(means3d, scales, quats, viewmat, fx, fy, cx, cy, W, H, eps2d=0.3, scales_are_log=True):
    """-> means2d (N,2), conics (N,3), depths (N,) ; no culling (mask with the radii you trust)."""
    V = viewmat.to(means3d.dtype)
    Rv, tv = V[:3, :3], V[:3, 3]
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    s = torch.exp(scales) if scales_are_log else scales
    M = R * s[:, None, :]
    cov = M @ M.transpose(1, 2)
    mc = means3d @ Rv.T + tv
    cc = Rv @ cov @ Rv.T
    X, Y, Z = mc.unbind(-1)
    tan_fovx, tan_fovy = 0.5 * W / fx, 0.5 * H / fy
    lxp, lxn = (W - cx) / fx + 0.3 * tan_fovx, cx / fx + 0.3 * tan_fovx
    lyp, lyn = (H - cy) / fy + 0.3 * tan_fovy, cy / fy + 0.3 * tan_fovy
    tx = Z * torch.clamp(X / Z, min=-lxn, max=lxp)
    ty = Z * torch.clamp(Y / Z, min=-lyn, max=lyp)
    O = torch.zeros_like(Z)
    J = torch.stack([fx / Z, O, -fx * tx / Z ** 2, O, fy / Z, -fy * ty / Z ** 2], -1).reshape(-1, 2, 3)
    c2 = J @ cc @ J.transpose(1, 2)
    a, b, c = c2[:, 0, 0] + eps2d, c2[:, 0, 1], c2[:, 1, 1] + eps2d
    det = a * c - b * c2[:, 1, 0]
    conics = torch.stack([c / det, -b / det, a / det], -1)
    means2d = torch.stack([fx * X / Z + cx, fy * Y / Z + cy], -1)
    return means2d, conics, Z


def rasterize(means2d, conics, colors, opacities, background, tile_ranges, flatten_ids, H, W, ts):
    """Per-tile vectorised compositor with the exact branch rules of the kernels:
    skip sigma<0 / alpha<1/255; stop before adding when T(1-alpha) <= 1e-4.
    -> image (H,W,C), alphas (H,W)."""
    dt, dev = means2d.dtype, means2d.device
    C = colors.shape[1]
    th, tw = tile_ranges.shape[:2]
    img = torch.zeros(H, W, C, dtype=dt, device=dev)
    alph = torch.zeros(H, W, dtype=dt, device=dev)
    ranges = tile_ranges.tolist()
    ids_all = flatten_ids.long()
    for ty in range(th):
        for tx in range(tw):
            y0, x0 = ty * ts, tx * ts
            y1, x1 = min(y0 + ts, H), min(x0 + ts, W)
            if y1 <= y0 or x1 <= x0:
                continue
            s, e = ranges[ty][tx]
            ys = torch.arange(y0, y1, dtype=dt, device=dev) + 0.5
            xs = torch.arange(x0, x1, dtype=dt, device=dev) + 0.5
            py, px = torch.meshgrid(ys, xs, indexing="ij")
            P = py.numel()
            if e <= s:
                T_fin = torch.ones(P, dtype=dt, device=dev)
                col = torch.zeros(P, C, dtype=dt, device=dev)
            else:
                g = ids_all[s:e]
                dx = means2d[g, 0][:, None] - px.reshape(1, -1)
                dy = means2d[g, 1][:, None] - py.reshape(1, -1)
                ca, cb, cc = conics[g, 0][:, None], conics[g, 1][:, None], conics[g, 2][:, None]
                sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
                alpha = torch.clamp(opacities[g][:, None] * torch.exp(-sigma), max=0.999)
                mask = (sigma >= 0) & (alpha >= ALPHA_THRESHOLD)
                a = torch.where(mask, alpha, torch.zeros_like(alpha))
                Tn = torch.cumprod(1 - a, dim=0)                       # T after each Gaussian
                live = mask & (Tn.detach() > 1e-4)                     # prefix-closed (Tn monotone)
                a = torch.where(live, a, torch.zeros_like(a))
                Tn = torch.cumprod(1 - a, dim=0)
                Tprev = torch.cat([torch.ones_like(Tn[:1]), Tn[:-1]], 0)
                wgt = a * Tprev
                col = wgt.transpose(0, 1) @ colors[g]
                T_fin = Tn[-1]
            if background is not None:
                col = col + T_fin[:, None] * background[None, :]
            img[y0:y1, x0:x1] = col.reshape(y1 - y0, x1 - x0, C)
            alph[y0:y1, x0:x1] = (1 - T_fin).reshape(y1 - y0, x1 - x0)
    return img, alph


def sh_colors(means3d, campos, coeffs, degree, clamp=True, radii=None):
    """Differentiable (float64-friendly) restatement of orc_sh_fwd (gsplat_oracle.c): colours
    (N,3) from SH coefficients (N,K,3) along normalise(mean - campos)."""
    d = means3d - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    b = [0.28209479177387814 * torch.ones_like(x),
         -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x,
         1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.31539156525252005 * (3 * zz - 1),
         -1.0925484305920792 * x * z, 0.5462742152960396 * (xx - yy),
         -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * x * y * z,
         -0.4570457994644658 * y * (5 * zz - 1), 0.3731763325901154 * z * (5 * zz - 3),
         -0.4570457994644658 * x * (5 * zz - 1), 1.445305721320277 * z * (xx - yy),
         -0.5900435899266435 * x * (xx - 3 * yy),
         2.5033429417967046 * x * y * (xx - yy), -1.7701307697799304 * y * z * (3 * xx - yy),
         0.9461746957575601 * x * y * (7 * zz - 1), -0.6690465435572892 * y * z * (7 * zz - 3),
         0.10578554691520431 * (35 * zz * zz - 30 * zz + 3), -0.6690465435572892 * x * z * (7 * zz - 3),
         0.47308734787878004 * (xx - yy) * (7 * zz - 1), -1.7701307697799304 * x * z * (xx - 3 * yy),
         0.6258357354491761 * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    ku = (degree + 1) ** 2
    B = torch.stack(b[:ku], dim=-1)
    col = (B.unsqueeze(-1) * coeffs[:, :ku]).sum(dim=1)
    if clamp:
        col = (col + 0.5).clamp_min(0.0)
    if radii is not None:
        col = col * ((radii[:, 0] > 0) & (radii[:, 1] > 0)).unsqueeze(-1).to(col.dtype)
    return col
