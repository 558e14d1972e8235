"""Sample renderer: the reference's render_sample.py (10k random Gaussians -> 1920x1080 image)
on the HIP backend.  Writes output/render_example.png (PIL if available, else .ppm).

    python examples/render_sample.py [--gaussians 10000] [--ell -2.0] [--out output/render_example.png]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojosplat_amd import render_gaussians  # noqa: E402
from mojosplat_amd.scenes import BACKGROUND_V1, randscene_v1  # noqa: E402


def save_image(img_u8: np.ndarray, path: str) -> str:
    try:
        from PIL import Image
        Image.fromarray(img_u8).save(path)
        return path
    except ImportError:
        path = os.path.splitext(path)[0] + ".ppm"
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (img_u8.shape[1], img_u8.shape[0]))
            f.write(img_u8.tobytes())
        return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=10_000)
    ap.add_argument("--ell", type=float, default=-2.0, help="mean log-scale (reference sample: -2.0)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--out", default="output/render_example.png")
    ap.add_argument("--raw", default=None, help="also save the float32 frame (H, W, 3) as .npy")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("backend='hip' needs a ROCm GPU")
    dev = torch.device("cuda:0")
    sc, cam = randscene_v1(args.gaussians, args.width, args.height, ell=args.ell, seed=42, device=dev)
    bg = torch.tensor(BACKGROUND_V1, device=dev)
    t0 = time.perf_counter()
    img = render_gaussians(sc["means3d"], sc["scales"], sc["quats"], sc["opacities"], sc["features"], cam,
                           background_color=bg, backend="hip")
    torch.cuda.synchronize()
    print(f"rendered {tuple(img.shape)} in {(time.perf_counter() - t0) * 1e3:.2f} ms (first call), "
          f"range [{img.min().item():.4f}, {img.max().item():.4f}]")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    if args.raw:
        np.save(args.raw, img.cpu().numpy())
    u8 = (img.clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
    print("saved", save_image(u8, args.out))


if __name__ == "__main__":
    main()
