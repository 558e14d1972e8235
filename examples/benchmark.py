"""Full-pipeline benchmark (the script the reference's README.md:129-130 names but does not ship):
frames/s of render_gaussians(backend='hip') per stage and end to end, for a list of scene sizes.

    python examples/benchmark.py [--sizes 100000 1000000] [--width 1920 --height 1080] [--ell -4.0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[100_000, 1_000_000])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ell", type=float, default=-4.0)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import stage_bench
    for n in a.sizes:
        sys.argv = ["stage_bench.py", str(n), str(a.width), str(a.height), str(a.ell), str(a.iters)]
        stage_bench.main()


if __name__ == "__main__":
    main()
