"""A longer fuzz campaign than the test suite runs: the two fuzz tests of tests/test_hip_fused.py (whole
frames through every shortcut of the fused path; random row bands) over many more seeds.

    python scripts/fuzz_more.py [first_seed] [count]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_fused as T

dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    for name, fn in (("frame", T.test_fused_path_fuzz_against_stagewise), ("bands", T.test_random_row_bands_fuzz_against_stagewise)):
        try:
            fn(dev, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", name, seed, str(e)[:200], flush=True)
    if (seed - first) % 100 == 99:
        print("seeds", first, "..", seed, "failures so far:", bad, flush=True)
print("done:", count, "seeds x 2 tests, failures:", bad)
sys.exit(1 if bad else 0)
