"""More seeds of tests/test_hip_fused.py::test_depth_cut_fuzz_against_stagewise (or, with `band`, of
::test_band_depth_cut_fuzz_against_stagewise; with `pipelined`, of ::test_pipelined_band_fuzz_with_deferred_clean_up):
python scripts/fuzz_cut.py [first] [count] [dense] [band | pipelined]
(dense: 250 k - 1 M Gaussians at 1000-1920 x 600-1080, opaque enough for lazily sorted fronts: most frames take the cut)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest, torch
import test_hip_fused as T

dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dense = len(sys.argv) > 3 and 'dense' in sys.argv[3:]
band = 'band' in sys.argv[3:]
pipelined = 'pipelined' in sys.argv[3:]
bad = 0
total = {}
mp = pytest.MonkeyPatch()
for seed in range(first, first + count):
    try:
        if pipelined:
            T.test_pipelined_band_fuzz_with_deferred_clean_up(dev, seed, dense)
        elif band:
            T.test_band_depth_cut_fuzz_against_stagewise(dev, seed, dense)
        else:
            T.test_depth_cut_fuzz_against_stagewise(dev, mp, seed, dense)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:300], flush=True)
    for k_, v_ in T.LAST_CUT_FUZZ_STATS.items():
        total[k_] = total.get(k_, 0) + v_
    if (seed - first) % 10 == 9:
        print("seeds", first, "..", seed, "failures so far:", bad, flush=True)
mp.undo()
print("done:", count, "seeds, failures:", bad, "frame statistics over all seeds:", total)
sys.exit(1 if bad else 0)
