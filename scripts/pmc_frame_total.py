"""Whole-frame HBM traffic of one workload from the FETCH_SIZE / WRITE_SIZE passes' per-kernel medians
(scripts/pmc_summary.py outputs): python scripts/pmc_frame_total.py <fetch.txt> <write.txt>.
Streaming kernels' FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; the rasteriser's and the sorts'
64-byte gather requests are counted at face value (profiles/r02_fetch_calibration.md).  One launch of each kernel per
frame is assumed (the clean-up launches k_far_regen run twice: their line carries x2)."""
import re, sys
GATHER = ("k_rasterize_fwd", "k_tile_front", "k_tile_redo", "k_tile_sort")
TWICE = ()   # (the two clean-up launches k_far_regen<0> / <1> are listed apart)


def table(path, counter):
    """kernel -> median KB per launch; kernels seen on fewer than half as many launches as the most frequent one are the
    run's first frames (split / uncut frames before the lane settles) and are left out of the steady frame's total"""
    out, n = {}, {}
    for line in open(path):
        m = re.match(r"(\S+)\s+" + counter + r"\s+median\s+([0-9.]+)\s+n=(\d+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))   # KB
            n[m.group(1)] = int(m.group(3))
    top = max(n.values()) if n else 0
    return {k: v for k, v in out.items() if 2 * n[k] >= top}


f, w = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
tot = 0.0
print(f"{'kernel':48s} {'fetched MB':>12s} {'written MB':>12s}")
for k in sorted(set(f) | set(w)):
    mult = 2 if k.startswith(TWICE) else 1
    fe = f.get(k, 0.0) * (1 if k.startswith(GATHER) else 2) * mult / 1024
    wr = w.get(k, 0.0) * mult / 1024
    tot += fe + wr
    print(f"{k:48s} {fe:12.1f} {wr:12.1f}")
print(f"{'frame total (one launch of each per frame)':48s} {tot:12.1f} MB")
