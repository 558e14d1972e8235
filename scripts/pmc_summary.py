"""Median of a rocprofv3 --pmc counter per kernel: python scripts/pmc_summary.py <counter_collection.csv> [min_calls] [max]
("max": also print the largest value -- a microbenchmark launches each kernel a few times with different trip counts)"""
import csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
acc = {}
for r in rows:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m:
        continue
    t = re.search(r"<([^>]*)>", r["Kernel_Name"])
    name = m.group(1) + ("<" + t.group(1).replace(" ", "")[:24] + ">" if t else "")
    acc.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (name, c), v in sorted(acc.items()):
    if len(v) >= mc:
        print(f"{name:48s} {c:12s} median {statistics.median(v):14.1f}  n={len(v)}" + (f"  max {max(v):14.1f}" if "max" in sys.argv[3:] else ""))
