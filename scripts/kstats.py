"""Print the per-kernel averages of a rocprofv3 kernel_stats.csv (microseconds)."""
import csv
import sys

for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48].ljust(48), r["Calls"].rjust(6),
          f"{float(r['AverageNs']) / 1e3:8.1f}")
