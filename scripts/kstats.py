"""Print a rocprofv3 kernel_stats.csv compactly: python scripts/kstats.py <file> [min_calls]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = {}
for r in rows:
    if int(r["Calls"]) < mc:
        continue
    m = re.search(r"(k_\w+|__amd\w+|\w+_kernel\w*)", r["Name"])
    name = m.group(1) if m else r["Name"][:30]
    t = re.search(r"<([^>]*)>", r["Name"])
    if t and name.startswith("k_"):
        name += "<" + t.group(1).replace(" ", "")[:24] + ">"
    out[name] = round(float(r["AverageNs"]) / 1e3, 1)
print(out)
