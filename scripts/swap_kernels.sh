#!/bin/bash
# Per-kernel durations of the frames around a scene swap (scripts/cut_miss_cost.py): the longest launches of each kernel.
# usage (on the GPU box): bash scripts/swap_kernels.sh cfg4 64 > gpurun_out/swap_kernels.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sw_prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/sw_prof -- python3 $R/scripts/cut_miss_cost.py "${1:-cfg4}" "${2:-64}" > /tmp/sw_run.txt 2> /tmp/sw_err.txt || { tail -5 /tmp/sw_err.txt; exit 1; }
cut -c1-400 /tmp/sw_run.txt
f=$(find /tmp/sw_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
    per[m.group(0)[:60] if m else r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -max(kv[1])):
    v2 = sorted(v, reverse=True)
    print(f"{k:60s} calls {len(v):4d}  longest us: " + " ".join(f"{x:9.1f}" for x in v2[:5]))
PY
