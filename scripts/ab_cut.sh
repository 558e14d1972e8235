#!/bin/bash
# depth cut on / off under rocprofv3, per workload: scripts/ab_cut.sh cfg4 cfg5 -> per-kernel average us + the bench line's ms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for c in "$@"; do
  for m in 0 2; do
    export MOJOSPLAT_DEPTH_CUT=$m
    rm -rf /tmp/abcut
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abcut -- python3 $R/bench.py --workload $c --steps 100 --no-cpu-baseline --no-verify --no-extras > /tmp/abcut.json 2>/tmp/abcut.err
    f=$(find /tmp/abcut -name "*kernel_stats.csv" | head -1)
    echo "$c cut=$m: $(python3 $R/scripts/kstats.py $f 100)"
    python3 -c "import json; d=json.loads(open('/tmp/abcut.json').read().strip().splitlines()[-1]); print('   bench', d['value'], d['ms_per_step'])"
  done
done
