#!/bin/bash
# A/B of environment switches under rocprofv3: scripts/ab_env.sh "" "MOJOSPLAT_LEAN=0" "MOJOSPLAT_LEAN=0 MOJOSPLAT_DEFER_TOTAL=0"
# -> per-kernel average us of `python bench.py --no-extras` (cfg3) and the bench line's own numbers, one line per variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
WL=${AB_WORKLOAD:-cfg3}
i=0
for v in "$@"; do
  i=$((i+1))
  ( for kv in $v; do export "$kv"; done
    rm -rf /tmp/abenv$i
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abenv$i -- python3 $R/bench.py --steps 100 --workload $WL --no-cpu-baseline --no-verify --no-extras > /tmp/abenv$i.json 2>/tmp/abenv$i.err
    f=$(find /tmp/abenv$i -name "*kernel_stats.csv" | head -1)
    echo "variant '$v': $(python3 $R/scripts/kstats.py $f 50)"
    python3 -c "import json,sys; d=json.loads(open('/tmp/abenv$i.json').read().strip().splitlines()[-1]); print('   bench', d['value'], d['ms_per_step'], d['ms_per_step_mean'], 'verified', d['verified'])" || tail -5 /tmp/abenv$i.err )
done
