#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
  export MOJOSPLAT_BWD_ZERO_ROWS=$v
  rm -rf /tmp/abz
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abz -- python3 $R/scripts/bwd_probe.py > /tmp/abz.json 2>/tmp/abz.err
  f=$(find /tmp/abz -name "*kernel_stats.csv" | head -1)
  echo "zero_rows=$v: $(python3 $R/scripts/kstats.py $f 50)"
  tail -1 /tmp/abz.json | cut -c1-120
done
