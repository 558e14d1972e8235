#!/bin/bash
# the backward rasteriser's kernels side by side on config 3's step: scripts/ab_bwd_kernels.sh mfma tree tree2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  export MOJOSPLAT_BWD_KERNEL=$k
  python3 $R/scripts/bwd_err.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$k', {n: (round(v['vs_stagewise'],7), v['finite']) for n, v in d.items()})"
  rm -rf /tmp/abk
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/scripts/bwd_probe.py > /tmp/abk.json 2>/tmp/abk.err
  f=$(find /tmp/abk -name "*kernel_stats.csv" | head -1)
  echo "$k: $(python3 $R/scripts/kstats.py $f 10 | tr ',' '\n' | grep 'bwd' | tr '\n' ' ')"
  tail -1 /tmp/abk.json | cut -c1-200
done
