#!/bin/bash
# per-kernel averages of one rank's band frame, shuffled and prepared scene: scripts/band_kernels.sh cfg5 8 3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for order in given prepared; do
  rm -rf /tmp/bk
  SCENE_ORDER=$order timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python3 $R/scripts/band_profile.py $1 $2 $3 > /dev/null 2>/tmp/bk.err || { tail -3 /tmp/bk.err; continue; }
  echo "== $1 world $2 rank $3, $order scene: kernel averages, us"
  python3 $R/scripts/kstats.py $(find /tmp/bk -name "*kernel_stats.csv" | head -1) 20
done
