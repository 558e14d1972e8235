#!/bin/bash
# forward + backward step of config 3 with the differentiable frame binned on 16 / 32 / 64-px tiles: scripts/ab_train_bin.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
WL=${1:-cfg3}
for px in ${TRAIN_PX:-16 32 64}; do
  export MOJOSPLAT_TRAIN_BIN_PX=$px
  rm -rf /tmp/abt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt -- python3 $R/scripts/bwd_probe.py $WL > /tmp/abt.json 2>/tmp/abt.err
  f=$(find /tmp/abt -name "*kernel_stats.csv" | head -1)
  echo "$WL train bin px $px: $(python3 $R/scripts/kstats.py $f 10)"
  tail -1 /tmp/abt.json
done
