#!/bin/bash
# every kernel of the config-3 forward + backward step (rocprofv3 --kernel-trace --stats over scripts/bwd_probe.py): scripts/step_kernels.sh [out.csv]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stk -- python3 $R/scripts/bwd_probe.py > /tmp/stk.json 2>/tmp/stk.err
f=$(find /tmp/stk -name "*kernel_stats.csv" | head -1)
[ -n "$1" ] && cp $f $1
python3 $R/scripts/kstats.py $f 10 | tr ',' '\n'
tail -1 /tmp/stk.json
