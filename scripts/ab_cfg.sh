#!/bin/bash
# per-kernel averages of one config's frame under rocprofv3: scripts/ab_cfg.sh cfg4 [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cfg=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/abcfg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abcfg -- python3 $R/scripts/config_sweep.py $cfg > /tmp/abcfg.json 2>/tmp/abcfg.err
f=$(find /tmp/abcfg -name "*kernel_stats.csv" | head -1)
echo "$cfg $*: $(python3 $R/scripts/kstats.py $f 100)"
python3 -c "import json; d=json.loads([l for l in open('/tmp/abcfg.json') if l.startswith('{')][-1]); print('   ms_fwd', d.get('ms_fwd'), 'bin', d.get('bin_px'))"
