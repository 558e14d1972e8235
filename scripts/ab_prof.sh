#!/bin/bash
# A/B of library builds under rocprofv3: scripts/ab_prof.sh "" _mw6 ... -> per-kernel average us of python bench.py
# (run on the GPU box from the repo root; variants are libmojosplat_hip<suffix>.so built with MS_VARIANT)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in "$@"; do
  [ "$v" = "base" ] && v=""
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
  rm -rf /tmp/abprof$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof$v -- python3 $R/bench.py --steps 100 --no-cpu-baseline --no-verify > /tmp/abprof$v.json 2>/tmp/abprof$v.err
  f=$(find /tmp/abprof$v -name "*kernel_stats.csv" | head -1)
  echo "variant '$v': $(python3 $R/scripts/kstats.py $f 50)"
  python3 -c "import json,sys; d=json.loads(open('/tmp/abprof$v.json').read().strip().splitlines()[-1]); print('   bench', d['value'], d['ms_per_step'])"
done
