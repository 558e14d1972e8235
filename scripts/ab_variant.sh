#!/bin/bash
# library variants under rocprofv3 on one workload: scripts/ab_variant.sh cfg4 base _name ... (MS_VARIANT builds)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
c=$1; shift
for v in "$@"; do
  [ "$v" = "base" ] && v=""
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
  rm -rf /tmp/abv
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abv -- python3 $R/bench.py --workload $c --steps 60 --no-cpu-baseline --no-verify --no-extras > /tmp/abv.json 2>/tmp/abv.err
  f=$(find /tmp/abv -name "*kernel_stats.csv" | head -1)
  echo "$c variant '$v': $(python3 $R/scripts/kstats.py $f 60)"
done
