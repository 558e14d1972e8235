#!/bin/bash
# forward + backward step of config 3 under library variants (MS_VARIANT builds): scripts/ab_bwd_variants.sh base _noatom ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  [ "$v" = "base" ] && v=""
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
  rm -rf /tmp/abb
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abb -- python3 $R/scripts/bwd_probe.py > /tmp/abb.json 2>/tmp/abb.err
  f=$(find /tmp/abb -name "*kernel_stats.csv" | head -1)
  echo "variant '$v': $(python3 $R/scripts/kstats.py $f 10 | tr ',' '\n' | grep 'bwd\|memset\|fwd<3,float,true' | tr '\n' ' ')"
  tail -1 /tmp/abb.json
done
