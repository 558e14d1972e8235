#!/bin/bash
# k_band_precull under build variants: scripts/precull_variants.sh cfg5 8 3 sub8 sub12
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
cfg=$1; world=$2; rank=$3; shift 3
for v in base "$@"; do
  lib=$R/mojosplat_amd/csrc/libmojosplat_hip.so; [ "$v" != base ] && lib=$R/mojosplat_amd/csrc/libmojosplat_hip_$v.so
  for order in given prepared; do
    rm -rf /tmp/bk
    MOJOSPLAT_HIP_LIB=$lib SCENE_ORDER=$order timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python3 $R/scripts/band_profile.py $cfg $world $rank > /dev/null 2>/tmp/bk.err || { tail -3 /tmp/bk.err; continue; }
    echo "== $v $cfg world $world rank $rank, $order"
    python3 $R/scripts/kstats.py $(find /tmp/bk -name "*kernel_stats.csv" | head -1) 20
  done
done
