#!/bin/bash
# Round 6, review item 4: dynamic instruction accounting of k_project_hist by ablation builds (MS_VARIANT libraries with
# -DMS_ABL_NOREC / _NOMASK / _NOWALK: measurement builds, their frames are WRONG by construction) -- vector / scalar / LDS
# instructions and bytes per launch from PMC passes (separate passes, --kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export MOJOSPLAT_BIN_PX=32
for v in ${VARIANTS:-"" _norec _nomask _nowalk _norecmaskwalk}; do
  [ "$v" = "base" ] && v=""
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
    rm -rf /tmp/php
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/php -- python3 $R/bench.py --workload cfg3 --steps 20 --no-cpu-baseline --no-verify --no-extras > /dev/null 2> /tmp/php.err
    echo "variant '$v' [$set]"
    python3 $R/scripts/pmc_summary.py $(find /tmp/php -name "*counter_collection.csv" | head -1) | grep "k_project_hist"
  done
done
