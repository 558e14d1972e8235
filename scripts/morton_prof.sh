#!/bin/bash
# per-kernel times (and LDS counters of the count / scatter kernels) of a workload in the given and the Morton order: scripts/morton_prof.sh cfg3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
W=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp
for order in given morton; do
  rm -rf /tmp/mk
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk -- python3 $R/scripts/morton_kernels.py $W $order > /dev/null 2>/tmp/mk.err || { tail -3 /tmp/mk.err; continue; }
  echo "== $W $order: kernel averages, us"
  python3 $R/scripts/kstats.py $(find /tmp/mk -name "*kernel_stats.csv" | head -1) 20
  for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    rm -rf /tmp/mk
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/mk -- python3 $R/scripts/morton_kernels.py $W $order 40 > /dev/null 2>/tmp/mk.err || { tail -3 /tmp/mk.err; continue; }
    python3 $R/scripts/pmc_summary.py $(find /tmp/mk -name "*counter_collection.csv" | head -1) | grep "k_project_hist\|k_isect_scatter"
  done
done
