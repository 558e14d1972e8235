#!/bin/bash
# PMC passes on the backward rasteriser of config 3 (separate passes, --kernel-trace + --pmc only): scripts/pmc_bwd.sh [kernel-regex]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
K=${1:-k_rasterize_bwd}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LEVEL_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/pmcb
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcb -- python3 $R/scripts/bwd_probe.py > /dev/null 2> /tmp/pmcb.err || { echo "set failed: $set"; tail -3 /tmp/pmcb.err; continue; }
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmcb -name "*counter_collection.csv" | head -1) | grep "$K"
done
