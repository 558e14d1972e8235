#!/bin/bash
# a few PMC passes of the backward rasteriser under a library variant: scripts/pmc_bwd_variant.sh _qabl2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
v=$1; [ "$v" = "base" ] && v=""
export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmcv
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcv -- python3 $R/scripts/bwd_probe.py > /dev/null 2> /tmp/pmcv.err || { echo "set failed: $set"; tail -3 /tmp/pmcv.err; continue; }
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmcv -name "*counter_collection.csv" | head -1) | grep "k_rasterize_bwd_"
done
