#!/bin/bash
# PMC passes on the binning kernels of one workload (separate passes, --kernel-trace + --pmc only): scripts/pmc_k3.sh cfg4
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
c=${1:-cfg4}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  rm -rf /tmp/pmck
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmck -- python3 $R/bench.py --workload $c --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras > /dev/null 2> /tmp/pmck.err || { echo "set failed: $set"; tail -3 /tmp/pmck.err; continue; }
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmck -name "*counter_collection.csv" | head -1) | grep "k_project_hist\|k_isect_scatter"
done
