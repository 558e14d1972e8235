#!/bin/bash
# PMC passes on the frame's kernels (separate passes, --kernel-trace + --pmc only): scripts/pmc_raster.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmcr
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcr -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2> /tmp/pmcr.err || { tail -5 /tmp/pmcr.err; continue; }
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmcr -name "*counter_collection.csv" | head -1) | grep "k_rasterize_fwd\|k_project_hist\|k_isect_scatter"
done
