#!/bin/bash
# Round 6, review item 1: the blend loop with log2(alpha) from the matrix pipe, in isolation (scripts/ubench/raster_mfma.hip):
# timings at 8 and 6 resident waves per SIMD, then PMC passes (separate passes, --kernel-trace + --pmc only) for the
# matrix / vector pipe co-execution.  Output: gpurun_out/raster_mfma_probe.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/raster_mfma_probe.txt
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
{
for w in 8 6; do
  echo "== timings, launch bound $w waves per SIMD"
  $R/scripts/ubench/raster_mfma_w$w 2000
done
for set in "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  rm -rf /tmp/pmcm
  echo "== pmc: $set"
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcm -- $R/scripts/ubench/raster_mfma_w8 400 > /dev/null 2> /tmp/pmcm.err || { tail -3 /tmp/pmcm.err; continue; }
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmcm -name "*counter_collection.csv" | head -1) 1 max
done
} > $OUT 2>&1
tail -60 $OUT
