#!/bin/bash
# Round 5, review item 7: the forward rasteriser with log2(alpha) expanded about the quad centre (MS_RASTER_EXPANDED=1,
# libmojosplat_hip_exp.so) against the shipped kernel: frame time, kernel time (rocprofv3), and the oracle check of the
# frame at the suite's eps (bench.py's verification block).  scripts/raster_expanded_probe.sh [cfg3]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
W=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp
for lib in libmojosplat_hip.so libmojosplat_hip_exp.so; do
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/$lib
  echo "== $lib $W"
  python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); o=r['verification'].get('oracle',{})
        print(json.dumps({'ms_per_step':r['ms_per_step'],'ms_mean':r['ms_per_step_mean'],'raster_us':r['roofline']['avg_kernel_us'],'bit_identical_to_stagewise':r['verification']['bit_identical_to_stagewise'],'max_abs_vs_stagewise':r['verification']['max_abs_vs_stagewise'],'oracle':{k:o.get(k) for k in ('px_beyond_1e-4','explained_by_branch_margin','unexplained_px','max_abs','max_abs_where_no_branch_is_close','margin_eps')}}))"
  rm -rf /tmp/rx
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rx -- python3 $R/scripts/morton_kernels.py $W given 150 > /dev/null 2>/tmp/rx.err
  python3 $R/scripts/kstats.py $(find /tmp/rx -name "*kernel_stats.csv" | head -1) 20
done
