#!/bin/bash
# Kernel stats of configs 4 and 5 with and without the depth cut (outputs under gpurun_out/final_*): run after collect_profiles.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg5; do
  for m in 0 1; do
    export MOJOSPLAT_DEPTH_CUT=$m
    rm -rf /tmp/fp_cut
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_cut -- python3 $R/bench.py --workload $c --steps 100 --no-cpu-baseline --no-verify --no-extras > $O/final_bench_${c}_cut$m.json 2> $O/final_cut.err || { tail -3 $O/final_cut.err; continue; }
    cp $(find /tmp/fp_cut -name "*kernel_stats.csv" | head -1) $O/final_${c}_cut${m}_kernel_stats.csv
    echo "$c cut=$m: $(python3 $R/scripts/kstats.py $O/final_${c}_cut${m}_kernel_stats.csv 100)"
  done
done
