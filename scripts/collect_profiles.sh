#!/bin/bash
# The round's profile artefacts, on the GPU box from the repo root (outputs under gpurun_out/final_*):
#   the un-profiled bench line (moving-camera and forward+backward legs included; --extras adds config 5 and the
#   multi-view batch), kernel stats of the headline frame alone (`bench.py --no-extras`: the legs run other frames
#   through the same kernels), kernel stats of the forward + backward step, FETCH_SIZE / WRITE_SIZE passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_stats /tmp/fp_fetch /tmp/fp_write /tmp/fp_bwd
python3 $R/bench.py --extras > $O/final_bench.json 2> $O/final_bench.err && echo "bench ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_stats -- python3 $R/bench.py --no-extras > $O/final_bench_under_rocprof.json 2> $O/final_prof.err &&
cp $(find /tmp/fp_stats -name "*kernel_stats.csv" | head -1) $O/final_kernel_stats.csv && echo "stats ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_bwd -- python3 $R/scripts/bwd_probe.py > $O/final_bwd_probe.json 2> $O/final_bwd.err &&
cp $(find /tmp/fp_bwd -name "*kernel_stats.csv" | head -1) $O/final_bwd_kernel_stats.csv && echo "bwd stats ok" &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fp_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras > /dev/null 2> $O/final_pmc_fetch.err &&
python3 $R/scripts/pmc_summary.py $(find /tmp/fp_fetch -name "*counter_collection.csv" | head -1) > $O/final_pmc_fetch.txt && echo "fetch ok" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/fp_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras > /dev/null 2> $O/final_pmc_write.err &&
python3 $R/scripts/pmc_summary.py $(find /tmp/fp_write -name "*counter_collection.csv" | head -1) > $O/final_pmc_write.txt && echo "write ok" &&
python3 $R/scripts/write_traffic.py measure $O/final_pmc_fetch.txt $O/final_pmc_write.txt $O/final_traffic.json && echo "traffic ok" &&
for c in cfg4 cfg5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fp_big
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/fp_big -- python3 $R/bench.py --workload $c --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-extras > /dev/null 2> $O/final_pmc_$c.err &&
    python3 $R/scripts/pmc_summary.py $(find /tmp/fp_big -name "*counter_collection.csv" | head -1) > $O/final_pmc_${ctr}_$c.txt
  done
  python3 $R/scripts/pmc_frame_total.py $O/final_pmc_FETCH_SIZE_$c.txt $O/final_pmc_WRITE_SIZE_$c.txt > $O/final_pmc_fetch_write_$c.txt && echo "$c pmc ok"
done
