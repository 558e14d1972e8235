#!/bin/bash
# Round 6's extra artefacts (outputs under gpurun_out/final6_*), after scripts/collect_profiles.sh: band rehearsals on the
# prepared scene, the config sweep, host overhead, the backward step's kernel stats at configs 4 / 5 sizes.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd $R
python3 scripts/band_bench.py --workload cfg5 --frames 150 --order prepared > $O/final6_band_cfg5_prepared.jsonl 2> $O/final6_band.err && echo "band cfg5 ok"
python3 scripts/band_bench.py --workload cfg3 --frames 150 --order prepared > $O/final6_band_cfg3_prepared.jsonl 2>> $O/final6_band.err && echo "band cfg3 ok"
python3 scripts/config_sweep.py > $O/final6_config_sweep.jsonl 2> $O/final6_sweep.err && echo "sweep ok"
python3 scripts/host_overhead.py > $O/final6_host_overhead.txt 2>&1 && echo "host ok"
python3 scripts/bwd_err.py > $O/final6_bwd_err.json 2> $O/final6_bwd_err.err && echo "bwd err ok"
