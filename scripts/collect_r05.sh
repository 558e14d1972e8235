#!/bin/bash
# Round 5's extra artefacts (outputs under gpurun_out/final5_*), after scripts/collect_profiles.sh: band rehearsals in the given
# and the prepared order, the Morton probes, the config sweeps in both orders, the backward's counters, host overhead.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out
cd $R
for o in given prepared; do
  python3 scripts/band_bench.py --workload cfg5 --frames 150 --order $o > $O/final5_band_cfg5_$o.jsonl 2> $O/final5_band.err && echo "band cfg5 $o ok"
  python3 scripts/band_bench.py --workload cfg3 --frames 150 --order $o > $O/final5_band_cfg3_$o.jsonl 2>> $O/final5_band.err && echo "band cfg3 $o ok"
done
for c in cfg3 cfg4 cfg5; do python3 scripts/morton_probe.py $c; done > $O/final5_morton_probe.jsonl 2> $O/final5_morton.err && echo "morton ok"
python3 scripts/config_sweep.py > $O/final5_config_sweep.jsonl 2> $O/final5_sweep.err && echo "sweep ok"
python3 scripts/config_sweep.py --order morton cfg2 cfg3 cfg4 cfg5 > $O/final5_config_sweep_morton.jsonl 2>> $O/final5_sweep.err && echo "sweep morton ok"
python3 scripts/bwd_err.py > $O/final5_bwd_err.json 2> $O/final5_bwd_err.err && echo "bwd err ok"
python3 scripts/host_overhead.py > $O/final5_host_overhead.txt 2>&1 && echo "host ok"
bash scripts/pmc_bwd.sh k_rasterize_bwd_quads > $O/final5_bwd_pmc_raw.txt 2>&1 && echo "bwd pmc ok"
bash scripts/morton_prof.sh cfg3 > $O/final5_morton_prof_cfg3.txt 2>&1 && echo "morton prof ok"
bash scripts/morton_prof.sh cfg5 2>&1 | grep "==\|{" > $O/final5_morton_prof_cfg5.txt && echo "morton prof cfg5 ok"
