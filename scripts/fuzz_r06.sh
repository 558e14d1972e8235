#!/bin/bash
# Round 6, final tree: the fused path's fuzzers on seeds the suite does not run (claimed rows ON: the default)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
O=$R/gpurun_out/fuzz_r06.txt
{
echo "# Round 6, final tree: the fused path's fuzzers on seeds the suite does not run"
echo "# commands, in order: fuzz_more.py 9000 400 | fuzz_cut.py 9500 150 | fuzz_cut.py 9700 60 dense | fuzz_cut.py 9800 100 band | fuzz_cut.py 9900 40 dense band | fuzz_cut.py 10000 100 pipelined | fuzz_cut.py 10100 40 dense pipelined"
python3 scripts/fuzz_more.py 9000 400 2>/dev/null | tail -2
python3 scripts/fuzz_cut.py 9500 150 2>/dev/null | tail -1
python3 scripts/fuzz_cut.py 9700 60 dense 2>/dev/null | tail -1
python3 scripts/fuzz_cut.py 9800 100 band 2>/dev/null | tail -1
python3 scripts/fuzz_cut.py 9900 40 dense band 2>/dev/null | tail -1
python3 scripts/fuzz_cut.py 10000 100 pipelined 2>/dev/null | tail -1
python3 scripts/fuzz_cut.py 10100 40 dense pipelined 2>/dev/null | tail -1
} > $O 2>&1
cat $O
