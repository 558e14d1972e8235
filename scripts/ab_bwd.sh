#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in "" _nopf "" _nopf; do
  export MOJOSPLAT_HIP_LIB=$R/mojosplat_amd/csrc/libmojosplat_hip$v.so
  echo "variant '$v': $(python3 $R/scripts/bwd_probe.py 2>/dev/null | tail -1)"
done
