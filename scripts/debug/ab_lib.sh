#!/bin/bash
# A/B of two builds of the library on one box: gpurun_tmp/libtrajopt_old.so against the in-tree one, alternating.
# usage: ab_lib.sh "<bench_backward args>" ...
for cfg in "$@"; do
  for rep in 1 2 3; do
    for which in old new; do
      if [ $which = old ]; then export TRAJOPT_AMD_SO=$PWD/gpurun_tmp/libtrajopt_old.so; else unset TRAJOPT_AMD_SO; fi
      echo "cfg=$cfg lib=$which"
      timeout -k 10 120 python scripts/bench_backward.py $cfg 2>&1 | grep -E "^(ilqr|riccati)" || exit 1
    done
  done
done
