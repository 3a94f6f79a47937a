#!/bin/bash
# A/B of two builds of the library on one box through bench.py: gpurun_tmp/libtrajopt_old.so against the in-tree one.
# usage: ab_bench.sh "<bench.py args>" ...   (prints ms_per_step and phase_ms of each run)
for cfg in "$@"; do
  for rep in 1 2; do
    for which in old new; do
      if [ $which = old ]; then export TRAJOPT_AMD_SO=$PWD/gpurun_tmp/libtrajopt_old.so; else unset TRAJOPT_AMD_SO; fi
      timeout -k 10 120 python bench.py --no-configs --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', '$cfg', round(d['ms_per_step'],4), {k: round(v,4) for k,v in (d.get('phase_ms') or {}).items()})" || exit 1
    done
  done
done
