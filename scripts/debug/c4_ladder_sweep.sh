#!/bin/bash
# C4 per-iteration backward times for the in-wave retry budgets of the three-launch lambda ladder (bwd_mfma16.hip)
for v in none 0 1 2 3 4; do
  if [ $v = none ]; then export TRAJOPT_NO_LADDER=1; unset TRAJOPT_LADDER_INKERNEL; else unset TRAJOPT_NO_LADDER; export TRAJOPT_LADDER_INKERNEL=$v; fi
  echo "== inkernel=$v"
  timeout -k 10 120 python scripts/debug/c4_phase_by_iter.py 12 1e6 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$7,$8}' | tr '\n' ';'; echo
  timeout -k 10 120 python bench.py --no-configs --no-cpu-baseline --env quadrotor --instances 4096 --alphas 8 --max-lmbda 1e6 --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms', round(d['ms_per_step'],3), 'backward avg launch', d['roofline'].get('avg_launch_ms'))" || exit 1
done
