#!/bin/bash
# process-to-process variance of the sub-batched C4 line (bench.py --sub-batches 4) on one box
for rep in 1 2 3 4; do
  timeout -k 10 120 python bench.py --no-configs --no-cpu-baseline --env quadrotor --instances 4096 --alphas 8 --max-lmbda 1e6 --steps 8 --warmup 3 --sub-batches 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sub-batches', d.get('sub_batches'), 'ms', round(d['ms_per_step'],3), 'one batch', round(d['one_batch_ms_per_step'],3))" || exit 1
done
