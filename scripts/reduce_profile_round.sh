#!/bin/bash
# After scripts/profile_round.sh ran on the GPU box and gpurun_out/prof came back: reduce the PMC passes and copy the
# summaries into profiles/ under the tag given as $1 (e.g. r03_b).  Runs on the CPU.
set -e
TAG=${1:-r03_b}
O=gpurun_out/prof
W1="bench.py --no-configs --no-cpu-baseline --steps 5 --warmup 2 (C2: 4096 cartpoles, T=500, 11 alphas)"
W5="scripts/bench_backward.py --variant riccati --n 64 --m 16 --T 1000 --N 1024 --layout 0 --reps 2 (C5)"
python scripts/pmc_traffic.py $O/pmc_fetch/c2_counter_collection.csv $O/pmc_write/c2_counter_collection.csv profiles/r03_pmc_traffic_c2.json "$W1" > /dev/null
python scripts/pmc_sq.py $O/pmc_sq/c2_counter_collection.csv profiles/r03_pmc_sq_c2.json "$W1" > /dev/null
python scripts/pmc_sq.py $O/pmc_c5/c5_counter_collection.csv profiles/r03_pmc_mfma_c5.json "$W5" > /dev/null
python scripts/pmc_sq.py $O/pmc_sq_65536/k_counter_collection.csv profiles/r03_pmc_sq_65536.json "bench.py --no-configs --instances 65536 --steps 3 --warmup 2 (65 536 cartpoles, T=500, 11 alphas)" > /dev/null
cp $O/bench_all.json profiles/${TAG}_bench_all_configs.json
cp $O/kt/c2_kernel_stats.csv profiles/${TAG}_bench_c2_kernel_stats.csv
cp $O/kt_65536/k_kernel_stats.csv profiles/${TAG}_bench_65536_kernel_stats.csv
cp $O/kt_c3/k_kernel_stats.csv profiles/${TAG}_bench_c3_kernel_stats.csv
cp $O/kt_c4/k_kernel_stats.csv profiles/${TAG}_bench_c4_kernel_stats.csv
cp $O/kt_c5/c5_kernel_stats.csv profiles/${TAG}_riccati_c5_kernel_stats.csv
cp $O/bench_c1.json profiles/${TAG}_bench_c1.json
cp $O/mpc.json profiles/${TAG}_mpc_control_steps.json
grep -v amdgpu.ids $O/backward_all_configs.log > profiles/${TAG}_backward_all_configs.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/prof/bench_all.json').read().strip().splitlines()[-1])
print('C2', round(d['ms_per_step'], 4), round(d['value'] / 1e9, 3), d['roofline']['bound'], round(d['roofline']['frac'], 3), d['source_stamp'])
for k, v in d['configs'].items():
    if 'ms' in v:
        print(k, v['ms'], v['value'], v['roofline'].get('frac'))
print(json.load(open('gpurun_out/prof/bench_c1.json')))
PY
