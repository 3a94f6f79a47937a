#!/bin/bash
# After scripts/profile_round.sh ran on the GPU box and gpurun_out/prof came back: reduce the PMC passes and copy the
# summaries into profiles/ under the tag given as $1 (e.g. r02_b).  Runs on the CPU.
set -e
TAG=${1:-r02_b}
O=gpurun_out/prof
W1="$(python -c "import json;print(json.load(open('profiles/r02_pmc_traffic_c2.json'))['workload'])")"
W5="$(python -c "import json;print(json.load(open('profiles/r02_pmc_mfma_c5.json'))['workload'])")"
python scripts/pmc_traffic.py $O/pmc_fetch/c2_counter_collection.csv $O/pmc_write/c2_counter_collection.csv profiles/r02_pmc_traffic_c2.json "$W1" > /dev/null
python scripts/pmc_sq.py $O/pmc_sq/c2_counter_collection.csv profiles/r02_pmc_sq_c2.json "$W1" > /dev/null
python scripts/pmc_sq.py $O/pmc_c5/c5_counter_collection.csv profiles/r02_pmc_mfma_c5.json "$W5" > /dev/null
cp $O/kt/c2_kernel_stats.csv profiles/${TAG}_bench_c2_kernel_stats.csv
cp $O/bench_65536.json profiles/${TAG}_bench_65536_instances.json
cp $O/bench_65536_20steps.json profiles/${TAG}_bench_65536_instances_20steps.json
python scripts/pmc_sq.py $O/pmc_sq_65536/k_counter_collection.csv profiles/r02_pmc_sq_65536.json "bench.py --instances 65536 --steps 3 --warmup 2 (65 536 cartpoles, T=500, 11 alphas)" > /dev/null
cp $O/kt_65536/k_kernel_stats.csv profiles/${TAG}_bench_65536_kernel_stats.csv
cp $O/bench_16384.json profiles/${TAG}_bench_16384_instances.json
cp $O/bench_32768.json profiles/${TAG}_bench_32768_instances.json
cp $O/bench_c3.json profiles/${TAG}_bench_c3_8192.json
cp $O/bench_c4.json profiles/${TAG}_bench_c4_4096.json
cp $O/bench_c1.json profiles/${TAG}_bench_c1.json
grep -v amdgpu.ids $O/backward_all_configs.log > profiles/${TAG}_backward_all_configs.log
cp $O/gps_bench.json profiles/${TAG}_gps_backward.json
cp $O/bsp_bench.json profiles/${TAG}_bsp_backward.json
python - <<'PY'
import json
for f in ['bench_c2', 'bench_16384', 'bench_32768', 'bench_65536', 'bench_65536_20steps', 'bench_c3', 'bench_c4']:
    d = json.load(open('gpurun_out/prof/%s.json' % f))
    print(f, round(d['ms_per_step'], 3), round(d['value'] / 1e9, 3), {k: round(v, 3) for k, v in d['phase_ms'].items()}, d['source_stamp'])
print(json.load(open('gpurun_out/prof/bench_c1.json')))
PY
