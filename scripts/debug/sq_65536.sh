#!/bin/bash
# Developer aid: issue counters and kernel trace of the iteration at 65 536 cartpoles; staged vs two-pass at 131 072.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/sq65k
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o k -- python bench.py --no-cpu-baseline --instances 65536 --steps 3 --warmup 2 > $O/pmc_sq.log 2>&1 || exit 1
python scripts/pmc_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -n 1) $O/sq.json "65536 cartpoles" > $O/sq.txt 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python bench.py --no-cpu-baseline --instances 65536 --steps 5 --warmup 2 > $O/kt.log 2>&1 || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv
for mode in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --instances 131072 --steps 5 --warmup 2 --staged $mode > $O/b131k_$mode.json 2> $O/b131k_$mode.err || exit 1
done
echo done
