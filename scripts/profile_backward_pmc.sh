#!/bin/bash
# HBM-side counters of the backward pass alone on dense inputs: 65 536 cartpoles / double cartpoles,
# full-API and gains-only launches.  One rocprofv3 --pmc run per pass (block limits: MI355X guide).  From the repo root.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_bwd
mkdir -p $O
run() {  # name, counters..., -- n m N
  name=$1; shift
  ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 150 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/$name -o k -- python scripts/bench_backward.py --n $1 --m $2 --T 500 --N $3 --kernel 3 --reps 2 > $O/$name.log 2>&1 || exit 1
}
for shape in "4 1 65536 cp" "6 1 65536 dcp"; do
  set -- $shape
  run fetch_$4 FETCH_SIZE -- $1 $2 $3
  run write_$4 WRITE_SIZE TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum -- $1 $2 $3
  run tcc_$4 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -- $1 $2 $3
  run tcp_$4 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -- $1 $2 $3
  run tcp2_$4 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum -- $1 $2 $3
  python scripts/pmc_reduce.py $O/r03_pmc_backward_$4.json "backward pass alone, dense inputs, dx=$1 du=$2 T=500 N=$3 (scripts/bench_backward.py --kernel 3), full-API and gains-only launches" $(find $O/fetch_$4 $O/write_$4 $O/tcc_$4 $O/tcp_$4 $O/tcp2_$4 -name "*counter_collection.csv") > $O/reduce_$4.log || exit 1
done
echo done
