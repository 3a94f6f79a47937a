#!/bin/bash
# One GPU call's worth of measurements for profiles/ (round 3): the bench line with every config, kernel traces of the
# headline and of the 65 536-instance line, the PMC passes (one rocprofv3 --pmc run per pass: HBM traffic, issue counters,
# matrix-core counters of C5), C1, the backward pass alone on C1-C5, MPC.  Run from the repo root on the GPU box.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_all.json 2> $O/bench_all.err || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c2 -- python bench.py --no-configs --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c2 -- python bench.py --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c2 -- python bench.py --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_write.log 2>&1 || exit 1
# issue counters of the same command (8 SQ + 1 GRBM: one pass)
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o c2 -- python bench.py --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1 || exit 1
# C5: matrix-core counters and the kernel trace of the Riccati sweep
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c5 -o c5 -- python scripts/bench_backward.py --variant riccati --n 64 --m 16 --T 1000 --N 1024 --layout 0 --reps 2 > $O/pmc_c5.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -o c5 -- python scripts/bench_backward.py --variant riccati --n 64 --m 16 --T 1000 --N 1024 --layout 0 --reps 3 > $O/kt_c5.log 2>&1 || exit 1
PYTHONPATH=. timeout -k 10 200 python scripts/bench_c1.py $O/bench_c1.json > $O/bench_c1.log 2>&1 || exit 1
{
  timeout -k 10 60 python scripts/bench_backward.py --n 2 --m 1 --T 100 --N 1 || exit 1
  for N in 4096 65536; do timeout -k 10 60 python scripts/bench_backward.py --n 4 --m 1 --T 500 --N $N || exit 1; done
  for N in 8192 65536; do timeout -k 10 60 python scripts/bench_backward.py --n 6 --m 1 --T 500 --N $N || exit 1; done
  timeout -k 10 60 python scripts/bench_backward.py --n 12 --m 4 --T 500 --N 4096 --layout 0 || exit 1
  timeout -k 10 120 python scripts/bench_backward.py --variant riccati --n 64 --m 16 --T 1000 --N 1024 --layout 0 --reps 3 || exit 1
} > $O/backward_all_configs.log 2>&1
# the 65 536-instance line: issue counters and kernel trace (two-pass line search: returns-only rollout, select, the accepted alpha again)
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_65536 -o k -- python bench.py --no-configs --no-cpu-baseline --instances 65536 --steps 3 --warmup 2 > $O/pmc_sq_65536.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_65536 -o k -- python bench.py --no-configs --no-cpu-baseline --instances 65536 --steps 5 --warmup 2 > $O/kt_65536.log 2>&1 || exit 1
# kernel traces of the C3 and C4 lines as the bench's configs block runs them
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -o k -- python bench.py --no-configs --no-cpu-baseline --env double_cartpole --instances 8192 --steps 10 --warmup 3 > $O/kt_c3.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o k -- python bench.py --no-configs --no-cpu-baseline --env quadrotor --instances 4096 --alphas 8 --steps 8 --warmup 3 > $O/kt_c4.log 2>&1 || exit 1
PYTHONPATH=. timeout -k 10 250 python scripts/bench_mpc.py $O/mpc.json > $O/mpc.log 2>&1 || exit 1
echo done
