#!/bin/bash
# Issue / matrix-core / LDS counters of the C4 iteration's kernels (4096 quadrotors, 8 alphas, max_lmbda as the bench's
# c4_quadrotor line): two rocprofv3 --pmc passes, each a run of its own.  Run from the repo root on the GPU box; reduce with
#   python scripts/pmc_sq.py gpurun_out/prof_c4/pmc_a/k_counter_collection.csv profiles/<tag>_pmc_sq_c4.json "<workload>"
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_c4
mkdir -p $O
CMD="python bench.py --no-configs --no-cpu-baseline --env quadrotor --instances 4096 --alphas 8 --max-lmbda 1e6 --steps 8 --warmup 3"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -o k -- $CMD > $O/pmc_a.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b -o k -- $CMD > $O/pmc_b.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- $CMD > $O/kt.log 2>&1 || exit 1
echo done
