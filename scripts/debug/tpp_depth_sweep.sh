#!/bin/bash
# lane-per-problem backward kernel: one vs two steps of prefetch distance (TRAJOPT_TPP_DEPTH), dense cartpole shape
set -o pipefail
for N in 8192 16384 32768 65536 131072 262144; do
  for D in 1 2; do
    echo "== N=$N depth=$D"
    TRAJOPT_TPP_DEPTH=$D timeout -k 10 120 python scripts/bench_backward.py --n 4 --m 1 --T 500 --N $N --kernel 3 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
for N in 32768 65536; do
  for D in 1 2; do
    echo "== (2,1) N=$N depth=$D"
    TRAJOPT_TPP_DEPTH=$D timeout -k 10 120 python scripts/bench_backward.py --n 2 --m 1 --T 500 --N $N --kernel 3 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
