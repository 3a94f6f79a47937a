#!/bin/bash
# does the power-of-two row stride (NP * 8 bytes between the elements of a step) cost HBM bandwidth?
set -o pipefail
for N in 65536 65600 66048 61440 69696 57344 49152 50000; do
  echo "== N=$N"
  timeout -k 10 120 python scripts/bench_backward.py --n 4 --m 1 --T 500 --N $N --kernel 3 2>&1 | grep -v amdgpu.ids || exit 1
done
for N in 65536 65600 50000; do
  echo "== (6,1) N=$N"
  timeout -k 10 120 python scripts/bench_backward.py --n 6 --m 1 --T 500 --N $N --kernel 3 2>&1 | grep -v amdgpu.ids || exit 1
done
