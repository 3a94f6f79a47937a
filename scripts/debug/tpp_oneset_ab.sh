#!/bin/bash
# dx = 6 dense backward pass: one input set re-loaded inside the step (TRAJOPT_TPP_DEPTH=0) against two sets (=1), alternating
for N in 32768 65536 131072; do
  for rep in 1 2 3; do
    for d in 0 1; do
      echo "N=$N depth=$d"
      TRAJOPT_TPP_DEPTH=$d timeout -k 10 120 python scripts/bench_backward.py --n 6 --m 1 --T 500 --N $N --layout 1 --reps 10 --kernel 3 2>&1 | grep "^ilqr" || exit 1
    done
  done
done
