#!/bin/bash
set -o pipefail
for v in base ntnt stnt ldnt; do
  for shape in "4 1" "6 1"; do
    set -- $shape
    echo "== $v n=$1"
    TRAJOPT_AMD_SO=$PWD/trajopt_amd/libtrajopt_amd_$v.so timeout -k 10 120 python scripts/bench_backward.py --n $1 --m $2 --T 500 --N 65536 --kernel 3 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
