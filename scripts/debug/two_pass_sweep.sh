#!/bin/bash
# Developer aid: iteration time with the one-launch (0) and the two-pass (2) line search over batch sizes.
set -e
out=gpurun_out/two_pass_sweep.txt
: > $out
for cfg in "cartpole 16384" "cartpole 24576" "cartpole 32768" "cartpole 49152" "cartpole 65536" "double_cartpole 16384" "double_cartpole 32768" "pendulum 65536"; do
  set -- $cfg
  for mode in 0 2; do
    timeout -k 10 150 python bench.py --env $1 --instances $2 --steps 5 --warmup 2 --no-cpu-baseline --staged $mode > gpurun_out/sweep_tmp.json 2> gpurun_out/sweep_tmp.err
    python - "$1" "$2" "$mode" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep_tmp.json"))
print(sys.argv[1], sys.argv[2], "mode", sys.argv[3], "plan", d["plan"]["staged_line_search"], "ms %.3f" % d["ms_per_step"],
      "rollout %.3f" % d["phase_ms"]["rollout"], "select %.3f" % d["phase_ms"]["select_commit"], "ret %.9f" % d["mean_return"])
PY
  done
done
cat $out
