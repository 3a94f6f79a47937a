"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run, as the MI355X guide's HBM
section prescribes) into per-kernel HBM bytes per launch -> profiles/*_pmc_traffic_*.json (bench.py reads it for
`roofline.traffic`).  FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 streaming reads are corrected x2 (same section).

  python scripts/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json "workload text"
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import _lib  # noqa: E402  (source_stamp only)


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            if not k.startswith("void trajopt::") and not k.startswith("trajopt::"):
                continue
            k = k.replace("void ", "").split("(")[0]
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] * 1024.0 for k in tot}


def main():
    fetch, write, out, workload = sys.argv[1:5]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        kernels[k] = {"fetch_size_raw": f.get(k, 0.0), "write_size": w.get(k, 0.0),
                      "hbm_bytes": 2.0 * f.get(k, 0.0) + w.get(k, 0.0)}
    phase = {}
    for k, v in kernels.items():
        for ph, tag in (("rollout", "rollout_kernel"), ("quadratize", "quadratize_kernel"), ("backward", "bwd_"),
                        ("backward", "fused_qb4_kernel"), ("select_commit", "select_kernel"), ("select_commit", "publish_kernel")):
            if tag in k:
                phase[ph] = phase.get(ph, 0.0) + v["hbm_bytes"]
    json.dump({"stamp": _lib.source_stamp(), "units": "bytes per launch (mean over the run's launches); FETCH_SIZE/WRITE_SIZE are reported in KiB by "
                        "rocprofv3; reads corrected x2 as MI355X_MICROARCH.md (HBM section) prescribes for gfx950",
               "workload": workload, "kernels": kernels, "by_phase": phase}, open(out, "w"), indent=1)
    print(json.dumps(phase))


if __name__ == "__main__":
    main()
