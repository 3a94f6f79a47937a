"""Merge rocprofv3 --pmc passes (one counter_collection.csv per pass) into per-kernel means per launch.

  python scripts/pmc_reduce.py <out.json> "workload text" <pass1.csv> [<pass2.csv> ...]

Only this library's kernels (trajopt::) are kept.  FETCH_SIZE is reported as collected AND corrected (x2: on gfx950 it
tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md "HBM"); FETCH_SIZE / WRITE_SIZE are in KiB."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import _lib  # noqa: E402  (source_stamp only)


def main():
    out, workload, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot, cnt, dur = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int)), defaultdict(list)
    for path in paths:
        with open(path, newline="") as f:
            seen = set()
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("void ", "").split("(")[0]
                if "trajopt::" not in k:
                    continue
                c = row["Counter_Name"]
                tot[k][c] += float(row["Counter_Value"])
                cnt[k][c] += 1
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    res = {}
    for k in sorted(tot):
        d = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        d["launches_per_pass"] = max(cnt[k].values())
        d["ms_under_the_profiler"] = sum(dur[k]) / max(len(dur[k]), 1)
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes"] = 2.0 * d["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
        if "GRBM_GUI_ACTIVE" in d:
            d["gpu_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
        res[k] = d
    json.dump({"stamp": _lib.source_stamp(), "workload": workload,
               "units": "means per launch; SQ_/TCP_/TA_/TCC_ counters summed over all instances", "kernels": res},
              open(out, "w"), indent=1)
    for k, d in res.items():
        print(k[:100], json.dumps({c: round(v, 1) for c, v in d.items()}))


if __name__ == "__main__":
    main()
