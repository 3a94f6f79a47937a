"""Reduce a rocprofv3 --pmc pass of SQ / GRBM counters to per-kernel means per launch.

  python scripts/pmc_sq.py <counter_collection.csv> <out.json> "workload text"

Counters (one pass: 8 SQ + 1 GRBM): SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import _lib  # noqa: E402  (source_stamp only)


def main():
    path, out, workload = sys.argv[1:4]
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].replace("void ", "").split("(")[0]
            if "trajopt::" not in k:
                continue
            c = row["Counter_Name"]
            tot[k][c] += float(row["Counter_Value"])
            cnt[k][c] += 1
    res = {}
    for k in sorted(tot):
        d = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        d["launches"] = max(cnt[k].values())
        if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU"):
            d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0          # shader-clock cycles the launch was busy (per XCD)
            d["gpu_cycles"] = cyc
            d["valu_insts_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (cyc * 1024.0)
        res[k] = d
    by_phase = {}
    for k, d in res.items():
        # (the kernel that does a phase's work: the tiled Riccati sweep, not the generic kernel that only redoes the
        # problems it flags - on a clean batch that one launches, finds nothing to do and leaves)
        for ph, tag, rank in (("rollout", "rollout_kernel", 1), ("quadratize", "quadratize_kernel", 1), ("backward", "bwd_", 1),
                              ("backward", "fused_qb4_kernel", 2), ("backward", "fused_tpp_kernel", 2), ("backward", "ric_mfma64_kernel", 3)):
            if tag in k:
                key = (rank, d.get("launches", 0))
                if key >= by_phase.get(ph, ({}, (0, 0)))[1]:
                    by_phase[ph] = (d, key)
    by_phase = {ph: v[0] for ph, v in by_phase.items()}
    json.dump({"stamp": _lib.source_stamp(), "workload": workload,
               "units": "means per launch; SQ_* summed over all SIMDs/XCDs", "kernels": res,
               "kernels_by_phase": by_phase}, open(out, "w"), indent=1)
    for k, d in res.items():
        print(k, json.dumps({c: round(v, 3) for c, v in d.items()}))


if __name__ == "__main__":
    main()
