"""Kernel overlap of a rocprofv3 --kernel-trace run: per queue the kernels' summed duration, and over all queues the time at
least one kernel was running (union) against the sum - how much of the sub-batches' work ran side by side.
  python scripts/overlap_summary.py <kernel_trace.csv> <out.json> "<workload>"
The window is the sub-batched part of the run: from the first library kernel on a second queue to the end."""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), str(r.get("Queue_Id", "?")), r["Kernel_Name"]) for r in rows
            if "trajopt::" in r["Kernel_Name"])
# the sub-batched part of the run: from the first library kernel on a queue other than the first one's to the end
main_q = ev[0][2]
cut = min(s for s, _, q, _ in ev if q != main_q)
ev = [e for e in ev if e[0] >= cut]
per_queue = {}
for s, e, q, _ in ev:
    d = per_queue.setdefault(q, [0, 0])
    d[0] += 1
    d[1] += e - s
total = sum(e - s for s, e, _, _ in ev)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
out = {"workload": sys.argv[3], "window": "from the first trajopt kernel on a second hardware queue to the end of the trace (warm-up + timed iterations of the sub-batched run)", "kernels": len(ev),
       "queues": {q: {"kernels": n, "busy_ms": d / 1e6} for q, (n, d) in sorted(per_queue.items())},
       "sum_of_kernel_durations_ms": total / 1e6, "time_with_a_kernel_running_ms": union / 1e6,
       "mean_kernels_in_flight": total / union}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
