"""Where does the GPU idle?  rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) -> per hardware queue: launches, busy time, the gaps
between consecutive kernels; over all queues: the union of busy intervals (GPU busy), the idle time, and the kernels that FOLLOW the
longest idle intervals.  Written for the open observation of round 3: with one encode lane and the decode stream at
GPU_MAX_HW_QUEUES=8 a bench step takes 1034 instead of 709 ms although the kernels' own durations are unchanged.
usage: python tools/hwq_gaps.py <kernel_trace.csv> [skip_first_ms=0]  > summary.json     (no GPU needed; unit test: tests/test_host_cpu.py)"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    return n.split("(")[0].split("<")[0].strip()[:60]


def summarise(rows, skip_ns=0):
    """rows: dicts with Kernel_Name, Start_Timestamp, End_Timestamp and (if the trace has them) Queue_Id / Stream_Id."""
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, str(r.get("Queue_Id", "?")), str(r.get("Stream_Id", "?")), family(r["Kernel_Name"])))
    if not ev:
        return {"launches": 0}
    ev.sort()
    t0 = ev[0][0] + skip_ns
    ev = [x for x in ev if x[0] >= t0]
    span = max(e for _, e, *_ in ev) - ev[0][0]
    # union of busy intervals over all queues
    busy, idle_iv, cur_s, cur_e = 0, [], ev[0][0], ev[0][1]
    for s, e, q, st, fam in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            idle_iv.append((s - cur_e, fam, q))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per_q = defaultdict(lambda: {"launches": 0, "busy_ms": 0.0, "gap_ms": 0.0, "gaps_over_50us": 0, "streams": set(), "last_end": None})
    for s, e, q, st, fam in ev:
        d = per_q[q]
        d["launches"] += 1
        d["busy_ms"] += (e - s) / 1e6
        d["streams"].add(st)
        if d["last_end"] is not None and s > d["last_end"]:
            d["gap_ms"] += (s - d["last_end"]) / 1e6
            d["gaps_over_50us"] += (s - d["last_end"]) > 50_000
        d["last_end"] = max(d["last_end"] or 0, e)
    after = defaultdict(lambda: [0, 0.0])
    for g, fam, q in idle_iv:
        after[fam][0] += 1
        after[fam][1] += g / 1e6
    overlap = sum((e - s) for s, e, *_ in ev) - busy          # kernel time that ran concurrently with another kernel
    return {
        "launches": len(ev), "span_ms": span / 1e6, "gpu_busy_ms": busy / 1e6, "gpu_idle_ms": (span - busy) / 1e6,
        "sum_of_kernel_ms": sum((e - s) for s, e, *_ in ev) / 1e6, "concurrent_kernel_ms": overlap / 1e6,
        "idle_intervals": len(idle_iv), "idle_over_50us": sum(g > 50_000 for g, *_ in idle_iv),
        "idle_ms_in_intervals_over_50us": sum(g for g, *_ in idle_iv if g > 50_000) / 1e6,
        "per_queue": {q: {"launches": d["launches"], "busy_ms": round(d["busy_ms"], 3), "gap_ms": round(d["gap_ms"], 3),
                          "gaps_over_50us": int(d["gaps_over_50us"]), "streams": sorted(d["streams"])} for q, d in sorted(per_q.items())},
        "idle_before_kernel_family_ms": {k: {"intervals": v[0], "idle_ms": round(v[1], 3)}
                                         for k, v in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]},
    }


def main():
    path = sys.argv[1]
    skip_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    with open(path, newline="") as fh:
        out = summarise(csv.DictReader(fh), int(skip_ms * 1e6))
    out["file"] = path
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
