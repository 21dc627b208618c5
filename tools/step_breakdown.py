#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: finds the period of
the graph-replayed step at the end of the trace and aggregates that one step by kernel name.

    python tools/step_breakdown.py <..._kernel_trace.csv> [top_n]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    period = None
    for p in range(200, min(20000, len(names) // 2)):
        if names[-p:] == names[-2 * p:-p]:
            period = p
            break
    if period is None:
        # multi-stream graphs interleave differently on every replay: match the multiset instead
        for p in range(200, min(20000, len(names) // 3)):
            a = collections.Counter(names[-p:])
            if a == collections.Counter(names[-2 * p:-p]) == collections.Counter(names[-3 * p:-2 * p]):
                period = p
                break
    if period is None:
        print("no periodic step found")
        return
    step = rows[-period:]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += d
        agg[r["Kernel_Name"]][1] += 1
    busy = sum(v[0] for v in agg.values())
    print(f"step: {period} kernels, wall {(t1 - t0) / 1e6:.3f} ms, sum of kernel durations "
          f"{busy / 1e6:.3f} ms")
    print(f"{'ms':>8} {'calls':>6} {'avg us':>8}  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{v[0] / 1e6:8.3f} {v[1]:6d} {v[0] / v[1] / 1e3:8.1f}  {k[:130]}")


if __name__ == "__main__":
    main()
