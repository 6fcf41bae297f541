#!/usr/bin/env python3
"""Where a simulation round's time goes, from a rocprofv3 --kernel-trace csv (per-dispatch start / end timestamps):
per queue (= engine stream), the busy time per kernel and the idle gaps between consecutive dispatches.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/notiming.py --cache --steps 1
    python tools/trace_gaps.py out/*/*_kernel_trace.csv [--skip-first 0.5]"""
import collections
import csv
import sys


def short(name):
    n = name.replace("void ", "").replace("azsp::", "").split("(")[0]
    return n[:60]


def main():
    path = sys.argv[1]
    skip = float(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0.5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip                       # (the first part of the run is warm-up)
    rows = [r for r in rows if r[0] >= cut]
    span = (rows[-1][1] - rows[0][0]) / 1e3
    print(f"{len(rows)} dispatches over {span / 1e3:.1f} ms")
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        if len(rs) < 50:
            continue
        busy = collections.defaultdict(lambda: [0, 0.0])
        gap_after = collections.defaultdict(lambda: [0, 0.0])
        for i, (s, e, _, k) in enumerate(rs):
            busy[k][0] += 1
            busy[k][1] += (e - s) / 1e3
            if i + 1 < len(rs):
                g = (rs[i + 1][0] - e) / 1e3
                gap_after[(k, rs[i + 1][3])][0] += 1
                gap_after[(k, rs[i + 1][3])][1] += g
        qspan = (rs[-1][1] - rs[0][0]) / 1e3
        print(f"-- queue {q}: {len(rs)} dispatches, span {qspan / 1e3:.1f} ms")
        for k, (n, us) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   busy  {us / qspan * 100:5.1f} %  {n:6d} x {us / n:8.1f} us  {k}")
        for (a, b), (n, us) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   gap   {us / qspan * 100:5.1f} %  {n:6d} x {us / n:8.1f} us  {a[:36]} -> {b[:36]}")


if __name__ == "__main__":
    main()
