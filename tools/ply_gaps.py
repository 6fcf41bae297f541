#!/usr/bin/env python3
"""Idle time of the whole GPU around every ply boundary, from a rocprofv3 --kernel-trace csv: for each k_finish, the time
since the last search kernel (any queue) ended before it, and from the end of k_emit_copy to the next kernel's start.
    python tools/ply_gaps.py out/*/*_kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("azsp::", "").split("(")[0][:24]))
rows.sort()
tot_before = tot_after = 0.0
n = 0
for i, (s, e, k) in enumerate(rows):
    if not k.startswith("k_finish"):
        continue
    last_end = max(r[1] for r in rows[max(0, i - 12):i])
    j = i
    while j < len(rows) and not rows[j][2].startswith("k_emit_copy"):
        j += 1
    nxt = rows[j + 1][0] if j + 1 < len(rows) else rows[j][1]
    before, after = (s - last_end) / 1e3, (nxt - rows[j][1]) / 1e3
    tot_before += max(before, 0)
    tot_after += max(after, 0)
    n += 1
    print(f"ply {n:3d}: idle before k_finish {before:9.1f} us   after k_emit_copy {after:9.1f} us")
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{n} plies: {tot_before / 1e3:.1f} ms idle before k_finish, {tot_after / 1e3:.1f} ms after k_emit_copy, of {span / 1e3:.1f} ms traced")
