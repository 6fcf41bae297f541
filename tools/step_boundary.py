"""Kernels around the step (generate_episodes call) boundaries of a rocprofv3 --kernel-trace csv:
    python tools/step_boundary.py <kernel_trace.csv>"""
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("void ","").replace("azsp::","").split("(")[0][:30]))
rows.sort()
# find the largest gap between consecutive dispatches (any queue) after the first 20 %
t0=rows[0][0]; best=None
ends=0
for i in range(1,len(rows)):
    ends=max(ends, rows[i-1][1])
    g=rows[i][0]-ends
    if rows[i][0] > t0 + 0.2*(rows[-1][1]-t0) and (best is None or g>best[0]): best=(g,i)
g,i=best
print("largest idle gap %.1f ms before dispatch %d"%(g/1e6,i))
for r in rows[i-6:i+12]:
    print(f"{(r[0]-rows[i][0])/1e3:10.1f} us dur {(r[1]-r[0])/1e3:8.1f} q{r[2]} {r[3]}")
