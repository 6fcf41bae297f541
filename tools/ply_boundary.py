"""Kernels around the ply boundaries of a rocprofv3 --kernel-trace csv (what runs between two searches):
    python tools/ply_boundary.py <kernel_trace.csv>"""
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].replace("void ","").replace("azsp::","").split("(")[0][:28]))
rows.sort()
fin=[i for i,r in enumerate(rows) if r[3].startswith("k_finish")]
i=fin[len(fin)//2]
t0=rows[i][0]
for r in rows:
    if t0-3.0e6 < r[0] < t0+1.5e6 and not (r[3].startswith("k_resnet") and False):
        print(f"{(r[0]-t0)/1e3:9.1f} us  dur {(r[1]-r[0])/1e3:7.1f}  q{r[2]}  {r[3]}")
