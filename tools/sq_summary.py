#!/usr/bin/env python3
"""Mean of each rocprofv3 --pmc counter per (kernel, grid size) from *_counter_collection.csv files.
    python tools/sq_summary.py <counter_collection.csv>... [--match substr]"""
import collections
import csv
import sys


def main():
    files = [a for a in sys.argv[1:] if a.endswith(".csv")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("azsp::", "")
                if match and match not in name:
                    continue
                key = (name[:70], int(r["Grid_Size"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        n = max(len(v) for v in acc[key].values())
        print(f"{key[0]} grid={key[1]} dispatches={n} mean_ns={sum(dur[key]) / len(dur[key]):.0f}")
        for k in sorted(c):
            print(f"    {k:28s} {c[k]:16.1f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"]:
            print(f"    MFMA busy / CU busy cycles   {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CU_CYCLES']:.3f}  (4 SIMDs per CU: /4 = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CU_CYCLES'] / 4:.3f} of the matrix pipes)")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in c:
                    print(f"    {k} / SQ_WAVE_CYCLES  {c[k] / c['SQ_WAVE_CYCLES']:.3f}")


if __name__ == "__main__":
    main()
