#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into profiles/pmc_traffic.json.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Per kernel: mean KiB per dispatch of each counter and `bytes_per_launch` = (2*FETCH_SIZE + WRITE_SIZE)*1024
(gfx950 correction from MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half of wide coalesced reads)."""
import collections
import csv
import json
import sys


def means(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
                acc[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    if len(sys.argv) not in (3, 4) or (len(sys.argv) == 4 and not sys.argv[3].endswith(".json")):
        sys.exit(__doc__)           # (a shell glob that matches two csv files per pass once overwrote a counter file)
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out[k] = {"fetch_kib": f, "write_kib": w, "dispatches": max(nf, nw),
                  "bytes_per_launch": (2 * f + w) * 1024}
    path = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    # fingerprint of the kernel sources the passes ran on: bench.py reports the traffic only while it matches the tree
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import kernel_source_sha
        out["_source_sha16"] = kernel_source_sha()
    except Exception as exc:           # noqa: BLE001  (torch missing etc.: the summary is still useful)
        out["_source_sha16"] = None
        print("no source fingerprint:", exc)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        if not isinstance(v, dict):
            continue
        print(f"{k:28s} fetch {v['fetch_kib']:10.1f} KiB  write {v['write_kib']:10.1f} KiB  -> {v['bytes_per_launch'] / 1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
