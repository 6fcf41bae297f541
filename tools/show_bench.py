#!/usr/bin/env python3
"""One-line digest of bench.py JSON lines (stdin or files): throughput, evaluations, roofline fractions, launch shape."""
import json
import sys


def show(name, r):
    rl = r.get("roofline") or {}
    tr = r.get("tree") or {}
    print(f"{name:22s} eps/s {r['value']:9.1f} ms/step {r['ms_per_step']:9.1f} evals {r.get('evals')} exec {r.get('evals_executed')} "
          f"frac {rl.get('frac', 0):.3f} alg {rl.get('algorithmic_frac', 0):.3f} wall {rl.get('frac_over_wall', 0):.3f} "
          f"launch {rl.get('avg_launch_ms', 0):.3f} ms x{rl.get('launches')} boards/launch {rl.get('boards_per_launch', 0):.0f} "
          f"conc {rl.get('concurrent_launches', 0):.2f} share {rl.get('infer_share_of_wall', 0):.2f} tree_ms {tr.get('select_ms', 0):.0f} rounds {tr.get('rounds')}")


for src in (sys.argv[1:] or ["-"]):
    for line in (sys.stdin if src == "-" else open(src)):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        print("#", d["config"]["workload"])
        show("headline " + d["dtype"][:5], d)
        for k in ("f16x2", "f32"):
            if k in d:
                show(k, d[k])
        for k, v in (d.get("eval_cache") or {}).items():
            if isinstance(v, dict) and "value" in v:
                show("eval_cache " + k, v)
