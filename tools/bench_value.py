"""Print episodes/s and the mean inference-launch duration of one bench.py run (helper for tools/ab.sh)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--single"] + sys.argv[1:], capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
print({"episodes_per_s": round(d["value"], 1), "launch_ms": round(d["roofline"]["avg_launch_ms"], 4),
       "frac": round(d["roofline"]["frac"], 3), "dtype": d["dtype"]})
