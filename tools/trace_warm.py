"""AZSP_RESNET_TRACE helper: warm the chip with untraced launches for a while, then trace (the trace prints only for
the first three traced launches of a process, so the warm-up runs in a child engine without the env)."""
import os, sys, subprocess, time
if __name__ == "__main__":
    B = sys.argv[1] if len(sys.argv) > 1 else "3072"
    extra = sys.argv[2:]                      # e.g. --f16
    env = dict(os.environ); env.pop("AZSP_RESNET_TRACE", None)
    subprocess.run([sys.executable, "tools/infer_bench.py", "--boards", B, "--reps", "3000"] + extra, env=env, stdout=subprocess.DEVNULL)
    env["AZSP_RESNET_TRACE"] = "1"
    subprocess.run([sys.executable, "tools/infer_bench.py", "--boards", B, "--reps", "2"] + extra, env=env)
