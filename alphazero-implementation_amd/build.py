#!/usr/bin/env python3
"""Builds csrc/libazsp.so for gfx950 with hipcc (cross-compiles without a GPU).

    python alphazero-implementation_amd/build.py [--force] [--save-temps] [--diag]

--diag builds csrc/libazsp_diag.so instead (objects under csrc/diag/), with -DAZSP_DIAG_BUILD: the timing-only
ablation instantiations of the split-fp16 kernel (AZSP_F16_ABLATE, wrong results by design) exist only there;
tools/ pick it up through AZSP_LIB.  The product library never contains them.

Tree/engine translation units are compiled with -ffp-contract=off (bit-exact float64/float32
arithmetic against the reference); the inference kernel keeps the default fused contraction."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libazsp.so")
ARCH = "gfx950"

COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function"]
UNITS = {
    "azsp_tree.hip": ["-ffp-contract=off"],
    "azsp_engine.hip": ["-ffp-contract=off"],
    "azsp_resnet.hip": [],
    "azsp_resnet_k1.hip": ["-ftemplate-depth=2048"],     # kernel instantiations (csrc/azsp_resnet_kernel.h), split so that
    "azsp_resnet_k2.hip": ["-ftemplate-depth=2048"],     # the statically scheduled towers compile side by side
    "azsp_resnet_k3.hip": [],
    "azsp_resnet_k4.hip": [],
    "azsp_resnet_k5.hip": ["-ftemplate-depth=2048"],     # the two large-batch kernels with the tree step fused in
    "azsp_resnet_f16.hip": ["-ftemplate-depth=2048"],
}
HEADERS = ["azsp_internal.h", "azsp_games.h", "azsp_resnet_host.h", "azsp_resnet_kernel.h", "azsp_tree_dev.h",
           os.path.join("..", "..", "include", "azsp.h")]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, save_temps: bool = False, diag: bool = False) -> str:
    cc = hipcc()
    objs, jobs = [], []
    hdr_paths = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objdir = os.path.join(CSRC, "diag") if diag else CSRC
    out = os.path.join(CSRC, "libazsp_diag.so") if diag else OUT
    os.makedirs(objdir, exist_ok=True)
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or newer(s, o) or any(newer(h, o) for h in hdr_paths):
            cmd = [cc, *COMMON, *extra, *(["-DAZSP_DIAG_BUILD"] if diag else []), "-c", s, "-o", o]
            if save_temps:
                cmd.insert(1, "-save-temps=obj")
            jobs.append(cmd)
        objs.append(o)
    if jobs:                                   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(run, jobs))
    if force or any(newer(o, out) for o in objs):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, diag="--diag" in sys.argv))
