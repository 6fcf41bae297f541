"""pytest configuration: registers the `gpu` marker and puts the repo root and the package
directory (`alphazero-implementation_amd/`, hyphenated, hence not importable by name) on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-implementation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _ensure_built():
    """The CPU suite checks that the C-ABI library loads and exports every declared symbol: build it on demand
    (hipcc cross-compiles without a GPU) when the tests are run on a fresh checkout."""
    lib = os.path.join(PKG, "csrc", "libazsp.so")
    if os.path.exists(lib):
        return
    import importlib.util
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return
    spec = importlib.util.spec_from_file_location("azsp_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()


_ensure_built()
