"""Runs a script of this repository (bench.py, a tools/ script) with the DEVELOPMENT build of the library (tools/build_dev.py: CAR_DEV_FLAGS,
e.g. -DCAR_WG_ORDER=2) in place of the product's: the package's loader is handed the development library before anything else asks for it.
Development tool only — the product never does this.    python tools/run_with_dev_lib.py bench.py --no-extras --cpu-rays 0"""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's, loaded first)

from build_dev import build_dev  # noqa: E402
from cross_attention_renderer_amd import _lib  # noqa: E402

lib = ctypes.CDLL(build_dev())
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args
_lib._lib = lib
print(f"[dev] library {lib._name} (flags: {os.environ.get('CAR_DEV_FLAGS', '')})", flush=True)
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(ROOT, script), run_name="__main__")
