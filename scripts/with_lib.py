#!/usr/bin/env python3
"""EXPERIMENT TOOL (not part of the product): run a script of this repository against ANOTHER build of libtangram_hip.so.

    python scripts/with_lib.py build/ab_variant.so bench.py --steps 40 ...

The product binding (tangram_amd/_capi.py) loads the in-tree library only; kernel A/B runs point the binding at a variant
build here, before anything has loaded the library, and then execute the script in this process."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
if not os.path.exists(lib):
    sys.exit(f"with_lib.py: {lib} does not exist")
from tangram_amd import _capi  # noqa: E402
_capi.LIB_PATH = lib
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
