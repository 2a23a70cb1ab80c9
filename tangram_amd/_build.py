"""Builds libtangram_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no fallback."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "tg_capi.hip")
OUT = os.path.join(CSRC, "libtangram_hip.so")
HEADERS = ("tg_device.h", "tg_kernels.h", "tg_peer.h", "tg_gemm.h", "tg_stats.h", "tg_spatial.h", "tg_small.h", "tg_update.h", "tg_setup.h")
DEPS = [SRC] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(os.path.dirname(HERE), "include", "tangram_hip.h")]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build the HIP extension")


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    if not force and not _stale(OUT, DEPS):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(OUT + ".tmp", OUT)
    return OUT
