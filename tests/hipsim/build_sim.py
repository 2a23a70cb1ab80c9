"""TEST INFRASTRUCTURE: builds the CPU-emulated variant of the C ABI (same sources, -DTG_SIM) used by
the `not gpu` tests.  The product never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "tangram_amd", "csrc", "tg_capi.hip")
OUT = os.path.join(HERE, "libtangram_sim.so")
import glob
DEPS = [SRC] + sorted(glob.glob(os.path.join(ROOT, "tangram_amd", "csrc", "tg_*.h"))) + [os.path.join(HERE, "hipsim.h"),
                                                                                       os.path.join(ROOT, "include", "tangram_hip.h")]


def host_clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(c):
            return c
    return None


def build_sim():
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    cc = host_clang()
    if cc is None:
        return None
    cmd = [cc, "-x", "c++", "-std=c++17", "-O2", "-DTG_SIM", "-I", HERE, "-shared", "-fPIC", "-w", SRC, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("sim build failed:\n" + r.stderr)
    return OUT
