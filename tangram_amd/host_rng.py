"""The initial values of a mapper -- `np.random.normal(0, 1, shape)` from NumPy's GLOBAL legacy generator, as the reference
draws them (tangram/mapping_optimizer.py:147-157, :473-490) -- produced by tangram_amd/csrc/tg_host_rng.c: the same bits and the
same generator state afterwards, with the Box-Muller transform spread over the host's threads (2.6e8 values: 2.6 s -> a few
tenths of a second).  Without the helper library (not built, no gcc) NumPy itself is called: same result, slower."""
from __future__ import annotations

import ctypes as ct
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tg_host_rng.c")
OUT = os.path.join(HERE, "csrc", "libtangram_host.so")
MIN_VALUES = 1 << 20                     # below this NumPy is as fast (10 ms)
_lib = None
_tried = False


def build(force=False, verbose=False):
    """gcc -O3 -fopenmp (plain x86-64: no FMA contraction, IEEE semantics); returns the path or None without a compiler."""
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        return None
    cmd = [cc, "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", SRC, "-o", OUT + ".tmp", "-lm"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the host helper failed:\n" + r.stdout + r.stderr)
    os.replace(OUT + ".tmp", OUT)
    return OUT


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.path.exists(OUT):
            try:
                lib = ct.CDLL(OUT)
                lib.tg_legacy_normal_f32.restype = ct.c_int
                lib.tg_legacy_normal_f32.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.POINTER(ct.c_double),
                                                     ct.c_void_p, ct.c_int64, ct.c_int]
                _lib = lib
            except OSError:
                _lib = None
    return _lib


def legacy_normal_f32(shape, discard=False):
    """`np.random.normal(0, 1, shape).astype(np.float32)` on the global generator (None with `discard`: the draws are only consumed)."""
    n = int(np.prod(shape))
    lib = _load() if n >= MIN_VALUES else None
    if lib is None:
        x = np.random.normal(0, 1, shape)
        return None if discard else x.astype(np.float32)
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        x = np.random.normal(0, 1, shape)
        return None if discard else x.astype(np.float32)
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    c_pos, c_has, c_g = ct.c_int(int(pos)), ct.c_int(int(has_gauss)), ct.c_double(float(cached))
    out = None if discard else np.empty(n, dtype=np.float32)
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    rc = lib.tg_legacy_normal_f32(key.ctypes.data, ct.byref(c_pos), ct.byref(c_has), ct.byref(c_g),
                                  None if discard else out.ctypes.data, n, threads)
    if rc != 0:
        raise MemoryError("tg_legacy_normal_f32: out of host memory")
    np.random.set_state((name, key, c_pos.value, c_has.value, c_g.value))
    return None if discard else out.reshape(shape)
