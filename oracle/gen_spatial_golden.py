"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Authoring container only:   python oracle/gen_spatial_golden.py

Runs the UNMODIFIED /root/reference/tangram/spatial_weights.py::spatial_weights on seeded spot graphs and stores its
output in tests/golden/spatial_weights.npz.  libpysal is absent from this image; a stand-in module with the one class the
reference uses (`libpysal.weights.W(neighbors, weights)` with a `.sparse` attribute that pairs neighbours and weights
positionally, ids in sorted order -- libpysal's documented behaviour) is injected into sys.modules before the import.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "spatial_weights.npz")


class _W:
    def __init__(self, neighbors, weights):
        ids = sorted(neighbors.keys())
        n = len(ids)
        pos = {k: i for i, k in enumerate(ids)}
        rows, cols, vals = [], [], []
        for k in ids:
            for j, x in zip(neighbors[k], weights[k]):
                rows.append(pos[k]); cols.append(pos[j]); vals.append(x)
        self.sparse = sp.csr_matrix((np.asarray(vals, dtype=np.float64), (rows, cols)), shape=(n, n))


class _Ad:
    def __init__(self, conn, dist):
        self.obsp = {"spatial_connectivities": conn, "spatial_distances": dist}


def make_graph(V, seed, mismatch):
    """k-nearest-neighbour-like random spot graph; `mismatch`: the distance matrix has a different pattern than the
    connectivities in some rows (extra / missing entries), which exercises the reference's positional pairing."""
    rng = np.random.default_rng(seed)
    conn = np.zeros((V, V))
    for i in range(V):
        nb = rng.choice([j for j in range(V) if j != i], size=rng.integers(0, 7), replace=False)
        conn[i, nb] = 1.0
    dist = conn * rng.uniform(0.5, 3.0, size=(V, V))
    if mismatch:
        for i in rng.choice(V, size=V // 3, replace=False):
            j = rng.integers(0, V)
            dist[i, j] = 0.0 if dist[i, j] != 0 else rng.uniform(0.5, 3.0)
    return conn, dist


def main():
    lp = types.ModuleType("libpysal")
    lp.weights = types.ModuleType("libpysal.weights")
    lp.weights.W = _W
    sys.modules["libpysal"] = lp
    sys.modules["libpysal.weights"] = lp.weights
    spec = importlib.util.spec_from_file_location("ref_sw", "/root/reference/tangram/spatial_weights.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for name, (V, seed, mismatch) in {"match": (40, 1, False), "mismatch": (37, 2, True)}.items():
        conn, dist = make_graph(V, seed, mismatch)
        out[name + "_conn"], out[name + "_dist"] = conn, dist
        for std in (True, False):
            for selfinc in (True, False):
                ad = _Ad(sp.csr_matrix(conn), sp.csr_matrix(dist))       # fresh: the reference normalises the distances in place
                out[f"{name}_std{int(std)}_self{int(selfinc)}"] = np.asarray(ref.spatial_weights(ad, std, selfinc), dtype=np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, sorted(out))


if __name__ == "__main__":
    main()
