"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of /root/reference/tangram/spatial_weights.py:5-29 (`spatial_weights`), plain loops, dense output like
the reference's `np.matrix`.  The arithmetic the reference delegates to third-party packages is restated from their
published behaviour:
  * sklearn.preprocessing.normalize(X, norm="l1", axis=1)  (scikit-learn, unpinned in the reference's setup.py:26-33):
    every row divided by the sum of the absolute values of its entries; all-zero rows stay zero;
  * libpysal.weights.W(neighbors, weights).sparse  (libpysal, pulled in by squidpy; absent here): a V x V matrix with
    sparse[i, neighbors[i][j]] = weights[i][j] -- neighbours and weights are paired POSITIONALLY (zip), ids in sorted order.
oracle/gen_spatial_golden.py runs the UNMODIFIED reference function (with a stand-in for the absent libpysal that implements
exactly that pairing) and stores tests/golden/spatial_weights.npz; tests/test_spatial_weights.py holds this restatement and
tangram_amd/spatial_weights.py to those fixtures.
"""
import numpy as np


def spatial_weights_oracle(conn, dist, standardized, self_inclusion):
    """conn, dist: dense V x V arrays (adata_sp.obsp['spatial_connectivities'] / ['spatial_distances'])."""
    conn = np.asarray(conn, dtype=np.float64)
    dist = np.asarray(dist, dtype=np.float64)
    V = conn.shape[0]
    if standardized:                                                 # reference :14-24
        g = dist.copy()
        for i in range(V):
            s = np.abs(g[i]).sum()
            if s != 0:
                g[i] /= s                                            # :16 normalize(norm="l1", axis=1)
        out = np.zeros((V, V))
        for i in range(V):
            neigh = np.where(conn[i] != 0)[0]                        # :20
            wts = g[i][np.where(g[i] != 0)[0]]                       # :21
            for j, x in zip(neigh, wts):                             # :23 libpysal.weights.W(...).sparse
                out[i, j] = x
    else:
        out = conn.copy()                                            # :26
    if self_inclusion:
        out = out + np.eye(V)                                        # :27-28
    return out
