"""Spot-graph weights for the neighbourhood extensions (reference: tangram/spatial_weights.py:5-29), returned as
scipy CSR matrices instead of dense V x V `np.matrix` objects (400 MB at V = 10k; the kernels consume CSR)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def spatial_weights(adata_sp, standardized, self_inclusion):
    if not set(["spatial_connectivities", "spatial_distances"]).issubset(set(adata_sp.obsp.keys())):
        raise ValueError("Missing spatial neighborhood parameters. Run `pp_adatas()` with the spatial information "
                         "stored in `spatial` in `adata_sp.obsm`.")                      # reference :11-12
    if standardized:
        # reference :14-24: for every spot, `neighbors` = the columns where the connectivity row is non-zero and
        # `neighbor_weights` = the non-zero values of the row-L1-normalised distance row, both in ascending column order;
        # libpysal.weights.W(neighbors, neighbor_weights).sparse then pairs them POSITIONALLY (j-th neighbour <- j-th weight,
        # the shorter list wins).  With squidpy's output the two patterns coincide and this is simply the normalised distance
        # at the connectivity positions; the positional pairing is reproduced for inputs where they do not.
        # (The reference also normalises adata_sp.obsp['spatial_distances'] IN PLACE, `copy=False` at :16 -- not reproduced.)
        conn = sp.csr_matrix(adata_sp.obsp["spatial_connectivities"]).astype(np.float32)
        dist = sp.csr_matrix(adata_sp.obsp["spatial_distances"]).astype(np.float64)
        conn.sum_duplicates(); conn.eliminate_zeros(); conn.sort_indices()
        dist.sum_duplicates(); dist.sort_indices()
        rs = np.asarray(np.abs(dist).sum(axis=1)).reshape(-1)
        rs[rs == 0] = 1.0
        dist = (sp.diags(1.0 / rs) @ dist).tocsr()
        dist.eliminate_zeros(); dist.sort_indices()
        n = conn.shape[0]
        nc, nd = np.diff(conn.indptr), np.diff(dist.indptr)
        take = np.minimum(nc, nd)                                   # zip() stops at the shorter list
        rows = np.repeat(np.arange(n), take)
        within = np.arange(take.sum()) - np.repeat(np.cumsum(take) - take, take)
        cols = conn.indices[np.repeat(conn.indptr[:-1], take) + within]
        vals = dist.data[np.repeat(dist.indptr[:-1], take) + within]
        w = sp.csr_matrix((vals.astype(np.float32), (rows, cols)), shape=conn.shape)
    else:
        w = sp.csr_matrix(adata_sp.obsp["spatial_connectivities"]).astype(np.float32)    # reference :26
    if self_inclusion:
        w = (w + sp.identity(w.shape[0], dtype=np.float32, format="csr")).tocsr()        # reference :27-28
    return w.astype(np.float32)


def one_hot_encoding(labels):
    """reference tangram/utils.py:105-123: one column per unique value, in order of first appearance."""
    import pandas as pd
    labels = pd.Series(labels)
    cols = list(labels.unique())
    return np.stack([(labels == c).to_numpy().astype(np.float32) for c in cols], axis=1), cols
