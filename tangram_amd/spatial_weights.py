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
        # reference :14-24: neighbours from the connectivity pattern, weights = row-L1-normalised distances
        conn = sp.csr_matrix(adata_sp.obsp["spatial_connectivities"]).astype(np.float32)
        dist = sp.csr_matrix(adata_sp.obsp["spatial_distances"]).astype(np.float32)
        rs = np.asarray(np.abs(dist).sum(axis=1)).reshape(-1)
        rs[rs == 0] = 1.0
        w = sp.diags(1.0 / rs) @ dist
        w = w.multiply(conn != 0).tocsr()              # keep the connectivity pattern
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
