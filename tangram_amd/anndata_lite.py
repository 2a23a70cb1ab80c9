"""A minimal AnnData-shaped container (X, obs, var, uns, obsm, obsp + `adata[:, genes]`), used
(a) as the return type of `map_cells_to_space` when the `anndata` package is not installed and
(b) as the stand-in input in tests (the authoring/GPU images have neither scanpy nor anndata).
It implements only what tangram/mapping_utils.py:141-428 touches."""
from __future__ import annotations

import numpy as np
import pandas as pd


class AnnDataLite:
    def __init__(self, X, obs=None, var=None, uns=None, obsm=None, obsp=None):
        self.X = X
        n_obs, n_var = X.shape
        self.obs = obs if obs is not None else pd.DataFrame(index=[str(i) for i in range(n_obs)])
        self.var = var if var is not None else pd.DataFrame(index=[str(i) for i in range(n_var)])
        if len(self.obs) != n_obs or len(self.var) != n_var:
            raise ValueError("obs/var lengths do not match X")
        self.uns = uns if uns is not None else {}
        self.obsm = obsm if obsm is not None else {}
        self.obsp = obsp if obsp is not None else {}

    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    @property
    def shape(self):
        return self.X.shape

    @property
    def var_names(self):
        return self.var.index

    @property
    def obs_names(self):
        return self.obs.index

    def __getitem__(self, key):
        if not (isinstance(key, tuple) and len(key) == 2):
            raise NotImplementedError("AnnDataLite only supports adata[:, genes]")
        rows, cols = key
        if not (isinstance(rows, slice) and rows == slice(None)):
            raise NotImplementedError("AnnDataLite only supports adata[:, genes]")
        if isinstance(cols, slice):
            idx = np.arange(self.n_vars)[cols]
        else:
            idx = self.var.index.get_indexer(list(cols))
            if (idx < 0).any():
                raise KeyError("unknown variable names")
        X = self.X[:, idx]
        return AnnDataLite(X, obs=self.obs, var=self.var.iloc[idx].copy(), uns=self.uns, obsm=self.obsm, obsp=self.obsp)

    def copy(self):
        return AnnDataLite(self.X.copy(), self.obs.copy(), self.var.copy(), dict(self.uns), dict(self.obsm), dict(self.obsp))


def make_result_anndata(X, obs, var):
    """AnnData if the package is available (what the reference returns), AnnDataLite otherwise."""
    try:
        import anndata  # noqa: F401
        return anndata.AnnData(X=X, obs=obs, var=var)
    except Exception:
        return AnnDataLite(X, obs=obs, var=var)
