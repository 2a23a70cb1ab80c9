"""
Device-side `project_genes` (reference: tangram/utils.py:338-375).  The reference multiplies the C x V mapping by the
full single-cell matrix on the host (`adata_map.X.T @ adata_sc.X`, utils.py:368; 2*C*V*K_all flop, K_all ~ 26k genes
in the tutorial); here the product runs in the forward GEMM kernel of the training loop (tg_mapper_project_genes),
block of genes by block of genes, with the mapping resident in HBM.

Same argument meaning, `ValueError` condition and result contract (`X` = spots x genes, `obs` = adata_map.var,
`var` = adata_sc.var + `is_training`, `uns` = adata_sc.uns).  scanpy is not imported: the gene filter
`sc.pp.filter_genes(min_cells=1)` (:357) and `var_names_make_unique` (:354) are restated on the duck-typed inputs.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from . import mapping_utils as mu
from .anndata_lite import AnnDataLite
from .engine import HipMapperEngine


def _make_unique(names):
    """anndata's `var_names_make_unique` rule: later duplicates get -1, -2, ... appended."""
    seen, out = {}, []
    taken = set(names)
    for n in names:
        if n not in seen:
            seen[n] = 0
            out.append(n)
            continue
        while True:
            seen[n] += 1
            cand = "{}-{}".format(n, seen[n])
            if cand not in taken:
                break
        taken.add(cand)
        out.append(cand)
    return out


def _result(X, obs, var, uns):
    try:
        import anndata
        return anndata.AnnData(X=X, obs=obs, var=var, uns=uns)
    except Exception:
        return AnnDataLite(X, obs=obs, var=var, uns=uns)


def _projection_engine(adata_map, device, gemm_precision):
    """A mapper whose softmax reproduces a GIVEN mapping matrix `adata_map.X`: logits = log P, so softmax(logits) =
    P / rowsum(P).  Returns (engine, row sums): a mapping whose rows do not sum to one (filtered / edited by the caller)
    is projected exactly by scaling the rows of the single-cell matrix with the row sums."""
    P = torch.as_tensor(np.asarray(adata_map.X), dtype=torch.float32, device=device)
    rows = P.sum(dim=1).cpu().numpy().astype(np.float64)
    M0 = torch.log(P).clamp_(min=-1.0e30)
    C, V = P.shape
    S1 = torch.ones((C, 1), dtype=torch.float32, device=device)
    G1 = torch.ones((V, 1), dtype=torch.float32, device=device)
    return HipMapperEngine(S1, G1, M0, device=device, precision=gemm_precision, lambdas=dict(lambda_g1=1.0)), rows


def project_genes(adata_map, adata_sc, cluster_label=None, scale=True, *, mapper=None, device="cuda:0",
                  gemm_precision="bf16x3"):
    """Transfer gene expression from the single cell data onto space (reference utils.py:338-375).

    Like the reference, the projection is `adata_map.X.T @ adata_sc.X` of the mapping matrix the caller passes in (which
    may have been edited or filtered since training): `adata_map.X` is uploaded once and multiplied on the device.
    Extra keywords: `mapper` -- pass the trained `Mapper`/`MapperConstrained` EXPLICITLY (e.g. the
    `adata_map._tangram_amd_mapper` of `map_cells_to_space(..., keep_mapper=True)`) to project with the mapping that is
    still resident in HBM instead; `device`, `gemm_precision` as in `map_cells_to_space`.
    Unlike the reference (`sc.pp.filter_genes`, `var_names_make_unique` mutate the caller's AnnData in place, :351-357),
    only `adata_sc.var.index` is rewritten in place; the gene filter and `n_cells` land on a view."""
    adata_sc.var.index = [g.lower() for g in adata_sc.var.index]                     # :351
    adata_sc.var.index = _make_unique(list(adata_sc.var.index))                      # :354
    X = adata_sc.X
    n_cells = np.asarray((X != 0).sum(axis=0)).reshape(-1)                           # :357 filter_genes(min_cells=1)
    keep = n_cells >= 1
    if not keep.all():
        adata_sc = adata_sc[:, list(adata_sc.var.index[keep])]
    adata_sc.var["n_cells"] = n_cells[keep]
    if cluster_label:                                                                # :359-360
        adata_sc = mu.adata_to_cluster_expression(adata_sc, cluster_label, scale=scale)
    if not adata_map.obs.index.equals(adata_sc.obs.index):                           # :362-363
        raise ValueError("The two AnnDatas need to have same `obs` index.")
    X_sc = adata_sc.X                                                                # :364-365: sparse stays sparse, the gene blocks
    S_all = X_sc if hasattr(X_sc, "tocsr") else np.ascontiguousarray(mu._dense(X_sc), dtype=np.float32)   # are expanded on the device
    own = mapper is None
    if own:
        engine, rows = _projection_engine(adata_map, torch.device(device), gemm_precision)
    else:
        engine, rows = mapper._engine, None
    if engine.C != S_all.shape[0]:
        raise ValueError("The two AnnDatas need to have same `obs` index.")
    if not own:
        # the trained mapper projects through its own seam: on a spot-sharded run that gathers every rank's block of spots
        # (mapper._engine alone is only the LOCAL shard); MapperConstrained: softmax(M) without the filter, like adata_map.X
        from .mapping_optimizer import MapperConstrained
        if isinstance(mapper, MapperConstrained):
            X_space = mapper.project_genes_device(S_all, unfiltered=True)
        else:
            X_space = mapper.project_genes_device(S_all)
        X_space = X_space.cpu().numpy()
        adata_ge = _result(X_space, adata_map.var, adata_sc.var, adata_sc.uns)
        training_genes = adata_map.uns["train_genes_df"].index.values
        adata_ge.var["is_training"] = adata_ge.var.index.isin(training_genes)
        return adata_ge
    if rows is not None and np.abs(rows - 1.0).max() > 1e-6:                         # not row-stochastic: P^T S = (P / r)^T (r S)
        if hasattr(S_all, "tocsr"):
            import scipy.sparse as sp
            S_all = (sp.diags(rows) @ S_all.tocsr()).tocsr()
        else:
            S_all = np.ascontiguousarray(S_all * rows[:, None], dtype=np.float32)
    X_space = engine.project_genes(S_all, unfiltered=True).cpu().numpy()             # :366  (adata_map.X.T @ adata_sc.X)
    if own:
        engine.release()
    adata_ge = _result(X_space, adata_map.var, adata_sc.var, adata_sc.uns)           # :367-369
    training_genes = adata_map.uns["train_genes_df"].index.values                    # :370-371
    adata_ge.var["is_training"] = adata_ge.var.index.isin(training_genes)
    return adata_ge
