"""
Drop-in `map_cells_to_space` (reference: tangram/mapping_utils.py:141-428) on top of the MI355X-native
`Mapper` / `MapperConstrained`.  Same keyword arguments, defaulting rules, `ValueError` conditions and
result contract (`.X`, `.obs`, `.var`, `.uns['train_genes_df']`, `.uns['training_history']`, `obs['F_out']`).

Deliberate differences from the reference, all documented in DESIGN.md:
  * `device` defaults to "cuda:0" (the reference defaults to "cpu"); there is no CPU path here.
  * scanpy is not imported: the AnnData inputs are duck-typed (`.X`, `.obs`, `.var`, `.uns`, `adata[:, genes]`),
    and the result is an `anndata.AnnData` when that package is installed, else `AnnDataLite`.
  * a dense `adata_sc.X` works (the reference calls `.toarray()` on an ndarray at :262).
  * the per-gene training scores (:402-410) are computed from the projection P^T S evaluated on the GPU instead
    of a NumPy `adata_map.X.T @ S` on the host.
  * the spatial terms (neighbourhood, cell-type islands, Getis-Ord, Moran, Geary) take the spot graph as scipy CSR
    (tangram_amd/spatial_weights.py) instead of dense V x V matrices.
Extra keyword: `gemm_precision` (see tangram_amd.mapping_optimizer).
"""
from __future__ import annotations

import logging

import numpy as np
import pandas as pd
import torch

from . import mapping_optimizer as mo
from . import spatial_weights as sw
from .anndata_lite import AnnDataLite, make_result_anndata

logging.getLogger().setLevel(logging.INFO)


def _dense(X):
    if hasattr(X, "toarray"):
        return np.asarray(X.toarray())
    if isinstance(X, np.matrix):
        return np.asarray(X)
    if isinstance(X, np.ndarray):
        return X
    raise NotImplementedError("AnnData X has unrecognized type: {}".format(type(X)))   # reference :264-266


def _training_matrix(adata, view, training_genes, device):
    """The [n_obs, K] float32 matrix of the training genes (reference :259-275).  A sparse `adata.X` is NOT densified on the host:
    its CSR arrays are uploaded once and the training-gene columns are gathered on the device (tangram_amd.preprocess) -- a
    float32 DEVICE tensor comes back, which `Mapper` consumes as it is.  Dense matrices take the host path like the reference."""
    X = adata.X
    if hasattr(X, "tocsr") and not isinstance(X, np.matrix):
        from . import preprocess as pre
        cols = pd.Index(adata.var.index).get_indexer(list(training_genes))
        if (cols < 0).any():
            raise ValueError("Given training genes list should be subset of two AnnDatas.")
        return pre.gather_training_genes(X, cols, device)
    return np.array(_dense(view.X), dtype="float32")


def _all_genes_expressed(A):
    if isinstance(A, torch.Tensor):
        return bool((A != 0).any(dim=0).all().item())
    return bool(A.any(axis=0).all())


def annotate_gene_sparsity(adata):
    """reference tangram/utils.py:46-61"""
    X = adata.X
    mask = (X != 0)
    gene_sparsity = np.asarray(mask.sum(axis=0)).reshape(-1) / adata.n_obs
    adata.var["sparsity"] = 1 - gene_sparsity


def density_priors(adata_sp, device="cuda:0"):
    """The two density priors of `pp_adatas` (reference tangram/mapping_utils.py:81-89) written to `adata_sp.obs`:
    `uniform_density` = 1 / n_spots and `rna_count_based_density` = fraction of the RNA counts per spot -- the row sums of
    `adata_sp.X` (sparse stays sparse) taken on the device in double (tangram_amd.preprocess.rna_count_density)."""
    from . import preprocess as pre
    n = adata_sp.X.shape[0]
    adata_sp.obs["uniform_density"] = np.ones(n) / n
    adata_sp.obs["rna_count_based_density"] = pre.rna_count_density(adata_sp.X, device).cpu().numpy()


def _cluster_sums_on_device(X, labels, unique_labels, scale, device, block=4096):
    """[n_clusters, n_genes] float64 like the reference's X_new, gene block by gene block: the block of columns is gathered on
    the device (CSR uploaded once; a dense X is uploaded block-wise) and reduced per cluster by tg_cluster_aggregate."""
    from . import preprocess as pre
    n_genes = X.shape[1]
    out = np.empty((len(unique_labels), n_genes))
    dev = pre._check_device(device)
    csr = pre.DeviceCSR(X, dev) if hasattr(X, "tocsr") else None
    for k0 in range(0, n_genes, block):
        k1 = min(n_genes, k0 + block)
        if csr is not None:
            blk = pre.gather_training_genes(csr, np.arange(k0, k1), dev)
        else:
            blk = torch.as_tensor(np.ascontiguousarray(np.asarray(X)[:, k0:k1], dtype=np.float32), device=dev)
        out[:, k0:k1] = pre.cluster_expression(blk, labels, unique_labels, scale=scale).cpu().numpy()
    return out


def adata_to_cluster_expression(adata, cluster_label, scale=True, add_density=True, *, device=None):
    """Cluster-level expression (reference tangram/mapping_utils.py:103-139): one observation per cluster,
    sum (scale=True) or mean of its cells; `obs['cluster_density']` = fraction of cells per cluster.
    Extra keyword `device`: run the per-cluster reductions on that HIP device (double accumulation, the sparse matrix is
    uploaded as CSR and never densified on the host) instead of the NumPy loop."""
    try:
        value_counts = adata.obs[cluster_label].value_counts(normalize=True)
    except KeyError:
        raise ValueError("Provided label must belong to adata.obs.")
    unique_labels = value_counts.index
    new_obs = pd.DataFrame({cluster_label: unique_labels})
    X = adata.X
    labels = adata.obs[cluster_label].to_numpy()
    if device is not None:
        X_new = _cluster_sums_on_device(X, labels, list(unique_labels), scale, device)
    else:
        X_new = np.empty((len(unique_labels), adata.shape[1]))
        for index, l in enumerate(unique_labels):
            rows = np.where(labels == l)[0]
            sub = X[rows]
            X_new[index] = np.asarray(sub.mean(axis=0) if not scale else sub.sum(axis=0)).reshape(-1)
    if add_density:
        new_obs["cluster_density"] = new_obs[cluster_label].map(lambda i: value_counts[i])
    new_obs.index = new_obs.index.astype(str)
    return AnnDataLite(X_new, obs=new_obs, var=adata.var, uns=adata.uns)


def map_cells_to_space(
    adata_sc,
    adata_sp,
    cv_train_genes=None,
    cluster_label=None,
    mode="cells",
    device="cuda:0",
    learning_rate=0.1,
    num_epochs=1000,
    scale=True,
    lambda_d=0,
    lambda_g1=1,
    lambda_g2=0,
    lambda_r=0,
    lambda_l1=0,
    lambda_l2=0,
    lambda_count=1,
    lambda_f_reg=1,
    target_count=None,
    lambda_neighborhood_g1=0,
    lambda_ct_islands=0,
    lambda_getis_ord=0,
    lambda_moran=0,
    lambda_geary=0,
    random_state=None,
    verbose=True,
    density_prior="rna_count_based",
    *,
    gemm_precision="bf16x3",
    keep_mapper=False,
    distributed=False,
    group=None,
    s_exact="auto",
    init="reference",
):
    """Map single cell data (`adata_sc`) on spatial data (`adata_sp`); see the reference docstring (:169-203).

    Extra keywords: `s_exact="auto"` (default): the library checks the training-gene matrix of `adata_sc` once; when it is bf16-exact
    (raw counts) both GEMMs run two matrix-core products per element instead of three -- the same values --, otherwise (normalised /
    log-transformed expression) the general three-product path; `s_exact=False` forces the general path; `init="device"` (opt-in):
    initial logits from the library's device generator instead of NumPy's stream (see tangram_amd.mapping_optimizer.Mapper); `gemm_precision` (tangram_amd.mapping_optimizer); `distributed=True` (+ optional `group`): shard the spots over
    the ranks of an initialised torch.distributed process group -- opt-in, the same call on every rank, every rank gets the full
    result (tangram_amd.mapping_optimizer); `keep_mapper=True` leaves the trained mapper on the
    result as `adata_map._tangram_amd_mapper` so that `tangram_amd.project_genes(..., mapper=adata_map._tangram_amd_mapper)`
    can project with the mapping still resident in HBM.  Default False: like the reference, the result owns no device
    memory -- the logits, both Adam moments, X and the workspace (>= 16 bytes per cell x spot) are released before returning."""
    # ---- argument checks, reference :205-229
    if lambda_g1 == 0:
        raise ValueError("lambda_g1 cannot be 0.")
    if (type(density_prior) is str) and (density_prior not in ["rna_count_based", "uniform", None]):
        raise ValueError("Invalid input for density_prior.")
    if density_prior is not None and (lambda_d == 0 or lambda_d is None):
        lambda_d = 1
    if lambda_d > 0 and density_prior is None:
        raise ValueError("When lambda_d is set, please define the density_prior.")
    if mode not in ["clusters", "cells", "constrained"]:
        raise ValueError('Argument "mode" must be "cells", "clusters" or "constrained')
    if mode == "clusters" and cluster_label is None:
        raise ValueError("A cluster_label must be specified if mode is 'clusters'.")
    if mode == "constrained" and not all([target_count, lambda_f_reg, lambda_count]):
        raise ValueError("target_count, lambda_f_reg and lambda_count must be specified if mode is 'constrained'.")

    if mode == "clusters":                                                     # :231-234 (reductions on the device)
        adata_sc = adata_to_cluster_expression(adata_sc, cluster_label, scale, add_density=True, device=device)

    # ---- tangram parameters in uns, :236-254
    if not set(["training_genes", "overlap_genes"]).issubset(set(adata_sc.uns.keys())):
        raise ValueError("Missing tangram parameters. Run `pp_adatas()`.")
    if not set(["training_genes", "overlap_genes"]).issubset(set(adata_sp.uns.keys())):
        raise ValueError("Missing tangram parameters. Run `pp_adatas()`.")
    assert list(adata_sp.uns["training_genes"]) == list(adata_sc.uns["training_genes"])
    if cv_train_genes is None:
        training_genes = adata_sc.uns["training_genes"]
    else:
        if set(cv_train_genes).issubset(set(adata_sc.uns["training_genes"])):
            training_genes = cv_train_genes
        else:
            raise ValueError("Given training genes list should be subset of two AnnDatas.")

    logging.info("Allocate tensors for mapping.")
    sc_view = adata_sc[:, training_genes]
    sp_view = adata_sp[:, training_genes]
    S = _training_matrix(adata_sc, sc_view, training_genes, device)           # :259-266
    G = _training_matrix(adata_sp, sp_view, training_genes, device)           # :268-275
    if not _all_genes_expressed(S) or not _all_genes_expressed(G):            # :277
        raise ValueError("Genes with all zero values detected. Run `pp_adatas()`.")

    # ---- density prior, :280-307
    d_source = None
    d_str = density_prior
    if type(density_prior) is np.ndarray:
        d_str = "customized"
    if isinstance(density_prior, str) and density_prior == "rna_count_based":
        density_prior = adata_sp.obs["rna_count_based_density"]
    elif isinstance(density_prior, str) and density_prior == "uniform":
        density_prior = adata_sp.obs["uniform_density"]
    if mode == "cells":
        d = density_prior
    if mode == "clusters":
        d_source = np.array(adata_sc.obs["cluster_density"])
    if mode in ["clusters", "constrained"]:
        if density_prior is None:
            d = adata_sp.obs["uniform_density"]
            d_str = "uniform"
        else:
            d = density_prior
        if lambda_d is None or lambda_d == 0:
            lambda_d = 1

    device = torch.device(device)                                             # :310
    print_each = 100 if verbose else None                                     # :312-315

    if mode in ["cells", "clusters"]:
        voxel_weights, neighborhood_filter, ct_encode, spatial_weights = None, None, None, None      # :318-329
        if lambda_neighborhood_g1 > 0:
            voxel_weights = sw.spatial_weights(adata_sp, standardized=True, self_inclusion=True)
        if lambda_ct_islands > 0:
            if cluster_label not in adata_sc.obs.keys():
                raise ValueError("cluster_label must be specified for the cell type island extension.")
            neighborhood_filter = sw.spatial_weights(adata_sp, standardized=False, self_inclusion=False)
            ct_encode, _ = sw.one_hot_encoding(adata_sc.obs[cluster_label])
        if lambda_moran > 0 or lambda_geary > 0:
            spatial_weights = sw.spatial_weights(adata_sp, standardized=True, self_inclusion=False)
        if lambda_getis_ord > 0:                                              # overrides the matrix above, like :328-329
            spatial_weights = sw.spatial_weights(adata_sp, standardized=False, self_inclusion=True)
        hyperparameters = {                                                   # :331-348
            "lambda_d": lambda_d, "lambda_g1": lambda_g1, "lambda_g2": lambda_g2, "lambda_r": lambda_r,
            "lambda_l1": lambda_l1, "lambda_l2": lambda_l2, "d_source": d_source,
            "lambda_neighborhood_g1": lambda_neighborhood_g1, "voxel_weights": voxel_weights,
            "lambda_ct_islands": lambda_ct_islands, "neighborhood_filter": neighborhood_filter, "ct_encode": ct_encode,
            "lambda_getis_ord": lambda_getis_ord, "lambda_moran": lambda_moran, "lambda_geary": lambda_geary,
            "spatial_weights": spatial_weights,
        }
        logging.info("Begin training with {} genes and {} density_prior in {} mode...".format(
            len(training_genes), d_str, mode))
        mapper = mo.Mapper(S=S, G=G, d=d, device=device, random_state=random_state, gemm_precision=gemm_precision,
                           distributed=distributed, group=group, s_exact=s_exact, init=init, **hyperparameters)   # :355-357
        mapping_matrix, training_history = mapper.train(
            learning_rate=learning_rate, num_epochs=num_epochs, print_each=print_each)   # :361-363
    else:
        hyperparameters = {                                                   # :367-375
            "lambda_d": lambda_d, "lambda_g1": lambda_g1, "lambda_g2": lambda_g2, "lambda_r": lambda_r,
            "lambda_count": lambda_count, "lambda_f_reg": lambda_f_reg, "target_count": target_count,
        }
        logging.info("Begin training with {} genes and {} density_prior in {} mode...".format(
            len(training_genes), d_str, mode))
        mapper = mo.MapperConstrained(S=S, G=G, d=d, device=device, random_state=random_state,
                                      gemm_precision=gemm_precision, distributed=distributed, group=group, s_exact=s_exact, init=init,
                                      **hyperparameters)                      # :383-385
        mapping_matrix, F_out, training_history = mapper.train(
            learning_rate=learning_rate, num_epochs=num_epochs, print_each=print_each)   # :387-389

    logging.info("Saving results..")
    adata_map = make_result_anndata(mapping_matrix, sc_view.obs.copy(), sp_view.obs.copy())   # :392-396
    if mode == "constrained":
        adata_map.obs["F_out"] = F_out                                        # :398-399

    # ---- per-gene training score, :401-410 (projection evaluated on the GPU; constrained: unfiltered like :402)
    if mode == "constrained":
        G_predicted = mapper.project_genes_device(S, unfiltered=True)
    else:
        G_predicted = mapper.project_genes_device()
    G_dev = torch.as_tensor(G).to(device=G_predicted.device, dtype=torch.float32)
    num = (G_dev * G_predicted).sum(dim=0)
    den = torch.linalg.norm(G_dev, dim=0) * torch.linalg.norm(G_predicted, dim=0)
    cos_sims = (num / den).cpu().numpy()
    df_cs = pd.DataFrame(cos_sims, list(training_genes), columns=["train_score"])
    df_cs = df_cs.sort_values(by="train_score", ascending=False)
    adata_map.uns["train_genes_df"] = df_cs

    # ---- sparsity annotations, :412-424
    annotate_gene_sparsity(adata_sc)
    annotate_gene_sparsity(adata_sp)
    adata_map.uns["train_genes_df"]["sparsity_sc"] = adata_sc[:, training_genes].var.sparsity
    adata_map.uns["train_genes_df"]["sparsity_sp"] = adata_sp[:, training_genes].var.sparsity
    adata_map.uns["train_genes_df"]["sparsity_diff"] = (
        adata_sp[:, training_genes].var.sparsity - adata_sc[:, training_genes].var.sparsity)
    adata_map.uns["training_history"] = training_history                      # :426
    if keep_mapper:     # opt-in: the trained mapping stays resident in HBM (not part of the AnnData contract)
        object.__setattr__(adata_map, "_tangram_amd_mapper", mapper)
    else:
        mapper.release()
    return adata_map
