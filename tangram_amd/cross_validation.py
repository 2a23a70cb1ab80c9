"""
`cross_val` / `cv_data_gen` with the reference's signatures (tangram/utils.py:462-668), built on the batched mappings of
SURVEY section 8, f-3.

The reference trains one mapping per held-out gene (or per tenth of the genes) strictly one after the other, each through
`map_cells_to_space(cv_train_genes=...)`, then projects the fold's genes on the host (`project_genes`) and scores them
(`compare_spatial_geneexp`).  Here the training-gene matrices are uploaded ONCE, a fold is a column subset gathered on the device,
`folds_per_launch` folds advance in one launch per kernel (`tangram_amd.batched.train_many` -> `tg_batch`), and a fold's held-out
genes are projected by its mapper while the mapping is still resident in HBM.  Every fold's mapping is the bits of
`map_cells_to_space(cv_train_genes=train_genes, ...)` on the same inputs; scores agree to fp32 rounding of the projection.

Deviations a caller can see: no tqdm bar; the root / anndata loggers are not disabled; `np.float` (removed from NumPy) is `float`.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from . import mapping_optimizer as mo
from .anndata_lite import make_result_anndata
from .batched import train_many
from .host_rng import legacy_normal_f32
from .mapping_utils import (_all_genes_expressed, _training_matrix, adata_to_cluster_expression, annotate_gene_sparsity)


def cv_data_gen(adata_sc, adata_sp, cv_mode="loo"):
    """Generates (train_genes, test_genes) pairs: leave-one-out or 10 contiguous folds of `uns['training_genes']`
    (reference tangram/utils.py:462-500; sklearn's LeaveOneOut / unshuffled KFold(10))."""
    if "training_genes" not in adata_sc.uns.keys():
        raise ValueError("Missing tangram parameters. Run `pp_adatas()`.")
    if "training_genes" not in adata_sp.uns.keys():
        raise ValueError("Missing tangram parameters. Run `pp_adatas()`.")
    if not list(adata_sp.uns["training_genes"]) == list(adata_sc.uns["training_genes"]):
        raise ValueError("Unmatched training_genes field in two Anndatas. Run `pp_adatas()`.")
    genes = np.array(adata_sp.uns["training_genes"])
    n = len(genes)
    if cv_mode == "loo":
        bounds = [(i, i + 1) for i in range(n)]
    elif cv_mode == "10fold":
        if n < 10:                                              # (what sklearn's KFold raises)
            raise ValueError("Cannot have number of splits n_splits=10 greater than the number of samples: n_samples={}.".format(n))
        sizes = np.full(10, n // 10)
        sizes[: n % 10] += 1
        ends = np.cumsum(sizes)
        bounds = list(zip(ends - sizes, ends))
    else:
        raise ValueError("cv_mode must be 'loo' or '10fold'")
    for lo, hi in bounds:
        yield list(genes[:lo]) + list(genes[hi:]), list(genes[lo:hi])


import logging
from contextlib import nullcontext as _nullcontext

from .batched import BATCH_MAX_ELEMENTS


def _fold_footprint(n_src, n_sp, n_genes, precision, constrained):
    """Device bytes ONE fold's mapper holds while it trains: what the C library itself asks for (`tg_query_sizes`: logits + both Adam
    moments; workspace = operand images, the backward product, partial sums), plus the result and the initial-logit upload."""
    import ctypes as ct

    from . import _capi
    cfg = _capi.TgConfig()
    cfg.abi_version = _capi.TG_ABI_VERSION
    cfg.mode = _capi.TG_MODE_CONSTRAINED if constrained else _capi.TG_MODE_MAPPER
    cfg.precision = _capi.PRECISIONS[precision]
    cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.n_spots_total = int(n_src), int(n_genes), int(n_sp), int(n_sp)
    cfg.has_density, cfg.lambda_g1, cfg.lambda_d = 1, 1.0, 1.0
    cfg.beta1, cfg.beta2, cfg.eps = 0.9, 0.999, 1e-8
    sizes = _capi.TgSizes()
    _capi.check(_capi.lib().tg_query_sizes(ct.byref(cfg), ct.byref(sizes)))
    return int(sizes.state_bytes) + int(sizes.workspace_bytes) + 2 * 4 * int(n_src) * int(n_sp)


def _folds_resident(requested, n_src, n_sp, n_genes, device, precision="bf16x3", constrained=False):
    """How many folds' mappers may be RESIDENT at once.  `train_many` builds every mapper of a group before it trains any.  The
    reference trains one fold at a time (utils.py:576-600), so a problem that fits there must fit here: above the size where one
    mapping fills the GPU (`batched.BATCH_MAX_ELEMENTS`: such folds are not batched anyway) the folds run one by one; below it the
    group is capped so that its footprint (`tg_query_sizes` per fold) stays within 0.6 of the free device memory.  The group size
    never changes a fold's result (a batched fold is the bits of the same fold trained alone; tests/test_cross_val.py), only how many
    share a launch; a reduction is logged."""
    if n_src * n_sp > BATCH_MAX_ELEMENTS:
        return 1
    if device.type != "cuda":
        return requested
    per_fold = _fold_footprint(n_src, n_sp, n_genes, precision, constrained)
    free, _ = torch.cuda.mem_get_info(device)
    fit = max(1, int(0.6 * free // max(per_fold, 1)))
    if fit < requested:
        logging.info("tangram_amd.cross_val: folds_per_launch %d -> %d (%.1f GB per fold, %.1f GB of device memory free)",
                     requested, fit, per_fold / 2 ** 30, free / 2 ** 30)
    return min(requested, fit)


def _to_device(x, device):
    return (x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x))).to(device=device, dtype=torch.float32)


def cross_val(
    adata_sc,
    adata_sp,
    cluster_label=None,
    mode="clusters",
    scale=True,
    lambda_d=0,
    lambda_g1=1,
    lambda_g2=0,
    lambda_r=0,
    lambda_count=1,
    lambda_f_reg=1,
    target_count=None,
    num_epochs=1000,
    device="cuda:0",
    learning_rate=0.1,
    cv_mode="loo",
    return_gene_pred=False,
    density_prior=None,
    random_state=None,
    verbose=False,
    *,
    gemm_precision="bf16x3",
    folds_per_launch=16,
    distributed=False,
    group=None,
):
    """Executes cross validation; arguments and returns as the reference (tangram/utils.py:503-668):
    `cv_dict` {'avg_test_score', 'avg_train_score'} and, with `return_gene_pred` in 'loo' mode, `adata_ge_cv` (the held-out
    genes' predicted spatial expression, spots x genes, `var['test_score']`) and `test_gene_df` ('score', 'is_training',
    'sparsity_sp', 'sparsity_sc', 'sparsity_diff').

    Extra keywords: `gemm_precision` (tangram_amd.mapping_optimizer); `folds_per_launch`: folds trained together (tg_batch);
    `distributed=True` (+ optional `group`): the folds are dealt out over the ranks of an initialised torch.distributed process
    group -- fold i to rank i mod world, every rank on its own `device`, no communication while training -- and the per-fold
    results are exchanged once at the end: the same call on every rank, every rank returns the full result."""
    # ---- the argument handling of map_cells_to_space (mapping_utils.py:205-229, :280-307), once for all folds
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
    folds = list(cv_data_gen(adata_sc, adata_sp, cv_mode))
    if not set(["training_genes", "overlap_genes"]).issubset(set(adata_sc.uns.keys())) or \
            not set(["training_genes", "overlap_genes"]).issubset(set(adata_sp.uns.keys())):
        raise ValueError("Missing tangram parameters. Run `pp_adatas()`.")
    device = torch.device(device)

    # the single-cell side every fold trains on and is scored against (:557-558 and mapping_utils.py:231-234)
    adata_src = adata_to_cluster_expression(adata_sc, cluster_label, scale, add_density=True, device=device) if mode == "clusters" else adata_sc
    genes = list(adata_sc.uns["training_genes"])
    pos = {g: i for i, g in enumerate(genes)}
    S_all = _training_matrix(adata_src, adata_src[:, genes], genes, device)
    G_all = _training_matrix(adata_sp, adata_sp[:, genes], genes, device)
    if not _all_genes_expressed(S_all) or not _all_genes_expressed(G_all):
        raise ValueError("Genes with all zero values detected. Run `pp_adatas()`.")
    S_all, G_all = _to_device(S_all, device), _to_device(G_all, device)

    d_source = None
    if isinstance(density_prior, str) and density_prior == "rna_count_based":
        density_prior = adata_sp.obs["rna_count_based_density"]
    elif isinstance(density_prior, str) and density_prior == "uniform":
        density_prior = adata_sp.obs["uniform_density"]
    d = density_prior
    if mode == "clusters":
        d_source = np.array(adata_src.obs["cluster_density"])
    if mode in ["clusters", "constrained"]:
        if density_prior is None:
            d = adata_sp.obs["uniform_density"]
        if lambda_d is None or lambda_d == 0:
            lambda_d = 1

    # A seeded run starts every fold from the same values: the reference re-seeds the global NumPy generator in each mapper and
    # draws (mapping_optimizer.py:147-157, :473-490).  They are drawn once here, exactly as the first fold would (2.3 ms per fold
    # of host time at 18 x 9 852); the generator is left in the state the reference leaves it in after a fold.
    init = {}
    if random_state:
        np.random.seed(seed=random_state)
        n_src, n_sp = int(S_all.shape[0]), int(G_all.shape[0])
        if mode == "constrained":
            legacy_normal_f32((n_src, n_sp), discard=True)                                 # (:475, discarded by :485)
            init = dict(M_init=legacy_normal_f32((n_src, n_sp)), F_init=np.random.normal(0, 1, n_src).astype(np.float32))
        else:
            init = dict(M_init=legacy_normal_f32((n_src, n_sp)))

    def builder(train_genes):
        idx = torch.as_tensor([pos[g] for g in train_genes], device=device, dtype=torch.long)

        def build():
            S, G = S_all.index_select(1, idx).contiguous(), G_all.index_select(1, idx).contiguous()
            if mode == "constrained":
                return mo.MapperConstrained(S=S, G=G, d=d, device=device, random_state=random_state, gemm_precision=gemm_precision,
                                            lambda_d=lambda_d, lambda_g1=lambda_g1, lambda_g2=lambda_g2, lambda_r=lambda_r,
                                            lambda_count=lambda_count, lambda_f_reg=lambda_f_reg, target_count=target_count, **init)
            return mo.Mapper(S=S, G=G, d=d, device=device, random_state=random_state, gemm_precision=gemm_precision,
                             lambda_d=lambda_d, lambda_g1=lambda_g1, lambda_g2=lambda_g2, lambda_r=lambda_r, d_source=d_source, **init)
        return build

    # ---- sparsity columns of compare_spatial_geneexp (:413, :443-449)
    annotate_gene_sparsity(adata_sp)
    annotate_gene_sparsity(adata_src)
    sparsity_sp = adata_sp[:, genes].var["sparsity"]
    sparsity_sc = adata_src[:, genes].var["sparsity"]

    rank, world = 0, 1
    if distributed:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("cross_val(distributed=True) needs an initialised torch.distributed process group")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = list(range(rank, len(folds), world))                  # this rank's folds
    records = {}                                                 # fold -> (test_genes, test_score, train_score, df, prediction or None)
    step = _folds_resident(max(1, int(folds_per_launch)), int(S_all.shape[0]), int(G_all.shape[0]), len(genes), device,
                           precision=gemm_precision, constrained=(mode == "constrained"))
    for g0 in range(0, len(mine), step):
        ids = mine[g0:g0 + step]
        fold_group = [folds[i] for i in ids]
        results, mappers = train_many([builder(tr) for tr, _ in fold_group], num_epochs, learning_rate, device=str(device))
        for fold_id, (train_genes, test_genes), res, mapper in zip(ids, fold_group, results, mappers):
            tidx = torch.as_tensor([pos[g] for g in test_genes], device=device, dtype=torch.long)
            S_test, G_test = S_all.index_select(1, tidx).contiguous(), G_all.index_select(1, tidx)
            # project_genes on the fold's genes (:596-598; constrained: adata_map.X is the unfiltered mapping), held-out columns only
            pred = mapper.project_genes_device(S_test, unfiltered=True) if mode == "constrained" else mapper.project_genes_device(S_test)
            pred = pred.to(torch.float32)
            score = ((pred * G_test).sum(dim=0) / (torch.linalg.norm(pred, dim=0) * torch.linalg.norm(G_test, dim=0))).cpu().numpy()
            mapper.release()
            df = pd.DataFrame({"score": score.astype(np.float64), "is_training": False}, index=test_genes)
            df["sparsity_sp"] = sparsity_sp.loc[test_genes]
            df["sparsity_sc"] = sparsity_sc.loc[test_genes]
            df["sparsity_diff"] = df["sparsity_sp"] - df["sparsity_sc"]
            df = df.sort_values(by="score", ascending=False)
            test_score = df["score"].mean()
            train_score = float(list(res[-1]["main_loss"])[-1])
            keep_pred = cv_mode == "loo" and return_gene_pred                # genes x spots, like adata_ge[:, test_genes].X.T (:602)
            records[fold_id] = (test_genes, test_score, train_score, df, pred.t().cpu().numpy() if keep_pred else None)
            if verbose is True:
                print("cv set: {}----train score: {:.3f}----test score: {:.3f}".format(fold_id + 1, train_score, test_score))
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        # (on the NCCL backend all_gather_object stages through torch.cuda.current_device(), not through `device`: make them agree,
        #  else every rank of a job that never called torch.cuda.set_device stages on cuda:0)
        with (torch.cuda.device(device) if device.type == "cuda" else _nullcontext()):
            dist.all_gather_object(parts, records, group=group)
        records = {k: v for part in parts for k, v in part.items()}
    test_genes_list, test_pred_list, test_score_list, train_score_list, test_df_list = [], [], [], [], []
    for i in range(len(folds)):
        test_genes, test_score, train_score, df, pred = records[i]
        test_genes_list.append(test_genes)
        test_score_list.append(test_score)
        train_score_list.append(train_score)
        test_df_list.append(df)
        if pred is not None:
            test_pred_list.append(pred)

    avg_test_score = np.nanmean(test_score_list)
    avg_train_score = np.nanmean(train_score_list)
    cv_dict = {"avg_test_score": avg_test_score, "avg_train_score": avg_train_score}
    if rank == 0:
        print("cv avg test score {:.3f}".format(avg_test_score))
        print("cv avg train score {:.3f}".format(avg_train_score))
    if cv_mode == "loo" and return_gene_pred:
        test_gene_df = pd.concat(test_df_list, axis=0)
        adata_ge_cv = make_result_anndata(np.squeeze(test_pred_list).T, adata_sp.obs.copy(),
                                          pd.DataFrame(test_score_list, columns=["test_score"], index=np.squeeze(test_genes_list)))
        return cv_dict, adata_ge_cv, test_gene_df
    return cv_dict
