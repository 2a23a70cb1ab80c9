"""The tutorial's leave-one-out cross-validation (tutorial_tangram_without_squidpy.ipynb:1445; tangram/utils.py:503-668) end to end:
20 000 cells in 18 clusters x 249 training genes x 9 852 spots, clusters mode, 1 000 epochs per fold, 249 folds -- through
`tangram_amd.cross_val` (16 folds per launch) and, for `--sequential N` folds, through the reference's procedure spelled out with
this package's `map_cells_to_space` one fold after the other (the time of all 249 is extrapolated from those N)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tangram_amd as tg  # noqa: E402
from tangram_amd.anndata_lite import AnnDataLite  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=20000)
    ap.add_argument("--clusters", type=int, default=18)
    ap.add_argument("--genes", type=int, default=249)
    ap.add_argument("--spots", type=int, default=9852)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--sequential", type=int, default=8, help="folds also mapped one after the other (0: skip)")
    ap.add_argument("--folds-per-launch", type=int, default=16)
    opt = ap.parse_args()
    dev = "cuda:0"
    w = make_workload(opt.cells, opt.genes, opt.spots, dev, seed=3)
    S, G = w["S"].cpu().numpy(), w["G"].cpu().numpy()
    genes = [f"g{i}" for i in range(opt.genes)]
    rng = np.random.default_rng(0)
    obs_sc = pd.DataFrame({"cluster": rng.integers(0, opt.clusters, opt.cells).astype(str)}, index=[f"c{i}" for i in range(opt.cells)])
    obs_sp = pd.DataFrame({"rna_count_based_density": G.sum(1) / G.sum(), "uniform_density": np.ones(opt.spots) / opt.spots},
                          index=[f"s{i}" for i in range(opt.spots)])
    ad_sc = AnnDataLite(S, obs=obs_sc, var=pd.DataFrame(index=genes))
    ad_sp = AnnDataLite(G, obs=obs_sp, var=pd.DataFrame(index=genes))
    for ad in (ad_sc, ad_sp):
        ad.uns["training_genes"] = genes
        ad.uns["overlap_genes"] = genes
    kw = dict(cluster_label="cluster", mode="clusters", num_epochs=opt.epochs, device=dev, random_state=1)
    out = {"shape": [opt.clusters, opt.genes - 1, opt.spots], "folds": opt.genes, "epochs": opt.epochs}
    tg.cross_val(ad_sc, ad_sp, **dict(kw, num_epochs=10))                       # warm-up: library load, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cv = tg.cross_val(ad_sc, ad_sp, folds_per_launch=opt.folds_per_launch, **kw)
    torch.cuda.synchronize()
    out["cross_val_s"] = time.perf_counter() - t0
    out["cv"] = {k: float(v) for k, v in cv.items()}
    if opt.sequential:
        folds = list(tg.cv_data_gen(ad_sc, ad_sp, "loo"))[:opt.sequential]
        src = tg.adata_to_cluster_expression(ad_sc, "cluster", True, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for train_genes, test_genes in folds:
            ad_map = tg.map_cells_to_space(ad_sc, ad_sp, cv_train_genes=train_genes, verbose=False, **kw)
            pred = ad_map.X.T @ np.asarray(src[:, test_genes].X)
            g = np.asarray(ad_sp[:, test_genes].X)
            _ = (pred * g).sum(0) / (np.linalg.norm(pred, axis=0) * np.linalg.norm(g, axis=0))
        torch.cuda.synchronize()
        per_fold = (time.perf_counter() - t0) / len(folds)
        out["sequential_s_per_fold"] = per_fold
        out["sequential_s_all_folds_extrapolated"] = per_fold * opt.genes
        out["speedup"] = per_fold * opt.genes / out["cross_val_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
