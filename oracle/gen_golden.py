"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz by running the UNMODIFIED reference optimizer
(/root/reference/tangram/mapping_optimizer.py, loaded standalone with importlib because
`import tangram` needs scanpy) on CPU, in fp32 (as shipped) and in fp64 (tensors cast to double).
Run in the authoring container only:   python oracle/gen_golden.py
The fixtures travel to the GPU box; the reference does not.

Each fixture stores the inputs' generator parameters (inputs are regenerated from the seed by
oracle.tangram_oracle.make_synthetic), the initial logits, the per-epoch history, the final
mapping P, the final projection P^T S, and the first-step gradient dM.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.tangram_oracle import make_synthetic, grid_graph  # noqa: E402

REF = "/root/reference/tangram/mapping_optimizer.py"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (C, K, V, seed, epochs, mode, kwargs)
    "cells_default": (300, 60, 120, 1, 60, "cells", dict(lambda_g1=1, lambda_d=1)),
    "cells_allreg": (96, 24, 40, 2, 40, "cells",
                     dict(lambda_g1=1, lambda_d=0.7, lambda_g2=0.5, lambda_r=1e-3, lambda_l1=1e-4, lambda_l2=1e-5)),
    "cells_nodensity": (64, 20, 48, 3, 30, "cells", dict(lambda_g1=1, lambda_d=0, no_density=True)),
    "clusters_dsource": (20, 80, 300, 4, 60, "clusters", dict(lambda_g1=1, lambda_d=1)),
    "cells_ragged": (131, 37, 53, 5, 40, "cells", dict(lambda_g1=1, lambda_d=1, lambda_g2=1)),
    "cells_spatial": (200, 40, 100, 6, 40, "spatial",
                      dict(lambda_g1=1, lambda_d=1, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17)),
    "cells_autocorr": (80, 20, 36, 10, 30, "autocorr",
                       dict(lambda_g1=1, lambda_d=1, lambda_getis_ord=0.6, lambda_moran=0.4, lambda_geary=0.3)),
    "cells_val": (100, 24, 40, 9, 20, "cells", dict(lambda_g1=1, lambda_d=1, lambda_g2=0.5, val_each=2)),
    "constrained": (150, 40, 60, 7, 50, "constrained",
                    dict(lambda_d=1, lambda_g1=1, lambda_g2=1, lambda_count=1, lambda_f_reg=1, target_count=40)),
    "constrained_entropy": (90, 30, 50, 8, 30, "constrained",
                            dict(lambda_d=1, lambda_g1=1, lambda_g2=0.3, lambda_r=1e-3, lambda_count=0.5,
                                 lambda_f_reg=2.0, target_count=30)),
    # The parameter grid of the reference's own tests (tests/tangram_test.py:67-103 and :159-170): mode='clusters', 500 epochs,
    # random_state=42, (lambda_g2, lambda_d, density_prior, scale) varied.  In clusters mode map_cells_to_space forces
    # lambda_d >= 1 and falls back to the uniform prior (mapping_utils.py:293-307), so the 9 + 6 rows of the reference grid
    # collapse to these 7 distinct optimizer configurations.  Their data files are missing from the checkout, so the cells
    # are synthetic; the cluster aggregation (sum if scale else mean, densities = cell fractions) follows :103-139.
    "grid_g2_0_d1_uniform_scaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=0, lambda_d=1, prior="uniform", scale=True)),
    "grid_g2_0_d1_uniform_unscaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=0, lambda_d=1, prior="uniform", scale=False)),
    "grid_g2_1_d1_uniform_scaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=1, lambda_d=1, prior="uniform", scale=True)),
    "grid_g2_1_d1_uniform_unscaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=1, lambda_d=1, prior="uniform", scale=False)),
    "grid_g2_0_d2_uniform_scaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=0, lambda_d=2, prior="uniform", scale=True)),
    "grid_g2_0_d1_rna_scaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=0, lambda_d=1, prior="rna_count_based", scale=True)),
    "grid_g2_0_d1_rna_unscaled": (12, 60, 80, 11, 500, "grid", dict(lambda_g1=1, lambda_g2=0, lambda_d=1, prior="rna_count_based", scale=False)),
}
GRID_CELLS = 360
RANDOM_STATE = 42


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_mo", REF)
    mo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mo)
    return mo


def cluster_inputs(n_clusters, K, V, seed, scale, prior):
    """Cell-level synthetic data aggregated like adata_to_cluster_expression (mapping_utils.py:103-139) + the density priors
    of pp_adatas (:88-94).  Returns S [clusters, K], G, d, d_source and the cell-level pieces (for the wrapper-level tests)."""
    data = make_synthetic(GRID_CELLS, K, V, seed=seed)
    rng = np.random.default_rng(seed + 7)
    sizes = GRID_CELLS // n_clusters - (n_clusters - 1) + 2 * np.arange(n_clusters)      # all different: no ties in value_counts
    sizes[-1] += GRID_CELLS - sizes.sum()
    labels = np.repeat(np.arange(n_clusters), sizes)
    rng.shuffle(labels)
    counts = np.bincount(labels, minlength=n_clusters)
    order = np.argsort(-counts, kind="stable")                    # value_counts order: most frequent first
    S_cells = data["S"]
    rows = [S_cells[labels == l].sum(axis=0) if scale else S_cells[labels == l].mean(axis=0) for l in order]
    S = np.stack(rows).astype(np.float32)
    d_source = (counts[order] / counts.sum()).astype(np.float32)
    G = data["G"]
    d = (G.sum(1) / G.sum()).astype(np.float32) if prior == "rna_count_based" else (np.ones(V) / V).astype(np.float32)
    return dict(S=S, G=G, d=d, d_source=d_source, S_cells=S_cells, labels=labels, order=order)


def build_inputs(name):
    C, K, V, seed, epochs, mode, kw = CASES[name]
    kw = dict(kw)
    if mode == "grid":
        ci = cluster_inputs(C, K, V, seed, kw.pop("scale"), kw.pop("prior"))
        args = dict(S=ci["S"], G=ci["G"], d=ci["d"], d_source=ci["d_source"])
        args.update(kw)
        return args, epochs, mode
    data = make_synthetic(C, K, V, seed=seed, n_types=5 if mode == "spatial" else 0)
    args = dict(S=data["S"], G=data["G"])
    if kw.pop("no_density", False):
        args["d"] = None
    else:
        args["d"] = data["d"]
    if mode == "clusters":
        rng = np.random.default_rng(seed + 100)
        ds = rng.random(C).astype(np.float32)
        args["d_source"] = ds / ds.sum()
    if mode == "spatial":
        args["voxel_weights"] = grid_graph(V, standardized=True, self_inclusion=True)
        args["neighborhood_filter"] = grid_graph(V, standardized=False, self_inclusion=False)
        args["ct_encode"] = data["ct_encode"]
    if mode == "autocorr":
        # one matrix serves all three indicators in the reference (mapping_optimizer.py:139-141); row-standardised, no self loops
        args["spatial_weights"] = grid_graph(V, standardized=True, self_inclusion=False)
    args.update(kw)
    return args, epochs, mode


def to_double(mapper):
    for k, v in list(vars(mapper).items()):
        if isinstance(v, torch.Tensor) and v.dtype == torch.float32 and not v.requires_grad:
            setattr(mapper, k, v.double())
    for nm in ("M", "F"):
        if hasattr(mapper, nm):
            setattr(mapper, nm, getattr(mapper, nm).detach().double().requires_grad_(True))
    if hasattr(mapper, "getis_ord_G_star_ref"):
        pass  # None for the cases used here


def run(mo, name, double):
    args, epochs, mode = build_inputs(name)
    val_each = args.pop("val_each", None)
    cls = mo.MapperConstrained if mode == "constrained" else mo.Mapper
    mapper = cls(device="cpu", random_state=RANDOM_STATE, **args)
    if double:
        to_double(mapper)
    M0 = mapper.M.detach().numpy().astype(np.float32).copy()
    F0 = mapper.F.detach().numpy().astype(np.float32).copy() if mode == "constrained" else None
    # first-step gradient
    loss = mapper._loss_fn(verbose=False)[0]
    loss.backward()
    dM0 = mapper.M.grad.detach().numpy().copy()
    dF0 = mapper.F.grad.detach().numpy().copy() if mode == "constrained" else None
    mapper.M.grad = None
    if mode == "constrained":
        mapper.F.grad = None
    if val_each is not None:
        res = mapper.train(num_epochs=epochs, learning_rate=0.1, print_each=None, val_each=val_each)
    else:
        res = mapper.train(num_epochs=epochs, learning_rate=0.1, print_each=None)
    out = dict(M0=M0, dM0=dM0, P=res[0])
    if mode == "constrained":
        out.update(F0=F0, dF0=dF0, F_out=res[1])
        hist = res[2]
        # history entries are str(...) (mapping_optimizer.py:630); the first is "tensor(x, grad_fn=...)"
        def parse(s):
            s = s.replace("tensor(", "").split(",")[0].rstrip(")")
            return float(s)
        for k, v in hist.items():
            out["hist_" + k] = np.array([parse(x) for x in v], dtype=np.float64)
        f = res[1].astype(np.float64)
        out["Ghat"] = res[0].astype(np.float64).T @ (args["S"].astype(np.float64) * f[:, None])
    else:
        hist = res[1]
        for k in ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"]:
            out["hist_" + k] = np.array([float(x) for x in hist[k]], dtype=np.float64)
        if val_each is not None:
            for k in ["val_total_loss", "val_gene_sim", "val_sp_sparsity_weighted_sim", "val_entropy"]:
                out["hist_" + k] = np.array([float(x) for x in hist[k]], dtype=np.float64)
        out["Ghat"] = res[0].astype(np.float64).T @ args["S"].astype(np.float64)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    mo = load_ref()
    names = sys.argv[1:] or list(CASES)          # optional: regenerate only the named cases
    for name in names:
        o32 = run(mo, name, double=False)
        o64 = run(mo, name, double=True)
        blob = {}
        for k, v in o32.items():
            blob["f32_" + k] = v
        for k, v in o64.items():
            if k in ("M0", "F0"):
                continue
            blob["f64_" + k] = v
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, os.path.getsize(path) // 1024, "KiB",
              "final main_loss f32/f64:", o32["hist_main_loss"][-1], o64["hist_main_loss"][-1])


if __name__ == "__main__":
    main()
