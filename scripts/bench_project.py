"""`project_genes` over ALL genes with the mapping resident in HBM (tg_mapper_project_genes) at the
BASELINE shape (30k cells x 10k spots) and the tutorial's gene count (26 496), next to the reference's host path
(`adata_map.X.T @ adata_sc.X`, utils.py:368, NumPy/BLAS on all host cores) timed on a bounded slice of genes.
Also times tiny clusters-mode problems (launch-bound)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tangram_amd.engine import HipMapperEngine  # noqa: E402
from tangram_amd.synthetic import make_workload, init_logits  # noqa: E402


def main():
    dev = "cuda:0"
    out = {}
    C, K, V, K_all = 30000, 1000, 10000, 26496
    w = make_workload(C, K, V, dev, seed=0)
    g = torch.Generator(device=dev).manual_seed(0)
    S_all = torch.rand((C, K_all), generator=g, device=dev)
    S_all = torch.where(S_all < 0.3, S_all * 10.0, torch.zeros_like(S_all))
    for prec in ("bf16x3", "bf16", "fp32"):
        e = HipMapperEngine(w["S"], w["G"], init_logits(C, V, dev, seed=42), d=w["d"], device=dev, precision=prec,
                            lambdas=dict(lambda_d=1.0))
        e.step(3, 0.1)
        e.project_genes(S_all[:, :2000])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Gh = e.project_genes(S_all)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"project_genes_{prec}"] = dict(seconds=dt, genes=K_all, tflops=2.0 * C * V * K_all / dt / 1e12,
                                            genes_per_s=K_all / dt)
        if prec == "bf16x3":
            # reference host path on a slice of 512 genes (P and S on the host, NumPy BLAS, all cores)
            P = e.result().cpu().numpy()
            Sh = S_all[:, :512].cpu().numpy()
            t0 = time.perf_counter()
            ref = P.T @ Sh
            dh = time.perf_counter() - t0
            err = float(np.abs(Gh[:, :512].cpu().numpy() - ref).max() / np.abs(ref).max())
            out["host_numpy"] = dict(seconds_512_genes=dh, genes_per_s=512 / dh, cores=os.cpu_count(),
                                     extrapolated_seconds_all_genes=dh * K_all / 512, max_rel_err_vs_device=err)
            t0 = time.perf_counter()
            e.result().cpu()
            out["d2h_copy_of_P_seconds"] = time.perf_counter() - t0
            del P, Sh, ref
        e.close(); del e, Gh
    del S_all, w
    # tiny clusters-mode problems: one iteration is launch-bound
    for (c, k, v) in ((18, 250, 9852), (30, 1000, 3000), (300, 1000, 3000)):
        w = make_workload(c, k, v, dev, seed=1)
        e = HipMapperEngine(w["S"], w["G"], init_logits(c, v, dev, seed=42), d=w["d"], device=dev, precision="bf16x3",
                            lambdas=dict(lambda_d=1.0))
        e.step(20, 0.1)
        torch.cuda.synchronize()
        n = 500
        hist = e.new_history(n)
        t0 = time.perf_counter()
        e.step(n, 0.1, hist)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[f"tiny_{c}x{k}x{v}"] = dict(us_per_step=1e6 * dt, iters_per_s=1 / dt, main_loss=float(hist[-1, 1]))
        e.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
