"""BASELINE config 5 on one GPU (SURVEY 0-7 splits it): 5a = mode 'constrained' (density + count + f_reg),
5b = mode 'cells' + lambda_neighborhood_g1 + lambda_ct_islands on a synthetic 2-D grid spot graph (CSR)."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tangram_amd.engine import HipMapperEngine  # noqa: E402
from tangram_amd.synthetic import make_workload, init_logits  # noqa: E402


def grid_csr(V):
    w = int(np.ceil(np.sqrt(V)))
    idx = np.arange(V)
    r, c = idx // w, idx % w
    rows, cols = [], []
    for dr, dc in ((0, 1), (1, 0), (0, -1), (-1, 0)):
        rr, cc = r + dr, c + dc
        j = rr * w + cc
        ok = (rr >= 0) & (cc >= 0) & (cc < w) & (j < V) & (j >= 0)
        rows.append(idx[ok]); cols.append(j[ok])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    return sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(V, V))


def timeit(eng, steps=20, warmup=4):
    eng.step(warmup, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = eng.new_history(steps)
    eng.step(steps, 0.1, hist)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, hist[-1].cpu().numpy()


def main():
    dev = "cuda:0"
    C, K, V, T = 30000, 1000, 10000, 18
    w = make_workload(C, K, V, dev, seed=0)
    out = {}
    for prec in ("bf16x3", "bf16"):
        M0 = init_logits(C, V, dev, seed=42)
        F0 = torch.randn(C, device=dev)
        e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], F0=F0, mode="constrained", device=dev, precision=prec,
                            lambdas=dict(lambda_g1=1, lambda_d=1, lambda_g2=0, lambda_count=1, lambda_f_reg=1), target_count=float(V))
        dt, row = timeit(e)
        out[f"cfg5a_constrained_{prec}"] = dict(ms_per_step=1e3 * dt, iters_per_s=1 / dt, main_loss=float(row[1]), count_reg=float(row[9]))
        e.close(); del e, M0
        N = grid_csr(V)
        rs = np.asarray(N.sum(1)).reshape(-1); rs[rs == 0] = 1
        W = (sp.diags(1.0 / rs) @ N + sp.identity(V, format="csr")).tocsr()
        lab = (w["assign"].cpu().numpy() * T) // V
        E = np.zeros((C, T), np.float32); E[np.arange(C), lab] = 1
        M0 = init_logits(C, V, dev, seed=42)
        e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=dev, precision=prec,
                            lambdas=dict(lambda_g1=1, lambda_d=1, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17),
                            voxel_weights=W, neighborhood_filter=N, ct_encode=E)
        dt, row = timeit(e)
        out[f"cfg5b_spatial_{prec}"] = dict(ms_per_step=1e3 * dt, iters_per_s=1 / dt, main_loss=float(row[1]), nb_score=float(row[7]), ct_penalty=float(row[8]))
        e.close(); del e, M0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
