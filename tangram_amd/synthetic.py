"""Synthetic planted-mapping workloads of the shapes named in BASELINE.json, generated
directly on the target device with torch RNG: count-like single-cell matrix S (negative-binomial x
Bernoulli(0.3), ~70 % zeros), Poisson spatial matrix G around the planted assignment, rna-count density d."""
from __future__ import annotations

import torch


def make_workload(C, K, V, device, seed=0, chunk=4096):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dev = torch.device(device)
    assign = torch.randint(0, V, (C,), device=dev, generator=g)
    S = torch.empty((C, K), dtype=torch.float32, device=dev)
    lam = torch.full((V, K), 0.05, dtype=torch.float32, device=dev)
    for lo in range(0, C, chunk):
        hi = min(C, lo + chunk)
        n = hi - lo
        u1 = torch.rand((n, K), device=dev, generator=g).clamp_min_(1e-12)
        u2 = torch.rand((n, K), device=dev, generator=g).clamp_min_(1e-12)
        nb = torch.floor(-torch.log2(u1)) + torch.floor(-torch.log2(u2))          # NB(2, 0.5) = sum of 2 geometric(0.5)
        keep = (torch.rand((n, K), device=dev, generator=g) < 0.3).float()
        s = nb * keep
        S[lo:hi] = s
        lam.index_add_(0, assign[lo:hi], 0.5 * s)
    G = torch.poisson(lam, generator=g)
    # every gene must be expressed somewhere in both matrices (guard of mapping_utils.py:277)
    zs = (S.sum(0) == 0).nonzero().flatten()
    S[0, zs] = 1.0
    zg = (G.sum(0) == 0).nonzero().flatten()
    G[0, zg] = 1.0
    d = G.sum(1) / G.sum()                                                          # mapping_utils.py:88-89
    return dict(S=S, G=G, d=d, assign=assign)


def init_logits(C, V, device, seed=42):
    """Throughput-run initialisation: N(0,1) logits drawn on the device (the reference draws them on the host
    in float64, mapping_optimizer.py:147-157 -- use Mapper(random_state=...) for that bit-identical path)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randn((C, V), dtype=torch.float32, device=device, generator=g)


def hex_grid_graph(V):
    """Synthetic spot graph of BASELINE config 5b: V spots on a 2-D hexagonal grid (odd rows shifted by half a
    spot), 6 neighbours inside the grid -- what `sq.gr.spatial_neighbors` yields for Visium spots.  Returns scipy CSR
    (N, W): N = binary adjacency without self loops (`neighborhood_filter`, mapping_utils.py:324), W = row-normalised
    weights + identity (`voxel_weights`, mapping_utils.py:320; spatial_weights.py:14-28 with unit distances)."""
    import numpy as np
    import scipy.sparse as sp
    w = int(np.ceil(np.sqrt(V)))
    idx = np.arange(V)
    r, c = idx // w, idx % w
    rows, cols = [], []
    for dr, dc_even, dc_odd in ((0, -1, -1), (0, 1, 1), (-1, -1, 0), (-1, 0, 1), (1, -1, 0), (1, 0, 1)):
        rr = r + dr
        cc = c + np.where(r % 2 == 0, dc_even, dc_odd)
        j = rr * w + cc
        ok = (rr >= 0) & (cc >= 0) & (cc < w) & (j >= 0) & (j < V)
        rows.append(idx[ok])
        cols.append(j[ok])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    N = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(V, V))
    N.sum_duplicates()
    N.data[:] = 1.0
    rs = np.asarray(N.sum(axis=1)).reshape(-1)
    rs[rs == 0] = 1.0
    W = (sp.diags((1.0 / rs).astype(np.float32)) @ N + sp.identity(V, dtype=np.float32, format="csr")).tocsr()
    return N, W


def cell_type_encoding(assign, V, n_types):
    """One-hot cell types for config 5b: the type of a cell is the region label of its planted spot."""
    import numpy as np
    assign = np.asarray(assign)
    lab = (assign.astype(np.int64) * n_types) // V
    E = np.zeros((len(assign), n_types), np.float32)
    E[np.arange(len(assign)), lab] = 1.0
    return E
