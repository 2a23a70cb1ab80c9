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
