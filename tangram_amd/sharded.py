"""Spot-sharded multi-GPU driver: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI), each rank owns a contiguous block of spots: M[:, V_g], its Adam moments, G[V_g, :], d[V_g].
S is replicated.  Per iteration exactly three small vectors cross GPUs:

    E2  per-gene cosine statistics [2][Kp]            all-reduce(sum)     after the forward GEMM
    E3  per-cell softmax-backward row dots [np][C]    all-reduce(sum)     after the first backward pass
    E1  per-cell (max, sum exp) of the new logits     all-gather + merge  after the Adam update

The gradient of M itself is column-sharded exactly like M and never leaves its GPU.  The reference has
no distributed code at all; this module is new capability, not a translation.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .engine import HipMapperEngine


class DistComm:
    """The collectives of the sharded step on a torch.distributed process group (RCCL on GPUs)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce(self, t):
        dist.all_reduce(t, group=self.group)

    def all_gather_into_tensor(self, out, t):
        dist.all_gather_into_tensor(out, t, group=self.group)

    def all_gather(self, outs, t):
        dist.all_gather(outs, t, group=self.group)


def shard_bounds(n, world, rank):
    """Contiguous, balanced partition of range(n) into `world` blocks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedMapperEngine:
    def __init__(self, S, G_local, M0_local, d_local=None, d_source=None, *, n_spots_total, device, precision="bf16x3",
                 lambdas=None, group=None, fwd_splits=0, tile_size=0, comm=None):
        # `comm`: anything with world, rank, all_reduce, all_gather_into_tensor, all_gather (tests drive several shards of one
        # GPU through an in-process communicator); default: the torch.distributed group
        self.comm = comm if comm is not None else DistComm(group)
        self.group = group
        self.world = self.comm.world
        self.rank = self.comm.rank
        self.lam = dict(lambda_g1=1.0, lambda_d=0.0, lambda_g2=0.0, lambda_r=0.0, lambda_l1=0.0, lambda_l2=0.0)
        self.lam.update(lambdas or {})
        self.eng = HipMapperEngine(S, G_local, M0_local, d=d_local, d_source=d_source, device=device,
                                   precision=precision, lambdas=self.lam, n_spots_total=n_spots_total,
                                   fwd_splits=fwd_splits, tile_size=tile_size)
        self.has_density = d_local is not None
        e = self.eng
        self.x_gene = e.exchange_buffer(_capi.X_GENESTAT)
        self.x_rowq = e.exchange_buffer(_capi.X_ROWQ)
        self.x_pair = e.exchange_buffer(_capi.X_ROWPAIR)
        self.gathered = torch.empty(self.world * self.x_pair.numel(), dtype=torch.float32, device=e.device)
        # set-up exchange: |G_k|^2 over all spots, then the softmax statistics of the initial logits
        self.comm.all_reduce(e.exchange_buffer(_capi.X_GNORM2))
        e.phase(0)
        self._exchange_row_stats()

    def _exchange_row_stats(self):
        self.comm.all_gather_into_tensor(self.gathered, self.x_pair)
        self.eng.phase(4, gathered=self.gathered, nranks=self.world)

    def step(self, lr, history_row=None):
        e = self.eng
        e.phase(1)
        self.comm.all_reduce(self.x_gene)
        e.phase(2, history_row=history_row)
        self.comm.all_reduce(self.x_rowq)
        e.phase(3, lr=lr, history_row=history_row)
        self._exchange_row_stats()

    def run(self, n_steps, lr, history=None, first_row=0):
        for i in range(n_steps):
            self.step(lr, history[first_row + i] if history is not None else None)

    def finalize_history(self, history):
        """Terms that are sums over spots were accumulated per shard: reduce them and recompose the total."""
        h = history.clone()
        add_cols = [_capi.H_VG, _capi.H_KL, _capi.H_ENTROPY, _capi.H_L1, _capi.H_L2]
        part = torch.nan_to_num(h[:, add_cols], nan=0.0)
        self.comm.all_reduce(part)
        lam = self.lam
        total = -lam["lambda_g1"] * h[:, _capi.H_MAIN]
        for col, key, sign in ((_capi.H_VG, "lambda_g2", -1.0), (_capi.H_KL, "lambda_d", 1.0),
                               (_capi.H_ENTROPY, "lambda_r", 1.0), (_capi.H_L1, "lambda_l1", 1.0),
                               (_capi.H_L2, "lambda_l2", 1.0)):
            j = add_cols.index(col)
            active = lam[key] != 0 and (col != _capi.H_KL or self.has_density)
            if active:
                h[:, col] = part[:, j]
                total = total + sign * lam[key] * part[:, j]
        h[:, _capi.H_TOTAL] = total
        return h

    def result_full(self):
        """All-gather the column blocks of softmax(M) -> [C, V_total] on every rank."""
        P_local = self.eng.result()
        sizes = [shard_bounds(self.eng.cfg.n_spots_total, self.world, r) for r in range(self.world)]
        widths = [b - a for a, b in sizes]
        wmax = max(widths)
        pad = torch.zeros((P_local.shape[0], wmax), dtype=torch.float32, device=P_local.device)
        pad[:, :P_local.shape[1]] = P_local
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.comm.all_gather(out, pad)
        return torch.cat([o[:, :w] for o, w in zip(out, widths)], dim=1)


def make_sharded(S, G, M0, d=None, d_source=None, *, device, precision="bf16x3", lambdas=None, group=None, fwd_splits=0, comm=None):
    """Slice full problem arrays (identical on every rank) into this rank's spot block."""
    comm = comm if comm is not None else DistComm(group)
    world, rank = comm.world, comm.rank
    V = G.shape[0]
    lo, hi = shard_bounds(V, world, rank)
    if hi - lo < 1:
        raise ValueError(f"rank {rank} would own no spots (V={V}, world={world})")
    G_l = G[lo:hi]
    M_l = M0[:, lo:hi]
    if isinstance(M_l, np.ndarray):
        M_l = np.ascontiguousarray(M_l)
    else:
        M_l = M_l.contiguous()
    d_l = None if d is None else d[lo:hi]
    return ShardedMapperEngine(S, G_l, M_l, d_l, d_source, n_spots_total=V, device=device, precision=precision,
                               lambdas=lambdas, group=group, fwd_splits=fwd_splits, comm=comm)
