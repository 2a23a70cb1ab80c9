"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A PyTorch-CPU *port* of the reference training loop with the same op sequence the reference
issues (softmax, matmul, cosine_similarity, KLDivLoss, autograd backward, torch.optim.Adam), so
that its wall time on the host CPU is representative of the reference's CPU path
(/root/reference/tangram/mapping_optimizer.py:189-309, :358-408; constrained :495-639).
/root/reference does not exist on the GPU box, so bench.py's `cpu_baseline` leg times this
port (`"kind": "port"`).  It is pinned to the reference by tests/test_oracle_golden.py through
the fixtures written by oracle/gen_golden.py.

It evaluates the lambda=0 regularisers exactly like the reference does (they dominate the
reference's CPU time), because the point of this file is to cost what the reference
costs, not to be fast.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.nn.functional import cosine_similarity, softmax


class TorchPortMapper:
    def __init__(self, S, G, d=None, d_source=None, lambda_g1=1.0, lambda_d=0.0, lambda_g2=0.0, lambda_r=0.0,
                 lambda_l1=0.0, lambda_l2=0.0, lambda_neighborhood_g1=0.0, voxel_weights=None,
                 lambda_ct_islands=0.0, neighborhood_filter=None, ct_encode=None,
                 M0=None, random_state=None, dtype=torch.float32):
        t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)
        self.S, self.G, self.d, self.d_source = t(S), t(G), t(d), t(d_source)
        self.W, self.N, self.E = t(voxel_weights), t(neighborhood_filter), t(ct_encode)
        self.lam = dict(g1=lambda_g1, d=lambda_d, g2=lambda_g2, r=lambda_r, l1=lambda_l1, l2=lambda_l2,
                        nb=lambda_neighborhood_g1, ct=lambda_ct_islands)
        if M0 is None:
            if random_state:
                np.random.seed(seed=random_state)
            M0 = np.random.normal(0, 1, (self.S.shape[0], self.G.shape[0]))
        self.M = torch.tensor(np.asarray(M0), dtype=dtype, requires_grad=True)
        self.kl = torch.nn.KLDivLoss(reduction="sum")

    def loss(self):
        lam = self.lam
        P = softmax(self.M, dim=1)
        Gp = P.t() @ self.S
        gv = lam["g1"] * cosine_similarity(Gp, self.G, dim=0).mean()
        vg = lam["g2"] * cosine_similarity(Gp, self.G, dim=1).mean()
        out = {"main_loss": (gv / lam["g1"]).tolist(),
               "vg_reg": (vg / lam["g2"]).tolist() if lam["g2"] else float("nan")}
        if self.d is not None:
            rho = (self.d_source @ P) if self.d_source is not None else P.sum(dim=0) / self.M.shape[0]
            dens = lam["d"] * self.kl(torch.log(rho), self.d)
            out["kl_reg"] = (dens / lam["d"]).tolist()
        else:
            dens, out["kl_reg"] = 0, float("nan")
        ent = lam["r"] * -(torch.log(P) * P).sum()
        out["entropy_reg"] = (ent / lam["r"]).tolist() if lam["r"] else float("nan")
        l1 = lam["l1"] * self.M.abs().sum()
        l2 = lam["l2"] * (self.M ** 2).sum()
        nb = 0
        if lam["nb"] > 0:
            nb = lam["nb"] * cosine_similarity(self.W @ Gp, self.W @ self.G, dim=0).mean()
        ct = 0
        if lam["ct"] > 0:
            cm = P.t() @ self.E
            ct = lam["ct"] * torch.max(cm - self.N @ cm, torch.zeros(1, dtype=cm.dtype)).mean()
        total = -(gv + vg) + dens + ent + l1 + l2 + ct - nb
        out["total_loss"] = float(total.detach())
        return total, out

    def train(self, num_epochs, learning_rate=0.1):
        opt = torch.optim.Adam([self.M], lr=learning_rate)
        hist = {k: [] for k in ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"]}
        for _ in range(num_epochs):
            total, out = self.loss()
            for k in hist:
                hist[k].append(out[k])
            opt.zero_grad()
            total.backward()
            opt.step()
        with torch.no_grad():
            return softmax(self.M, dim=1).numpy(), hist


class TorchPortMapperConstrained:
    def __init__(self, S, G, d, lambda_d=1.0, lambda_g1=1.0, lambda_g2=1.0, lambda_r=0.0, lambda_count=1.0,
                 lambda_f_reg=1.0, target_count=None, M0=None, F0=None, random_state=None, dtype=torch.float32):
        t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)
        self.S, self.G, self.d = t(S), t(G), t(d)
        self.lam = dict(d=lambda_d, g1=lambda_g1, g2=lambda_g2, r=lambda_r, c=lambda_count, f=lambda_f_reg)
        self.target = self.G.shape[0] if target_count is None else target_count
        if M0 is None or F0 is None:
            if random_state:
                np.random.seed(seed=random_state)
            np.random.normal(0, 1, (self.S.shape[0], self.G.shape[0]))
            M0 = np.random.normal(0, 1, (self.S.shape[0], self.G.shape[0]))
            F0 = np.random.normal(0, 1, self.S.shape[0])
        self.M = torch.tensor(np.asarray(M0), dtype=dtype, requires_grad=True)
        self.F = torch.tensor(np.asarray(F0), dtype=dtype, requires_grad=True)
        self.kl = torch.nn.KLDivLoss(reduction="sum")

    def loss(self):
        lam = self.lam
        P = softmax(self.M, dim=1)
        f = torch.sigmoid(self.F)
        out = {}
        total = 0
        if self.d is not None:
            rho = (P * f[:, None]).sum(dim=0) / f.sum()
            dens = lam["d"] * self.kl(torch.log(rho), self.d)
            out["kl_reg"] = (dens / lam["d"]).tolist()
            total = total + dens
        else:
            out["kl_reg"] = float("nan")
        Gp = P.t() @ (self.S * f[:, None])
        gv = lam["g1"] * cosine_similarity(Gp, self.G, dim=0).mean()
        vg = lam["g2"] * cosine_similarity(Gp, self.G, dim=1).mean()
        ent = lam["r"] * (torch.log(P) * P).sum()
        cnt = lam["c"] * torch.abs(f.sum() - self.target)
        freg = lam["f"] * (f - f * f).sum()
        total = total - (gv + vg) - ent + cnt + freg
        out.update(main_loss=(gv / lam["g1"]).tolist(),
                   vg_reg=(vg / lam["g2"]).tolist() if lam["g2"] else float("nan"),
                   entropy_reg=(ent / lam["r"]).tolist() if lam["r"] else float("nan"),
                   count_reg=(cnt / lam["c"]).tolist(), lambda_f_reg=(freg / lam["f"]).tolist(),
                   total_loss=float(total.detach()))
        return total, out

    def train(self, num_epochs, learning_rate=0.1):
        opt = torch.optim.Adam([self.M, self.F], lr=learning_rate)
        keys = ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg", "count_reg", "lambda_f_reg"]
        hist = {k: [] for k in keys}
        for _ in range(num_epochs):
            total, out = self.loss()
            for k in keys:
                hist[k].append(out[k])
            opt.zero_grad()
            total.backward()
            opt.step()
        with torch.no_grad():
            return softmax(self.M, dim=1).numpy(), torch.sigmoid(self.F).numpy(), hist
