"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Closed-form NumPy restatement of one Tangram mapping iteration
(softmax -> P^T S -> cosine/density/regulariser losses -> analytic backward -> Adam),
following the reference implementation line by line:

    /root/reference/tangram/mapping_optimizer.py
        Mapper.__init__            :19-157   (initial M, density flags)
        Mapper._loss_fn            :189-309  (forward terms, total loss :266-270)
        Mapper.train               :358-408  (Adam loop, history keys :378-392)
        MapperConstrained._loss_fn :495-587
        MapperConstrained.train    :589-639
    torch/optim/adam.py::_single_tensor_adam (update order)
    torch cosine_similarity (installed torch 2.10): x.y / (max(|x|,eps) * max(|y|,eps)), eps = 1e-8
    torch.nn.KLDivLoss(reduction="sum")(logq, p) = sum xlogy(p, p) - p * logq

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity status: PINNED against the live reference.  `oracle/gen_golden.py` imports the
unmodified reference optimizer (standalone importlib load, CPU, fp32 and fp64) in the authoring
container and stores its outputs in tests/golden/*.npz; tests/test_oracle_golden.py checks this
restatement against those fixtures.  The reference's own known-answer tests
(tests/tangram_test.py:67-103,159-210) need data/test_ad_*.h5ad which are absent from the
checkout (.MISSING_LARGE_BLOBS), so they cannot be replayed; see DESIGN.md.

The arithmetic dtype is a parameter (float64 = ground truth, float32 = same-precision
comparison).  Everything is dense NumPy; sizes are expected to be "finishes in seconds".
"""
from __future__ import annotations

import numpy as np

EPS_COS = 1e-8  # torch.nn.functional.cosine_similarity default eps

HISTORY_KEYS = ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"]       # mapping_optimizer.py:378
HISTORY_KEYS_CONSTRAINED = HISTORY_KEYS + ["count_reg", "lambda_f_reg"]             # mapping_optimizer.py:609-617


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def softmax_rows(M):
    """softmax(M, dim=1)  (mapping_optimizer.py:201)."""
    mx = M.max(axis=1, keepdims=True)
    E = np.exp(M - mx)
    return E / E.sum(axis=1, keepdims=True)


def cos_stats(A, B, axis):
    """dot, |A|, |B| (clamped like torch) along `axis`."""
    dot = (A * B).sum(axis=axis)
    na = np.maximum(np.sqrt((A * A).sum(axis=axis)), EPS_COS)
    nb = np.maximum(np.sqrt((B * B).sum(axis=axis)), EPS_COS)
    return dot, na, nb


def cos_mean_and_grad(A, B, axis):
    """mean over the *other* axis of cos(A, B) along `axis`, and d(mean)/dA.

    cosgrad(A,B)[.] = ( B/(|A||B|) - cos * A/|A|^2 ) / n_vectors     
    """
    dot, na, nb = cos_stats(A, B, axis)
    cos = dot / (na * nb)
    n = cos.size
    if axis == 0:
        g = (B / (na * nb)[None, :] - A * (cos / (na * na))[None, :]) / n
    else:
        g = (B / (na * nb)[:, None] - A * (cos / (na * na))[:, None]) / n
    return cos.mean(), g, cos


def xlogy(x, y):
    out = np.zeros_like(x)
    nz = x != 0
    out[nz] = x[nz] * np.log(y[nz])
    return out


def adam_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update, step count t >= 1 (in place)."""
    m += (g - m) * (1.0 - beta1)                      # exp_avg.lerp_(grad, 1-beta1)
    v *= beta2
    v += (1.0 - beta2) * g * g                        # exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    bc1 = 1.0 - beta1 ** t
    bc2_sqrt = np.sqrt(1.0 - beta2 ** t)
    step_size = lr / bc1
    denom = np.sqrt(v) / p.dtype.type(bc2_sqrt) + p.dtype.type(eps)
    p -= p.dtype.type(step_size) * (m / denom)        # param.addcdiv_(exp_avg, denom, value=-step_size)


def reference_init_M(C, V, random_state):
    """Initial logits exactly as mapping_optimizer.py:147-157: host float64 normal draw -> float32."""
    if random_state:
        np.random.seed(seed=random_state)
    return np.random.normal(0, 1, (C, V)).astype(np.float32)


def reference_init_MF_constrained(C, V, random_state):
    """MapperConstrained init (mapping_optimizer.py:472-493): M is drawn twice (second wins), then F."""
    if random_state:
        np.random.seed(seed=random_state)
    np.random.normal(0, 1, (C, V))
    M = np.random.normal(0, 1, (C, V)).astype(np.float32)
    F = np.random.normal(0, 1, C).astype(np.float32)
    return M, F


# ----------------------------------------------------------------------------------------------
# Mapper (modes 'cells' / 'clusters')
# ----------------------------------------------------------------------------------------------
class OracleMapper:
    """Closed-form restatement of `Mapper` (mapping_optimizer.py:14-408)."""

    def __init__(self, S, G, d=None, d_source=None, lambda_g1=1.0, lambda_d=0.0, lambda_g2=0.0,
                 lambda_r=0.0, lambda_l1=0.0, lambda_l2=0.0,
                 lambda_neighborhood_g1=0.0, voxel_weights=None,
                 lambda_ct_islands=0.0, neighborhood_filter=None, ct_encode=None,
                 lambda_getis_ord=0.0, lambda_moran=0.0, lambda_geary=0.0, spatial_weights=None,
                 M0=None, random_state=None, dtype=np.float64):
        self.dt = np.dtype(dtype)
        f = self.dt.type
        self.S = np.asarray(S, dtype=np.float32).astype(self.dt)
        self.G = np.asarray(G, dtype=np.float32).astype(self.dt)
        self.C, self.K = self.S.shape
        self.V = self.G.shape[0]
        self.d = None if d is None else np.asarray(d, dtype=np.float32).astype(self.dt)
        self.d_source = None if d_source is None else np.asarray(d_source, dtype=np.float32).astype(self.dt)
        self.lg1, self.ld, self.lg2 = f(lambda_g1), f(lambda_d), f(lambda_g2)
        self.lr_, self.ll1, self.ll2 = f(lambda_r), f(lambda_l1), f(lambda_l2)
        self.lnb, self.lct = f(lambda_neighborhood_g1), f(lambda_ct_islands)
        self.W = None if voxel_weights is None else np.asarray(voxel_weights, dtype=np.float32).astype(self.dt)
        self.N = None if neighborhood_filter is None else np.asarray(neighborhood_filter, dtype=np.float32).astype(self.dt)
        self.E = None if ct_encode is None else np.asarray(ct_encode, dtype=np.float32).astype(self.dt)
        if M0 is None:
            M0 = reference_init_M(self.C, self.V, random_state)
        self.M = np.asarray(M0, dtype=np.float32).astype(self.dt)
        self.m = np.zeros_like(self.M)
        self.v = np.zeros_like(self.M)
        self.t = 0
        if self.W is not None:
            self.WG = self.W @ self.G          # constant (mapping_optimizer.py:236 recomputes it)
        # spatial autocorrelation terms (mapping_optimizer.py:159-187, :251-263)
        self.lgo, self.lmo, self.lge = f(lambda_getis_ord), f(lambda_moran), f(lambda_geary)
        self.Ws = None if spatial_weights is None else np.asarray(spatial_weights, dtype=np.float32).astype(self.dt)
        if self.lgo > 0 or self.lmo > 0 or self.lge > 0:
            self.ref_getis, self.ref_moran, self.ref_geary = self._indicators(self.G)     # :144

    def _indicators(self, X):
        """_spatial_local_indicators (mapping_optimizer.py:159-187) on a V x K matrix."""
        V = X.shape[0]
        Ws = self.Ws
        getis = moran = geary = None
        if self.lgo > 0:
            getis = (Ws @ X) / X.sum(axis=0)                                   # :171
        if self.lmo > 0:
            z = X - X.mean(axis=0)
            moran = (V * z * (Ws @ z)) / (z * z).sum(axis=0)                    # :175-176
        if self.lge > 0:
            m2 = ((X - X.mean(axis=0)) ** 2).sum(axis=0) / (V - 1)              # :181
            r, c = Ws.sum(axis=1), Ws.sum(axis=0)
            A = (c[:, None] * X * X).sum(axis=0) + (r[:, None] * X * X).sum(axis=0) - 2 * (X * (Ws @ X)).sum(axis=0)
            geary = A / (2 * m2)                                                # :182-185 (sum_ij w_ij (x_j - x_i)^2)
        return getis, moran, geary

    def _autocorr_terms(self, Ghat):
        """values and d(-lambda * term)/dGhat of the Getis-Ord / Moran / Geary terms (:251-263, total :266-270)."""
        V = Ghat.shape[0]
        Ws = self.Ws
        dG = np.zeros_like(Ghat)
        vals = {}
        if self.lgo > 0:
            s = Ghat.sum(axis=0)
            Y = Ws @ Ghat
            pred = Y / s
            val, g, _ = cos_mean_and_grad(pred, self.ref_getis, 0)             # cos is symmetric in its arguments
            vals["getis"] = val
            dY = g / s
            ds = -(g * Y).sum(axis=0) / (s * s)
            dG += -self.lgo * (Ws.T @ dY + ds[None, :])
        if self.lmo > 0:
            mu = Ghat.mean(axis=0)
            z = Ghat - mu
            u = Ws @ z
            q = (z * z).sum(axis=0)
            I = V * z * u / q
            val, g, _ = cos_mean_and_grad(I, self.ref_moran, 0)
            vals["moran"] = val
            dq = -(g * V * z * u).sum(axis=0) / (q * q)
            dz = V * g * u / q + Ws.T @ (V * g * z / q) + 2 * z * dq[None, :]
            dG += -self.lmo * (dz - dz.mean(axis=0))
        if self.lge > 0:
            mu = Ghat.mean(axis=0)
            xc = Ghat - mu
            m2 = (xc * xc).sum(axis=0) / (V - 1)
            r, c = Ws.sum(axis=1), Ws.sum(axis=0)
            Y, Z = Ws @ Ghat, Ws.T @ Ghat
            A = ((c + r)[:, None] * Ghat * Ghat).sum(axis=0) - 2 * (Ghat * Y).sum(axis=0)
            p = A / (2 * m2)
            ref = self.ref_geary
            npn = max(np.sqrt((p * p).sum()), EPS_COS)
            nrn = max(np.sqrt((ref * ref).sum()), EPS_COS)
            cosv = (p * ref).sum() / (npn * nrn)
            vals["geary"] = cosv
            gp = ref / (npn * nrn) - cosv * p / (npn * npn)                     # d cos / d p_k
            dA = 2 * ((c + r)[:, None] * Ghat - Z - Y)
            dm2 = 2 * xc / (V - 1)
            dp = dA / (2 * m2) - (A / (2 * m2 * m2)) * dm2
            dG += -self.lge * gp[None, :] * dp
        return vals, dG

    # -- forward + analytic backward -----------------------------------------------------------
    def loss_and_grad(self):
        f = self.dt.type
        M, S, G = self.M, self.S, self.G
        C, V, K = self.C, self.V, self.K
        P = softmax_rows(M)                                    # :201
        Ghat = P.T @ S                                         # :202
        terms = {}

        gv, g0, _ = cos_mean_and_grad(Ghat, G, 0)              # :205
        vg, g1, _ = cos_mean_and_grad(Ghat, G, 1)              # :206 (always evaluated)
        dGhat = -self.lg1 * g0
        if self.lg2 != 0:
            dGhat = dGhat - self.lg2 * g1
        terms["main_loss"] = gv                                # gv_term / lambda_g1   (:208)
        terms["vg_reg"] = vg if self.lg2 != 0 else np.nan      # 0*x/0 -> nan         (:209)
        total = -(self.lg1 * gv) - (self.lg2 * vg)

        dP = np.zeros_like(P)
        if self.d is not None:                                 # :212-221
            if self.d_source is not None:
                rho = self.d_source @ P                        # :215
                w_c = self.d_source
            else:
                rho = P.sum(axis=0) / f(C)                     # :217
                w_c = np.full(C, f(1.0) / f(C), dtype=self.dt)
            kl = (xlogy(self.d, self.d) - self.d * np.log(rho)).sum()
            terms["kl_reg"] = kl
            total = total + self.ld * kl
            dP += self.ld * (-(self.d / rho))[None, :] * w_c[:, None]
        else:
            terms["kl_reg"] = np.nan

        if self.lr_ != 0:                                      # :224-225
            logP = np.log(P)
            ent = -(logP * P).sum()
            terms["entropy_reg"] = ent
            total = total + self.lr_ * ent
            dP += -self.lr_ * (logP + 1)
        else:
            terms["entropy_reg"] = np.nan

        dM_extra = None
        if self.ll1 != 0:                                      # :228
            total = total + self.ll1 * np.abs(M).sum()
            dM_extra = self.ll1 * np.sign(M)
        if self.ll2 != 0:                                      # :230
            total = total + self.ll2 * (M * M).sum()
            dM_extra = (0 if dM_extra is None else dM_extra) + 2 * self.ll2 * M

        if self.lnb > 0:                                       # :234-239
            WGhat = self.W @ Ghat
            nb, gnb, _ = cos_mean_and_grad(WGhat, self.WG, 0)
            total = total - self.lnb * nb
            dGhat = dGhat - self.lnb * (self.W.T @ gnb)
            terms["nb_sim"] = nb
        if self.lct > 0:                                       # :242-248
            ct = P.T @ self.E
            D = ct - self.N @ ct
            isl = np.maximum(D, 0).mean()
            total = total + self.lct * isl
            mask = (D > 0).astype(self.dt) / f(D.size)
            dct = mask - self.N.T @ mask
            dP += self.lct * (self.E @ dct.T)
            terms["ct_island"] = isl

        if self.lgo > 0 or self.lmo > 0 or self.lge > 0:       # :251-263; total -= each term (:270)
            vals, dGa = self._autocorr_terms(Ghat)
            dGhat = dGhat + dGa
            for nm, lamv in (("getis", self.lgo), ("moran", self.lmo), ("geary", self.lge)):
                if nm in vals:
                    total = total - lamv * vals[nm]
                    terms[nm + "_sim"] = vals[nm]

        dP += S @ dGhat.T                                      # second GEMM (autograd of :202)
        r = (P * dP).sum(axis=1, keepdims=True)
        dM = P * (dP - r)                                      # softmax backward
        if dM_extra is not None:
            dM = dM + dM_extra
        terms["total_loss"] = total
        self.last = dict(P=P, Ghat=Ghat, dGhat=dGhat, dM=dM, r=r[:, 0])
        return terms, dM

    def step(self, learning_rate=0.1):
        terms, dM = self.loss_and_grad()
        self.t += 1
        adam_update(self.M, dM, self.m, self.v, self.t, learning_rate)
        return terms

    def train(self, num_epochs, learning_rate=0.1):
        hist = {k: [] for k in HISTORY_KEYS}
        for _ in range(num_epochs):
            terms = self.step(learning_rate)
            for k in HISTORY_KEYS:
                hist[k].append(float(terms[k]))
        return softmax_rows(self.M).astype(np.float32), hist

    def project(self):
        """P^T S with the current logits (what mapping_utils.py:402 recomputes on the host)."""
        return softmax_rows(self.M).T @ self.S


# ----------------------------------------------------------------------------------------------
# MapperConstrained
# ----------------------------------------------------------------------------------------------
class OracleMapperConstrained:
    """Closed-form restatement of `MapperConstrained` (mapping_optimizer.py:411-639)."""

    def __init__(self, S, G, d, lambda_d=1.0, lambda_g1=1.0, lambda_g2=1.0, lambda_r=0.0,
                 lambda_count=1.0, lambda_f_reg=1.0, target_count=None,
                 M0=None, F0=None, random_state=None, dtype=np.float64):
        self.dt = np.dtype(dtype)
        f = self.dt.type
        self.S = np.asarray(S, dtype=np.float32).astype(self.dt)
        self.G = np.asarray(G, dtype=np.float32).astype(self.dt)
        self.C, self.K = self.S.shape
        self.V = self.G.shape[0]
        self.d = None if d is None else np.asarray(d, dtype=np.float32).astype(self.dt)
        self.ld, self.lg1, self.lg2, self.lr_ = f(lambda_d), f(lambda_g1), f(lambda_g2), f(lambda_r)
        self.lc, self.lf = f(lambda_count), f(lambda_f_reg)
        self.target = f(self.V if target_count is None else target_count)   # :480-483
        if M0 is None or F0 is None:
            M0, F0 = reference_init_MF_constrained(self.C, self.V, random_state)
        self.M = np.asarray(M0, dtype=np.float32).astype(self.dt)
        self.F = np.asarray(F0, dtype=np.float32).astype(self.dt)
        self.mM, self.vM = np.zeros_like(self.M), np.zeros_like(self.M)
        self.mF, self.vF = np.zeros_like(self.F), np.zeros_like(self.F)
        self.t = 0

    def loss_and_grad(self):
        M, S, G = self.M, self.S, self.G
        P = softmax_rows(M)                                    # :506
        fp = 1.0 / (1.0 + np.exp(-self.F))                     # :507
        Ghat = P.T @ (S * fp[:, None])                         # :519-521
        gv, g0, _ = cos_mean_and_grad(Ghat, G, 0)              # :522
        vg, g1, _ = cos_mean_and_grad(Ghat, G, 1)              # :523
        dGhat = -self.lg1 * g0 - self.lg2 * g1
        terms = {"main_loss": gv, "vg_reg": vg if self.lg2 != 0 else np.nan}
        total = -(self.lg1 * gv) - (self.lg2 * vg)

        X = S @ dGhat.T                                        # C x V
        dP = X * fp[:, None]
        df = (P * X).sum(axis=1)

        if self.d is not None:                                 # :511-515
            colsum = (P * fp[:, None]).sum(axis=0)
            fsum = fp.sum()
            rho = colsum / fsum
            kl = (xlogy(self.d, self.d) - self.d * np.log(rho)).sum()
            terms["kl_reg"] = kl
            total = total + self.ld * kl
            a = -(self.d / colsum)
            dP += self.ld * a[None, :] * fp[:, None]
            df += self.ld * ((P * a[None, :]).sum(axis=1) + self.d.sum() / fsum)
        else:
            terms["kl_reg"] = np.nan

        if self.lr_ != 0:                                      # :526, total has -entropy_term (:575)
            logP = np.log(P)
            ent = (logP * P).sum()
            terms["entropy_reg"] = ent
            total = total - self.lr_ * ent
            dP += -self.lr_ * (logP + 1)
        else:
            terms["entropy_reg"] = np.nan

        cnt = fp.sum() - self.target                           # :528-529
        terms["count_reg"] = np.abs(cnt)
        total = total + self.lc * np.abs(cnt)
        df += self.lc * np.sign(cnt)
        freg = (fp - fp * fp).sum()                            # :531-532
        terms["lambda_f_reg"] = freg
        total = total + self.lf * freg
        df += self.lf * (1 - 2 * fp)

        r = (P * dP).sum(axis=1, keepdims=True)
        dM = P * (dP - r)
        dF = df * fp * (1 - fp)
        terms["total_loss"] = total
        self.last = dict(P=P, Ghat=Ghat, dGhat=dGhat, dM=dM, dF=dF, f=fp)
        return terms, dM, dF

    def step(self, learning_rate=0.1):
        terms, dM, dF = self.loss_and_grad()
        self.t += 1
        adam_update(self.M, dM, self.mM, self.vM, self.t, learning_rate)
        adam_update(self.F, dF, self.mF, self.vF, self.t, learning_rate)
        return terms

    def train(self, num_epochs, learning_rate=0.1):
        hist = {k: [] for k in HISTORY_KEYS_CONSTRAINED}
        for _ in range(num_epochs):
            terms = self.step(learning_rate)
            for k in HISTORY_KEYS_CONSTRAINED:
                hist[k].append(float(terms[k]))
        fp = 1.0 / (1.0 + np.exp(-self.F))
        return softmax_rows(self.M).astype(np.float32), fp.astype(np.float32), hist


# ----------------------------------------------------------------------------------------------
# synthetic workloads: planted mapping, count-like S, Poisson G
# ----------------------------------------------------------------------------------------------
def make_synthetic(C, K, V, seed=0, n_types=0):
    """Planted-assignment synthetic data with the shape statistics of SURVEY 8(d).

    Returns dict(S [C,K] f32, G [V,K] f32, d [V] f32 (rna_count_based density), assign [C],
                 ct_encode [C,T] or None)
    """
    rng = np.random.default_rng(seed)
    assign = rng.integers(0, V, size=C)
    S = (rng.negative_binomial(2, 0.5, size=(C, K)) * (rng.random((C, K)) < 0.3)).astype(np.float32)
    for k in np.where(~S.any(axis=0))[0]:
        S[rng.integers(0, C), k] = 1.0
    lam = np.full((V, K), 0.05, dtype=np.float64)
    np.add.at(lam, assign, 0.5 * S)
    G = rng.poisson(lam).astype(np.float32)
    for k in np.where(~G.any(axis=0))[0]:
        G[rng.integers(0, V), k] = 1.0
    d = (G.sum(axis=1) / G.sum()).astype(np.float32)           # mapping_utils.py:88-89
    out = dict(S=S, G=G, d=d, assign=assign, ct_encode=None)
    if n_types:
        lab = (assign * n_types) // V
        E = np.zeros((C, n_types), dtype=np.float32)
        E[np.arange(C), lab] = 1.0
        out["ct_encode"] = E
    return out


def grid_graph(V, standardized, self_inclusion):
    """Dense V x V weights of a 2-D 4-neighbour grid, mimicking spatial_weights.py:5-29."""
    w = int(np.ceil(np.sqrt(V)))
    W = np.zeros((V, V), dtype=np.float32)
    for i in range(V):
        r, c = divmod(i, w)
        for dr, dc in ((0, 1), (1, 0), (0, -1), (-1, 0)):
            rr, cc = r + dr, c + dc
            j = rr * w + cc
            if 0 <= rr and 0 <= cc < w and j < V:
                W[i, j] = 1.0
    if standardized:
        rs = W.sum(axis=1, keepdims=True)
        rs[rs == 0] = 1
        W = W / rs
    if self_inclusion:
        W = W + np.eye(V, dtype=np.float32)
    return W
