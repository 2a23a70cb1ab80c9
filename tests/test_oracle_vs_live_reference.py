"""Live differential test of the oracle against the UNMODIFIED reference optimizer, beyond the committed fixtures: seeded random
configurations (shape, priors, every combination of loss terms incl. the spatial ones, both mapper classes), the reference run in
float64 on the CPU, the oracle started from the reference's own initial logits.

Only where the reference checkout exists (the authoring container): /root/reference is absent on the GPU box, so the module skips
itself there -- like oracle/gen_golden.py, which produced tests/golden/ the same way."""
import os

import numpy as np
import pytest

REF = "/root/reference/tangram/mapping_optimizer.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is not present on this machine")


@pytest.fixture(scope="module")
def ref_mo():
    import torch
    from oracle.gen_golden import load_ref
    torch.set_num_threads(1)
    return load_ref()


@pytest.mark.parametrize("seed", range(12))
def test_oracle_follows_the_live_reference(ref_mo, seed):
    from oracle import tangram_oracle as orc
    from oracle.gen_golden import to_double
    rng = np.random.default_rng(7000 + seed)
    C, K, V = int(rng.integers(2, 90)), int(rng.integers(1, 40)), int(rng.integers(4, 70))
    data = orc.make_synthetic(C, K, V, seed=seed, n_types=3)
    pick = lambda vals: float(rng.choice(vals))
    constrained = seed % 3 == 2
    n = 6
    if constrained:
        lam = dict(lambda_d=pick([0.5, 1.0]), lambda_g1=1.0, lambda_g2=pick([0.0, 0.5, 1.0]), lambda_r=pick([0.0, 1e-3]),
                   lambda_count=pick([0.5, 1.0]), lambda_f_reg=pick([0.5, 1.0]), target_count=float(rng.integers(1, max(2, C))))
        m = ref_mo.MapperConstrained(S=data["S"], G=data["G"], d=data["d"], device="cpu", random_state=seed + 1, **lam)
        to_double(m)
        M0, F0 = m.M.detach().numpy().copy(), m.F.detach().numpy().copy()
        P_ref, F_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
        parse = lambda s: float(str(s).replace("tensor(", "").split(",")[0].rstrip(")"))
        ref_total = np.array([parse(x) for x in hist["total_loss"]])
        ref_main = np.array([parse(x) for x in hist["main_loss"]])
        o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, dtype=np.float64, **lam)
        P, F, ho = o.train(n, 0.1)
        np.testing.assert_allclose(F, F_ref, atol=2e-8)
    else:
        lam = dict(lambda_g1=1.0, lambda_d=pick([0.0, 0.5, 1.0]), lambda_g2=pick([0.0, 0.4, 1.0]), lambda_r=pick([0.0, 1e-3]),
                   lambda_l1=pick([0.0, 1e-4]), lambda_l2=pick([0.0, 1e-5]), lambda_neighborhood_g1=pick([0.0, 0.96]),
                   lambda_ct_islands=pick([0.0, 0.17]), lambda_getis_ord=pick([0.0, 0.5]), lambda_moran=pick([0.0, 0.4]),
                   lambda_geary=pick([0.0, 0.3]))
        kw = {}
        d = data["d"] if lam["lambda_d"] > 0 else None
        if d is not None and rng.random() < 0.5:
            ds = (rng.random(C) + 0.1).astype(np.float32)
            kw["d_source"] = ds / ds.sum()
        if lam["lambda_neighborhood_g1"] > 0:
            kw["voxel_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=True)
        if lam["lambda_ct_islands"] > 0:
            kw["neighborhood_filter"] = orc.grid_graph(V, standardized=False, self_inclusion=False)
            kw["ct_encode"] = data["ct_encode"]
        if lam["lambda_getis_ord"] > 0 or lam["lambda_moran"] > 0 or lam["lambda_geary"] > 0:
            kw["spatial_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=False)
        m = ref_mo.Mapper(S=data["S"], G=data["G"], d=d, device="cpu", random_state=seed + 1, **lam, **kw)
        to_double(m)
        for nm in ("getis_ord_G_star_ref", "moran_I_ref", "gearys_C_ref"):          # reference indicators, computed in fp32 at construction
            if getattr(m, nm, None) is not None:
                setattr(m, nm, getattr(m, nm).double())
        M0 = m.M.detach().numpy().copy()
        P_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
        ref_total = np.array([float(x) for x in hist["total_loss"]])
        ref_main = np.array([float(x) for x in hist["main_loss"]])
        o = orc.OracleMapper(data["S"], data["G"], d=d, M0=M0, dtype=np.float64, **lam, **kw)
        P, ho = o.train(n, 0.1)
    scale = max(1.0, np.abs(ref_total).max())
    # (MapperConstrained stringifies its history, mapping_optimizer.py:630: str(tensor) keeps 4 decimals -- P and F carry the precision)
    tol = 6e-5 if constrained else 2e-8
    np.testing.assert_allclose(np.array(ho["total_loss"], dtype=np.float64), ref_total, atol=tol * scale, rtol=0, err_msg=str(lam))
    np.testing.assert_allclose(np.array(ho["main_loss"], dtype=np.float64), ref_main, atol=tol, rtol=0, err_msg=str(lam))
    np.testing.assert_allclose(P, P_ref, atol=1e-7, err_msg=str(lam))       # (the reference builds its spatial indicators of G in fp32)
