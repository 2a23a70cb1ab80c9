"""Live differential tests against the UNMODIFIED reference optimizer, beyond the committed fixtures, on seeded random
configurations (shape, priors, every combination of loss terms incl. the spatial ones, both mapper classes):
 (1) the oracle vs the reference run in float64, started from the reference's own initial logits;
 (2) the product (the C-ABI library's kernels on the CPU emulator, fp32 and split-bf16 paths) vs the reference as shipped (fp32).

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


@pytest.mark.parametrize("seed", range(8))
def test_emulated_library_follows_the_live_reference_fp32(ref_mo, seed):
    """The product itself (the C-ABI library's kernels on the CPU emulator, fp32-parity paths) against the reference AS SHIPPED
    (float32, torch CPU) on seeded random configurations, within the stated fp32 tolerances of tests/parity_common.py -- the same
    comparison the golden fixtures make, on configurations no fixture spells out."""
    from tests.hipsim.build_sim import build_sim
    from tangram_amd import _capi
    from tests import parity_common as pc
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    from oracle import tangram_oracle as orc
    rng = np.random.default_rng(9000 + seed)
    C, K, V = int(rng.integers(2, 120)), int(rng.integers(1, 50)), int(rng.integers(4, 140))
    data = orc.make_synthetic(C, K, V, seed=seed, n_types=3)
    pick = lambda vals: float(rng.choice(vals))
    constrained = seed % 4 == 3
    n = 8
    _capi._install_library_for_tests(sim_path)
    try:
        from tangram_amd.engine import HipMapperEngine
        if constrained:
            lam = dict(lambda_d=pick([0.5, 1.0]), lambda_g1=1.0, lambda_g2=pick([0.0, 0.5]), lambda_r=pick([0.0, 1e-3]),
                       lambda_count=pick([0.5, 1.0]), lambda_f_reg=pick([0.5, 1.0]))
            tc = float(rng.integers(1, max(2, C)))
            m = ref_mo.MapperConstrained(S=data["S"], G=data["G"], d=data["d"], device="cpu", random_state=seed + 1, target_count=tc, **lam)
            M0, F0 = m.M.detach().numpy().copy(), m.F.detach().numpy().copy()
            P_ref, F_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
            mk = lambda p: HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device="cpu", precision=p,
                                           lambdas=lam, target_count=tc)
            ref_total = None                      # (stringified with 4 decimals: the mapping and the filter are compared instead)
        else:
            lam = dict(lambda_g1=1.0, lambda_d=pick([0.0, 1.0]), lambda_g2=pick([0.0, 0.5]), lambda_r=pick([0.0, 1e-3]),
                       lambda_l2=pick([0.0, 1e-5]), lambda_neighborhood_g1=pick([0.0, 0.96]), lambda_ct_islands=pick([0.0, 0.17]),
                       lambda_moran=pick([0.0, 0.4]))
            kw = {}
            d = data["d"] if lam["lambda_d"] > 0 else None
            if lam["lambda_neighborhood_g1"] > 0:
                kw["voxel_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=True)
            if lam["lambda_ct_islands"] > 0:
                kw["neighborhood_filter"] = orc.grid_graph(V, standardized=False, self_inclusion=False)
                kw["ct_encode"] = data["ct_encode"]
            if lam["lambda_moran"] > 0:
                kw["spatial_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=False)
            m = ref_mo.Mapper(S=data["S"], G=data["G"], d=d, device="cpu", random_state=seed + 1, **lam, **kw)
            M0 = m.M.detach().numpy().copy()
            P_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
            ref_total = np.array([float(x) for x in hist["total_loss"]], dtype=np.float64)
            F_ref = None
            mk = lambda p: HipMapperEngine(data["S"], data["G"], M0, d=d, device="cpu", precision=p, lambdas=lam, **kw)
        for prec in ("fp32", "bf16x3"):
            e = mk(prec)
            h = e.new_history(n)
            e.step(n, 0.1, h)
            tol = pc.TOL[prec]
            if ref_total is not None:
                err = np.abs(h[:, _capi.H_TOTAL].numpy().astype(np.float64) - ref_total).max()
                assert err <= 2 * tol["loss"] * max(1.0, np.abs(ref_total).max()), (prec, lam, err)
            out = e.result(with_filter=constrained)
            P = (out[0] if constrained else out).numpy()
            assert np.abs(P - P_ref).max() <= tol["P"], (prec, lam)
            if constrained:
                assert np.abs(out[1].numpy() - F_ref).max() <= 1e-4
    finally:
        _capi._install_library_for_tests(None)
