"""The oracle (closed-form NumPy restatement + torch port) against fixtures produced by the
unmodified reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import tangram_oracle as orc
from oracle.gen_golden import CASES, build_inputs

NAMES = list(CASES)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _make(name, z, dtype):
    args, epochs, mode = build_inputs(name)
    args.pop("val_each", None)
    if mode == "constrained":
        m = orc.OracleMapperConstrained(M0=z["f32_M0"], F0=z["f32_F0"], dtype=dtype, **args)
    else:
        m = orc.OracleMapper(M0=z["f32_M0"], dtype=dtype, **args)
    return m, epochs, mode


@pytest.mark.parametrize("name", NAMES)
def test_init_matches_reference_rng(golden_dir, name):
    z = _load(golden_dir, name)
    C, K, V, seed, epochs, mode, kw = CASES[name]
    if mode == "constrained":
        M0, F0 = orc.reference_init_MF_constrained(C, V, 42)
        np.testing.assert_array_equal(F0, z["f32_F0"])
    else:
        M0 = orc.reference_init_M(C, V, 42)
    np.testing.assert_array_equal(M0, z["f32_M0"])


@pytest.mark.parametrize("name", NAMES)
def test_first_step_gradient_fp64(golden_dir, name):
    z = _load(golden_dir, name)
    m, _, mode = _make(name, z, np.float64)
    out = m.loss_and_grad()
    dM = out[1]
    ref = z["f64_dM0"]
    # the Geary term is a difference of large sums (the reference builds a V x V x K tensor): looser there
    rtol = 1e-7 if name == "cells_autocorr" else 1e-12
    assert np.abs(dM - ref).max() <= rtol * max(1.0, np.abs(ref).max()) + 1e-15
    if mode == "constrained":
        assert np.abs(out[2] - z["f64_dF0"]).max() <= 1e-12


@pytest.mark.parametrize("name", NAMES)
def test_trajectory_fp64(golden_dir, name):
    z = _load(golden_dir, name)
    m, epochs, mode = _make(name, z, np.float64)
    res = m.train(epochs, 0.1)
    hist = res[-1]
    # The 500-epoch grid cases (the reference's own test grid) leave the well-conditioned regime after ~170 epochs: the
    # REFERENCE's fp32 and fp64 runs then drift apart by up to 1.3e-4 in main_loss / kl_reg (they trade off along a flat
    # valley) while total_loss stays within 6e-6.  Individual terms are therefore pinned over the first WELL epochs, the
    # total over the whole run, the end point relative to the reference's own fp32-vs-fp64 spread.
    WELL = 150 if mode == "grid" else epochs
    grow = max(1.0, WELL / 50.0)            # round-off differences between two fp64 runs grow along the trajectory
    for k in hist:
        ref = z["f64_hist_" + k]
        got = np.array(hist[k])
        if np.isnan(ref).all():
            assert np.isnan(got).all(), k
        else:
            if mode == "constrained" and k == "total_loss":
                # the reference stores str(tensor) here (mapping_optimizer.py:630): 4 printed decimals
                np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4, err_msg=k)
            else:
                np.testing.assert_allclose(got[:WELL], ref[:WELL], rtol=0, atol=(5e-8 if name == "cells_autocorr" else 2e-9 * grow), err_msg=k)
                if k == "total_loss":
                    spread = float(np.abs(z["f32_hist_" + k] - ref).max())
                    np.testing.assert_allclose(got, ref, rtol=0, atol=max(2e-9 * grow, 0.1 * spread), err_msg=k + " (full run)")
    np.testing.assert_allclose(res[0], z["f64_P"], atol=(1e-6 if name == "cells_autocorr" else max(1e-7 * grow, (0.1 if mode == "grid" else 0.0) * float(np.abs(z["f32_P"] - z["f64_P"]).max()))))
    if mode == "constrained":
        np.testing.assert_allclose(res[1], z["f64_F_out"], atol=1e-7)


@pytest.mark.parametrize("name", NAMES)
def test_trajectory_fp32_vs_reference_fp32(golden_dir, name):
    """Same-precision comparison: tolerance = the reference's own fp32-vs-fp64 spread."""
    z = _load(golden_dir, name)
    m, epochs, mode = _make(name, z, np.float32)
    res = m.train(epochs, 0.1)
    hist = res[-1]
    for k in ("main_loss", "total_loss", "kl_reg"):
        ref = z["f32_hist_" + k]
        if np.isnan(ref).all():
            continue
        spread = float(np.abs(ref - z["f64_hist_" + k]).max())          # what fp32 costs the reference itself on this case
        tol = dict(rtol=1e-4, atol=1e-4) if (mode == "constrained" and k == "total_loss") else dict(rtol=0, atol=max(2e-5, 5 * spread))
        np.testing.assert_allclose(np.array(hist[k]), ref, err_msg=k, **tol)
    assert np.abs(res[0] - z["f32_P"]).max() < max(2e-4, 5 * float(np.abs(z["f32_P"] - z["f64_P"]).max()))


@pytest.mark.parametrize("name", ["cells_default", "cells_allreg", "cells_spatial", "constrained"])
def test_torch_port_matches_reference(golden_dir, name):
    from oracle import torch_port as tp
    z = _load(golden_dir, name)
    args, epochs, mode = build_inputs(name)
    if mode == "constrained":
        m = tp.TorchPortMapperConstrained(M0=z["f32_M0"], F0=z["f32_F0"], **args)
    else:
        m = tp.TorchPortMapper(M0=z["f32_M0"], **args)
    res = m.train(epochs, 0.1)
    hist = res[-1]
    for k in ("main_loss", "total_loss"):
        tol = dict(rtol=1e-4, atol=1e-4) if (mode == "constrained" and k == "total_loss") else dict(rtol=0, atol=2e-6)
        np.testing.assert_allclose(np.array(hist[k]), z["f32_hist_" + k], err_msg=k, **tol)
    assert np.abs(res[0] - z["f32_P"]).max() < 1e-5


def test_train_score_invariant():
    """Re-creation of the reference's test_train_score_match (tests/tangram_test.py:159-210):
    the reported main_loss equals the mean per-gene cosine recomputed from the returned mapping."""
    data = orc.make_synthetic(120, 30, 50, seed=11)
    m = orc.OracleMapper(data["S"], data["G"], d=data["d"], lambda_d=1, random_state=42)
    P, hist = m.train(30)
    m2 = orc.OracleMapper(data["S"], data["G"], d=data["d"], lambda_d=1, M0=m.M)
    terms, _ = m2.loss_and_grad()
    Gp = P.astype(np.float64).T @ data["S"].astype(np.float64)
    G = data["G"].astype(np.float64)
    cs = [(a @ b) / (np.linalg.norm(a) * np.linalg.norm(b)) for a, b in zip(G.T, Gp.T)]
    assert round(float(np.mean(cs)), 3) == round(float(terms["main_loss"]), 3)


def test_closed_form_gradient_equals_autograd_on_random_configurations():
    """Property test (hypothesis): for random shapes, priors and hyper-parameters the closed-form gradient of the NumPy
    oracle equals torch autograd on the port with the reference's op sequence (both in float64), and three Adam steps
    agree.  Covers combinations the fixtures do not (e.g. d_source with regularisers, lambda_g2 = 0 with spatial terms)."""
    import torch
    from hypothesis import given, settings, strategies as st, HealthCheck
    from oracle import torch_port as tp

    lam_st = st.fixed_dictionaries(dict(
        lambda_g2=st.sampled_from([0.0, 0.3, 1.0]), lambda_d=st.sampled_from([0.0, 0.5, 1.0, 2.0]),
        lambda_r=st.sampled_from([0.0, 1e-3, 1e-2]), lambda_l1=st.sampled_from([0.0, 1e-4]),
        lambda_l2=st.sampled_from([0.0, 1e-5]), lambda_neighborhood_g1=st.sampled_from([0.0, 0.96]),
        lambda_ct_islands=st.sampled_from([0.0, 0.17])))

    @settings(max_examples=25, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.too_slow])
    @given(C=st.integers(2, 40), K=st.integers(1, 12), V=st.integers(2, 30), seed=st.integers(0, 10_000),
           use_dsource=st.booleans(), lam=lam_st)
    def check(C, K, V, seed, use_dsource, lam):
        data = orc.make_synthetic(C, K, V, seed=seed, n_types=3)
        rng = np.random.default_rng(seed)
        M0 = rng.normal(size=(C, V)).astype(np.float32)      # (the reference keeps M as a float32 leaf)
        kw = dict(lam)
        d = data["d"] if kw["lambda_d"] > 0 else None
        kw["d"] = d
        if use_dsource and d is not None:
            ds = rng.random(C) + 0.1
            kw["d_source"] = (ds / ds.sum()).astype(np.float32)
        if kw["lambda_neighborhood_g1"] > 0:
            kw["voxel_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=True)
        if kw["lambda_ct_islands"] > 0:
            kw["neighborhood_filter"] = orc.grid_graph(V, standardized=False, self_inclusion=False)
            kw["ct_encode"] = data["ct_encode"]
        o = orc.OracleMapper(data["S"], data["G"], M0=M0, dtype=np.float64, **kw)
        terms, dM = o.loss_and_grad()
        t = tp.TorchPortMapper(data["S"], data["G"], M0=M0, dtype=torch.float64, **kw)
        total, out = t.loss()
        total.backward()
        g = t.M.grad.numpy()
        assert abs(terms["total_loss"] - out["total_loss"]) <= 1e-10 * max(1.0, abs(out["total_loss"]))
        assert np.abs(dM - g).max() <= 1e-10 * max(1.0, np.abs(g).max())
        t2 = tp.TorchPortMapper(data["S"], data["G"], M0=M0, dtype=torch.float64, **kw)
        P_o, h_o = o.train(3, 0.1)
        P_t, h_t = t2.train(3, 0.1)
        np.testing.assert_allclose(h_o["total_loss"], h_t["total_loss"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(P_o, P_t, atol=1e-7)

    check()
