"""The opt-in two-product split-bf16 path for a bf16-exact S (PrecBF16x2S, tg_config.s_exact_mode / `s_exact="auto"`).

In both GEMMs of the iteration (mapping_optimizer.py:202 forward, its autograd backward :395) the constant operand is S.  When
every element of S is exactly representable in bf16 its lo part is zero and the product a_hi * S_lo of the three-product scheme
adds exact zeros; skipping it must change NOTHING: every test here compares with the general path for EQUALITY (==, which only
forgives the sign of an exact zero), on the CPU emulator (same kernel sources).  The library must also refuse the shortcut by
itself when S, d_source or the cell-type encoding is not exact."""
import numpy as np
import pytest

from tests.hipsim.build_sim import build_sim


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


def _run(data, M0, n, s_exact, tile=0, mode="mapper", F0=None, lam=None, **kw):
    from tangram_amd.engine import HipMapperEngine
    lam = lam or dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode=mode, device="cpu", precision="bf16x3", lambdas=lam,
                        tile_size=tile, s_exact=s_exact, **kw)
    h = e.new_history(n)
    e.step(n, 0.1, h)
    out = dict(P=e.result().numpy(), h=h.numpy(), G=e.project().numpy(), eff=e.effective_precision)
    M, m1, m2, _ = e.logits()
    out.update(M=M.numpy().copy(), m1=m1.numpy().copy())
    return out


@pytest.mark.parametrize("shape,tile", [((300, 70, 200), 128), ((131, 37, 53), 128), ((520, 40, 300), 256)])
def test_two_products_equal_three_on_count_data(sim, shape, tile):
    from oracle import tangram_oracle as orc
    C, K, V = shape
    data = orc.make_synthetic(C, K, V, seed=3)                   # S: negative-binomial counts -- bf16-exact
    assert data["S"].max() < 256 and (data["S"] == np.round(data["S"])).all()
    M0 = orc.reference_init_M(C, V, 42)
    a = _run(data, M0, 6, s_exact=False, tile=tile)
    b = _run(data, M0, 6, s_exact="auto", tile=tile)
    assert a["eff"] == "bf16x3" and b["eff"].startswith("bf16x3 (S exact")
    for k in ("P", "h", "G", "M", "m1"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_constrained_and_against_the_oracle(sim):
    from oracle import tangram_oracle as orc
    C, K, V = 150, 40, 60
    data = orc.make_synthetic(C, K, V, seed=7)
    M0, F0 = orc.reference_init_MF_constrained(C, V, 5)
    lam = dict(lambda_d=1.0, lambda_g1=1.0, lambda_g2=1.0, lambda_count=1.0, lambda_f_reg=1.0)
    a = _run(data, M0, 5, False, mode="constrained", F0=F0, lam=lam, target_count=40.0)
    b = _run(data, M0, 5, "auto", mode="constrained", F0=F0, lam=lam, target_count=40.0)
    assert b["eff"].startswith("bf16x3 (S exact")
    for k in ("P", "h", "G"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, dtype=np.float64, target_count=40.0, **lam)
    Po, Fo, ho = o.train(5, 0.1)
    assert np.abs(b["P"] - Po).max() <= 2e-4
    assert np.abs(b["h"][:, 1] - np.array(ho["main_loss"])).max() <= 1e-5


def test_the_library_refuses_the_shortcut_when_s_is_not_exact(sim):
    from oracle import tangram_oracle as orc
    C, K, V = 96, 24, 40
    data = orc.make_synthetic(C, K, V, seed=2, n_types=3)
    M0 = orc.reference_init_M(C, V, 1)
    # (1) log-normalised expression: not exact
    logd = dict(data, S=np.log1p(data["S"]).astype(np.float32))
    a, b = _run(logd, M0, 3, False, tile=128), _run(logd, M0, 3, "auto", tile=128)
    assert b["eff"] == "bf16x3" and np.array_equal(a["P"], b["P"])
    # (2) exact S, but the augmentation column d_source is not
    ds = (np.random.default_rng(0).random(C) + 0.1).astype(np.float32)
    ds /= ds.sum()
    b = _run(data, M0, 3, "auto", tile=128, d_source=ds)
    assert b["eff"] == "bf16x3"
    # (3) one value just past bf16 precision
    odd = dict(data, S=data["S"].copy())
    odd["S"][5, 7] = 257.0
    assert _run(odd, M0, 2, "auto", tile=128)["eff"] == "bf16x3"
    # (4) the one-hot cell-type encoding IS exact: the spatial ct-islands term rides the same GEMMs
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_ct_islands=0.17)
    kw = dict(neighborhood_filter=orc.grid_graph(V, standardized=False, self_inclusion=False), ct_encode=data["ct_encode"])
    a, b = _run(data, M0, 4, False, tile=128, lam=lam, **kw), _run(data, M0, 4, "auto", tile=128, lam=lam, **kw)
    assert b["eff"].startswith("bf16x3 (S exact")
    for k in ("P", "h"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_project_genes_of_an_exact_handle_takes_any_gene_block(sim):
    """tg_mapper_project_genes on a two-product handle: the projected block (other genes, not exact) runs on the general kernels."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 120, 30, 80
    data = orc.make_synthetic(C, K, V, seed=4)
    M0 = orc.reference_init_M(C, V, 3)
    S_other = np.random.default_rng(1).gamma(2.0, 1.3, size=(C, 45)).astype(np.float32)          # not bf16-exact
    outs = []
    for se in (False, "auto"):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=dict(lambda_d=1.0), s_exact=se)
        e.step(3, 0.1)
        outs.append(e.project_genes(S_other).numpy())
    assert np.array_equal(outs[0], outs[1])
    ref = outs[0]
    P = None
