"""CPU tests of the real kernel sources under the HIP emulator of tests/hipsim (the authoring
container has no GPU): index arithmetic, masking, barriers, launch sequence and host logic are
checked against fixtures generated from the unmodified reference.  The GPU parity tests proper
are tests/test_gpu_parity.py (-m gpu)."""
import ctypes as ct

import numpy as np
import pytest

from tests.hipsim.build_sim import build_sim
from tests import parity_common as pc


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


@pytest.mark.parametrize("name,precision,epochs", [
    ("cells_ragged", "bf16x3", 12),      # ragged C/K/V, lambda_g2, density
    ("cells_nodensity", "fp32", 8),      # no density term
    ("clusters_dsource", "bf16x3", 6),   # d_source, 3 spot tiles, tiny C
    ("cells_allreg", "fp32", 8),         # entropy + L1 + L2 regularisers
    ("cells_ragged", "bf16", 6),
    ("constrained", "fp32", 8),          # MapperConstrained: filter F, count / f_reg terms
    ("constrained_entropy", "bf16x3", 6),
    ("cells_val", "bf16x3", 7),          # val_each: validation metrics of _val_loss_fn
    ("cells_spatial", "fp32", 6),
    ("cells_autocorr", "fp32", 6),       # Getis-Ord + Moran + Geary on a CSR spot graph        # neighbourhood-weighted gene term + cell-type islands on a CSR spot graph
])
def test_emulated_kernels_match_reference(sim, name, precision, epochs):
    res = pc.run_case(name, "cpu", precision, epochs=epochs)
    pc.check_against_golden(res, precision, full_length=False)


def test_emulated_forward_splits_agree(sim):
    """Splitting the contraction over cells (forward GEMM) must not change the result beyond rounding."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(200, 20, 40, seed=3)
    M0 = orc.reference_init_M(200, 40, 7)
    outs = []
    for splits in (1, 3):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32",
                            lambdas=dict(lambda_d=1.0), fwd_splits=splits)
        outs.append(e.project().numpy())
    ref = orc.softmax_rows(M0.astype(np.float64)).T @ data["S"].astype(np.float64)
    for o in outs:
        assert np.linalg.norm(o - ref) / np.linalg.norm(ref) < 1e-6


@pytest.mark.parametrize("K", [24, 290])
@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16"])
def test_emulated_large_tile_geometry(sim, precision, K):
    """Force the 256 x 256 / 512-thread geometry on a small ragged problem and compare one step with the oracle.
    K = 290 pads to 512 gene columns: the bf16 / bf16x3 forward then runs on the 128 x 512 tiles (TgGeoWide)."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, V = 300, 270                   # 2 x 2 tiles of 256, ragged
    data = orc.make_synthetic(C, K, V, seed=9)
    M0 = orc.reference_init_M(C, V, 4)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.3, lambda_r=1e-3)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision=precision, lambdas=lam, tile_size=256)
    geo = (ct.c_int * 8)()
    assert e._lib.tg_debug_layout(ct.byref(e.cfg), geo) == 0
    assert geo[0] == 256 and geo[5] == int(K == 290 and precision == "bf16x3")
    n = 2
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)
    tol = pc.TOL[precision]
    from tangram_amd import _capi
    for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss"), (_capi.H_VG, "vg_reg"),
                   (_capi.H_KL, "kl_reg"), (_capi.H_ENTROPY, "entropy_reg")):
        scale = max(1.0, abs(ho[k][0]))
        np.testing.assert_allclose(hist[:, col].numpy(), np.array(ho[k]), atol=tol["loss"] * scale, rtol=0, err_msg=k)
    assert np.abs(e.result().numpy() - Po).max() < tol["P"]


@pytest.mark.parametrize("mode", ["mapper", "constrained"])
def test_emulated_backward_on_small_tiles_under_the_256_layout(sim, mode):
    """bwd_tile=128: the backward GEMM of a 256-layout problem on 128^2 tiles (what the fixed rule picks for thin shard grids) --
    alone (X-only epilogue) and as a 1-rank spot shard (row-dot epilogue, twice the partials per row); against the fp64 oracle,
    and the X-only path bit-identical to the 256^2 run."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from tests.local_comm import run_ranks
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    C, K, V = 610, 20, 270
    data = orc.make_synthetic(C, K, V, seed=21)
    n = 2
    if mode == "constrained":
        M0, F0 = orc.reference_init_MF_constrained(C, V, 5)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_count=0.8, lambda_f_reg=1.2)
        kw = dict(F0=F0, mode="constrained", target_count=100.0)
        o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, target_count=100.0, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
    else:
        M0 = orc.reference_init_M(C, V, 5)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_l1=1e-4)
        kw = {}
        o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)

    def check(hist, P):
        for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss"), (_capi.H_KL, "kl_reg"), (_capi.H_ENTROPY, "entropy_reg")):
            ref = np.array([float(x) for x in ho[k]])
            np.testing.assert_allclose(hist[:, col], ref, atol=1e-5 * max(1.0, np.abs(ref).max()), rtol=0, err_msg=k)
        assert np.abs(P - Po).max() < 2e-4

    def alone(bwd_tile):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=lam, tile_size=256,
                            bwd_tile=bwd_tile, **kw)
        hist = e.new_history(n)
        e.step(n, 0.1, hist)
        return hist.numpy().copy(), e.result().numpy().copy()

    h256, P256 = alone(256)
    h128, P128 = alone(128)
    check(h128, P128)
    np.testing.assert_array_equal(P128, P256)
    np.testing.assert_array_equal(h128, h256)

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=lam, tile_size=256,
                          bwd_tile=128, comm=comm, **kw)
        h = sh.eng.new_history(n)
        sh.run(n, 0.1, h)
        return h.numpy(), sh.result_full().numpy()

    (h1, P1), = run_ranks(1, rank_fn)
    check(h1, P1)


@pytest.mark.parametrize("bands,tile", [(3, 128), (2, 256)])
def test_emulated_cell_band_pipeline(sim, bands, tile):
    """The 3-stream cell-band schedule (backward | Adam | next forward) must give the sequential schedule's results."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V = 420, 16, 150
    data = orc.make_synthetic(C, K, V, seed=13)
    M0 = orc.reference_init_M(C, V, 6)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.4, lambda_r=1e-3)
    outs = []
    for pb in (1, bands):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=lam,
                            tile_size=tile, pipeline_bands=pb)
        hist = e.new_history(3)
        e.step(2, 0.1, hist, 0)          # two steps in one call: step 2 uses the pre-launched forward
        e.step(1, 0.1, hist, 2)
        outs.append((e.result().numpy(), hist.numpy()))
    np.testing.assert_allclose(outs[0][1][:, :5], outs[1][1][:, :5], atol=2e-6, rtol=1e-6)
    # (the forward partial sums are cut at band boundaries instead of equal step ranges: fp32 summation order differs)
    np.testing.assert_allclose(outs[0][0], outs[1][0], atol=2e-5)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(3, 0.1)
    np.testing.assert_allclose(outs[1][1][:, 0], np.array(ho["total_loss"]), atol=1e-5)
    assert np.abs(outs[1][0] - Po).max() < 2e-5


@pytest.mark.parametrize("shape", [(2, 3, 4), (5, 1, 7), (3, 2, 1), (1, 4, 6), (17, 130, 9)])
def test_emulated_degenerate_shapes(sim, shape):
    """Tiny / degenerate problem sizes (single cell, single gene, single spot, K spilling into a second gene tile)."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V = shape
    rng = np.random.default_rng(C * 100 + K * 10 + V)
    S = rng.integers(1, 5, size=(C, K)).astype(np.float32)
    G = rng.integers(1, 5, size=(V, K)).astype(np.float32)
    d = (G.sum(1) / G.sum()).astype(np.float32)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    e = HipMapperEngine(S, G, M0, d=d, device="cpu", precision="fp32", lambdas=lam)
    hist = e.new_history(3)
    e.step(3, 0.1, hist)
    o = orc.OracleMapper(S, G, d=d, M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(3, 0.1)
    from tangram_amd import _capi
    np.testing.assert_allclose(hist[:, _capi.H_TOTAL].numpy(), np.array(ho["total_loss"]), atol=2e-5)
    np.testing.assert_allclose(e.result().numpy(), Po, atol=2e-5)
    np.testing.assert_allclose(e.project().numpy(), Po.T @ S.astype(np.float64), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", ["mapper", "constrained"])
def test_emulated_checkpoint_resume(sim, mode):
    """tg_mapper_state / tg_mapper_set_step: logits + both Adam moments + step counter copied into a fresh handle continue the
    run: same trajectory up to the rounding of the softmax normaliser (set_step rebuilds the row statistics the kernels otherwise
    carry between iterations with a different summation order: 1 ulp)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    from oracle import tangram_oracle as orc
    _cols_cf = [_capi.H_COUNT, _capi.H_FREG]
    C, K, V = 90, 20, 70
    data = orc.make_synthetic(C, K, V, seed=4)
    M0 = orc.reference_init_M(C, V, 2)
    if mode == "constrained":
        M0, F0 = orc.reference_init_MF_constrained(C, V, 2)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_count=0.7, lambda_f_reg=1.3)
        kw = dict(mode="constrained", target_count=40.0)
    else:
        F0, kw = None, {}
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_l2=1e-6)
    mk = lambda M, F=F0: HipMapperEngine(data["S"], data["G"], M, d=data["d"], F0=F, device="cpu", precision="bf16x3", lambdas=lam, **kw)
    a = mk(M0)
    ha = a.new_history(7)
    a.step(7, 0.1, ha)
    b = mk(M0)
    hb = b.new_history(7)
    b.step(3, 0.1, hb, 0)
    Mb, m1b, m2b, step = b.logits()
    assert step == 3
    c = mk(np.zeros_like(M0), None if F0 is None else np.ones_like(F0))      # a fresh handle with unrelated logits
    Mc, m1c, m2c, _ = c.logits()
    Mc.copy_(Mb); m1c.copy_(m1b); m2c.copy_(m2b)
    if mode == "constrained":
        c.filter_state().copy_(b.filter_state())
    c.set_step(step)
    hc = c.new_history(7)
    c.step(4, 0.1, hc, 3)
    np.testing.assert_allclose(hc.numpy()[3:, :5], ha.numpy()[3:, :5], rtol=5e-6, atol=1e-7)
    np.testing.assert_allclose(c.result().numpy(), a.result().numpy(), rtol=0, atol=1e-6)
    if mode == "constrained":
        np.testing.assert_allclose(c.result(with_filter=True)[1].numpy(), a.result(with_filter=True)[1].numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(hc.numpy()[3:, _cols_cf], ha.numpy()[3:, _cols_cf], rtol=5e-6, atol=1e-7)
    else:
        with pytest.raises(Exception):
            c.filter_state()
    assert c.logits()[3] == 7


@pytest.mark.parametrize("case", range(16))
def test_emulated_random_configurations(sim, case):
    """Seeded random draws of shape, mode, regularisers, priors and GEMM precision (ragged sizes around the 128-tile, 64-pitch and
    32 / 64-step boundaries), 3 epochs each against the fp64 oracle: history columns, mapping, filter."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    rng = np.random.default_rng(1000 + case)
    C = int(rng.choice([1, 7, 31, 33, 63, 65, 127, 129, 140]))
    K = int(rng.choice([1, 2, 15, 31, 64, 127, 130]))
    V = int(rng.choice([1, 3, 63, 64, 65, 128, 131, 200]))
    precision = ["fp32", "bf16x3", "bf16x3", "bf16"][int(rng.integers(4))]
    constrained = bool(rng.integers(2))
    data = orc.make_synthetic(C, K, V, seed=50 + case)
    n = 3
    pick = lambda vals: float(rng.choice(vals))
    lam = dict(lambda_g1=pick([1.0, 0.7]), lambda_d=pick([0.0, 1.0, 0.5]), lambda_g2=pick([0.0, 0.5]), lambda_r=pick([0.0, 1e-3]))
    d = data["d"] if lam["lambda_d"] > 0 or constrained else None
    if d is None:
        lam["lambda_d"] = 0.0
    if constrained:
        lam["lambda_d"] = lam["lambda_d"] or 1.0
        lam.update(lambda_count=pick([1.0, 0.3]), lambda_f_reg=pick([1.0, 2.0]))
        tc = max(1.0, 0.5 * C)
        M0, F0 = orc.reference_init_MF_constrained(C, V, case)
        o = orc.OracleMapperConstrained(data["S"], data["G"], d, M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=d, F0=F0, mode="constrained", device="cpu", precision=precision, lambdas=lam,
                            target_count=tc)
    else:
        lam.update(lambda_l1=pick([0.0, 1e-4]), lambda_l2=pick([0.0, 1e-5]))
        ds = None
        if d is not None and rng.integers(2):
            ds = rng.dirichlet(np.ones(C)).astype(np.float32)
        M0 = orc.reference_init_M(C, V, case)
        o = orc.OracleMapper(data["S"], data["G"], d=d, d_source=ds, M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=d, d_source=ds, device="cpu", precision=precision, lambdas=lam)
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    h = hist.numpy().astype(np.float64)
    tol = pc.TOL[precision]
    cols = [(_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss")]
    if lam["lambda_g2"] > 0:
        cols.append((_capi.H_VG, "vg_reg"))
    if lam["lambda_d"] > 0:
        cols.append((_capi.H_KL, "kl_reg"))
    if lam["lambda_r"] > 0:
        cols.append((_capi.H_ENTROPY, "entropy_reg"))
    for col, k in cols:
        ref = np.array([float(x) for x in ho[k]])
        err = np.abs(h[:, col] - ref).max()
        assert err <= 3 * tol["loss"] * max(1.0, np.abs(ref).max()), (case, C, K, V, precision, constrained, k, err)
    if constrained:
        P, F = e.result(with_filter=True)
        assert np.abs(F.numpy() - Fo).max() <= (2e-5 if precision != "bf16" else 5e-3)
    else:
        P = e.result()
    assert np.abs(P.numpy() - Po).max() <= tol["P"], (case, C, K, V, precision)


# (C, K, V, constrained, lambda_g2): cluster counts around the compile-time bounds (multiples of 4, max 32), gene counts of one chunk,
# a partial second chunk (K = 300 pads to 384) and three chunks, spot counts around the 64-spot block
SMALL_C_CASES = [
    (5, 40, 70, False, 0.0), (18, 250, 330, False, 0.0), (18, 250, 130, True, 0.5), (32, 300, 129, False, 0.7),
    (20, 600, 64, False, 0.0), (3, 1, 1, False, 0.5), (12, 127, 200, True, 0.0), (29, 20, 63, False, 1.0),
]


@pytest.mark.parametrize("i", range(len(SMALL_C_CASES)))
def test_emulated_small_cluster_path(sim, i):
    """The clusters-mode kernels (tg_sc_softmax / tg_sc_forward / tg_sc_backward) against the fp64 oracle."""
    C, K, V, constrained, g2 = SMALL_C_CASES[i]
    pc.small_cluster_case("cpu", C, K, V, constrained, g2, seed=300 + i)


def test_emulated_tile_size_pins_the_gemm_path(sim):
    pc.small_cluster_case("cpu", 18, 250, 130, False, 0.5, seed=311, precision="fp32", tile_size=128)


@pytest.mark.parametrize("precision,rtol", [("fp32", 2e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)])
def test_emulated_project_genes_all_genes(sim, precision, rtol):
    """tg_mapper_project_genes: softmax(M)^T S_all over a gene set wider than the training genes (several blocks of
    cfg.n_genes, ragged last block, padded row pitch) against P^T S_all in float64 (reference utils.py:366-368)."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    import torch
    C, K, V, K_all = 70, 9, 33, 31
    data = orc.make_synthetic(C, K, V, seed=21)
    rng = np.random.default_rng(5)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision=precision,
                        lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    e.step(2, 0.1, e.new_history(2))
    P = e.result().numpy().astype(np.float64)
    wide = torch.as_tensor(rng.gamma(1.0, 2.0, size=(C, K_all + 5)).astype(np.float32))
    S_all = wide[:, 2:2 + K_all]                                  # a view: row pitch 36 != 31
    out = e.project_genes(S_all).numpy()
    want = P.T @ S_all.numpy().astype(np.float64)
    assert out.shape == (V, K_all)
    assert np.abs(out - want).max() <= rtol * np.abs(want).max()
    # the training state is untouched: the next step equals an uninterrupted run
    e2 = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision=precision,
                         lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    h2 = e2.new_history(3); e2.step(3, 0.1, h2)
    h1 = e.new_history(1); e.step(1, 0.1, h1)
    np.testing.assert_array_equal(h1[0].numpy(), h2[2].numpy())
    with pytest.raises(ValueError):
        e.project_genes(np.zeros((C + 1, 4), np.float32))


def test_emulated_project_genes_constrained_filter(sim):
    """unfiltered=True: softmax(M)^T S (what adata_map.X.T @ S gives, mapping_optimizer.py:637); False: with the filter."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V = 40, 7, 19
    data = orc.make_synthetic(C, K, V, seed=22)
    rng = np.random.default_rng(6)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    F0 = rng.normal(size=(C,)).astype(np.float32)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device="cpu", precision="fp32",
                        lambdas=dict(lambda_g1=1.0, lambda_d=1.0, lambda_count=1.0, lambda_f_reg=1.0), target_count=10)
    e.step(2, 0.1, e.new_history(2))
    P, F = e.result(with_filter=True)
    P = P.numpy().astype(np.float64); F = F.numpy().astype(np.float64)
    S_all = rng.gamma(1.0, 2.0, size=(C, 16)).astype(np.float32)
    np.testing.assert_allclose(e.project_genes(S_all, unfiltered=True).numpy(), P.T @ S_all, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(e.project_genes(S_all, unfiltered=False).numpy(), P.T @ (S_all * F[:, None]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K", [22, 23])
def test_emulated_spatial_terms_with_ragged_gene_counts(sim, K):
    """All CSR spatial terms at gene counts that are not multiples of 4 (tg_spmm handles 4 genes per thread: the last, partial
    quad is guarded per element) against the fp64 oracle."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    C, V = 50, 64
    data = orc.make_synthetic(C, K, V, seed=K, n_types=3)
    M0 = orc.reference_init_M(C, V, 9)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17,
               lambda_getis_ord=0.3, lambda_moran=0.4, lambda_geary=0.2)
    kw = dict(voxel_weights=orc.grid_graph(V, standardized=True, self_inclusion=True),
              neighborhood_filter=orc.grid_graph(V, standardized=False, self_inclusion=False), ct_encode=data["ct_encode"],
              spatial_weights=orc.grid_graph(V, standardized=True, self_inclusion=False))
    n = 3
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam, **kw)
    Po, ho = o.train(n, 0.1)
    for prec in ("fp32", "bf16x3"):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision=prec, lambdas=lam, **kw)
        hist = e.new_history(n)
        e.step(n, 0.1, hist)
        np.testing.assert_allclose(hist[:, _capi.H_TOTAL].numpy(), np.array(ho["total_loss"], dtype=np.float64), atol=2e-5, rtol=2e-5)
        np.testing.assert_allclose(e.result().numpy(), Po, atol=5e-5)


def test_emulated_three_shards_in_threads(sim):
    """tangram_amd.sharded with three spot shards as threads of this process (tests/local_comm.py; the gloo test in
    test_sharded_gloo.py covers real process groups): same history and mapping as the unsharded engine."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from oracle import tangram_oracle as orc
    from tests.local_comm import run_ranks
    from tangram_amd import _capi
    C, K, V = 40, 12, 91
    data = orc.make_synthetic(C, K, V, seed=2)
    M0 = orc.reference_init_M(C, V, 3)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3)

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=lam, comm=comm)
        h = sh.eng.new_history(3)
        sh.run(3, 0.1, h)
        return sh.finalize_history(h).numpy(), sh.result_full().numpy(), sh.validate()

    res = run_ranks(3, rank_fn)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=lam)
    h1 = e.new_history(3)
    e.step(3, 0.1, h1)
    val1 = e.validate()
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY]
    for hist, P, val in res:
        np.testing.assert_allclose(hist[:, cols], h1.numpy()[:, cols], atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(P, e.result().numpy(), atol=1e-6)
        # _val_loss_fn over all spots (per-gene sums, spot cosines, entropy and non-zero fractions all-reduced in the library)
        np.testing.assert_allclose(val, val1, rtol=2e-6, atol=1e-7)
        assert val == res[0][2]                       # every rank holds the same four numbers


def test_emulated_project_genes_from_csr(sim):
    """project_genes with a scipy CSR single-cell matrix: the gene blocks are expanded on the device (tg_csr_columns_to_dense)
    and must give exactly the dense path's result (reference: adata_sc.X.toarray() on the host, utils.py:364-365)."""
    import scipy.sparse as sp
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V, K_all = 50, 8, 21, 29
    data = orc.make_synthetic(C, K, V, seed=23)
    rng = np.random.default_rng(7)
    e = HipMapperEngine(data["S"], data["G"], rng.normal(size=(C, V)).astype(np.float32), d=data["d"], device="cpu",
                        precision="fp32", lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    e.step(2, 0.1, e.new_history(2))
    dense = (rng.gamma(1.0, 2.0, size=(C, K_all)) * (rng.random((C, K_all)) < 0.3)).astype(np.float32)
    dense[3] = 0.0                                               # an empty row
    for fmt in (sp.csr_matrix, sp.csc_matrix, sp.coo_matrix):
        out = e.project_genes(fmt(dense)).numpy()
        np.testing.assert_array_equal(out, e.project_genes(dense).numpy())
    with pytest.raises(ValueError):
        e.project_genes(sp.csr_matrix(dense[:-1]))


@pytest.mark.parametrize("units", [1, 2, 5, 7, 13, 29])
def test_emulated_forward_work_units_across_tile_boundaries(sim, units):
    """The forward GEMM's (spot tile, contraction step) space cut into `units` equal pieces per gene tile whatever the tile boundaries
    (fwd_splits < 0): 3 spot tiles of 128 x 11 steps = 33 global steps (x 2 gene tiles), so 5 / 7 / 13 / 29 pieces start and end in
    the middle of tiles, a piece spans up to three of them, a tile is summed from up to 11 partial slots.  Every decomposition: the oracle's trajectory within the fp32
    tolerance, and the same projection as the one-unit-per-tile run up to summation order."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 340, 200, 300                          # 3 spot tiles x 2 gene tiles (201 gene columns), 11 steps of 32 cells
    data = orc.make_synthetic(C, K, V, seed=17)
    M0 = orc.reference_init_M(C, V, 4)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=lam, tile_size=128,
                        fwd_splits=-units)
    h = e.new_history(3)
    e.step(3, 0.1, h)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(3, 0.1)
    assert np.abs(h.numpy()[:, 1] - np.array(ho["main_loss"])).max() <= 1e-5
    assert np.abs(h.numpy()[:, 0] - np.array(ho["total_loss"])).max() <= 1e-5
    assert np.abs(e.result().numpy() - Po).max() <= 2e-4
    Gh = e.project().numpy().astype(np.float64)
    ref = Po.T @ data["S"].astype(np.float64)
    assert np.linalg.norm(Gh - ref) / np.linalg.norm(ref) <= 1e-4


@pytest.mark.parametrize("s_exact", [False, "auto"])
def test_emulated_wide_forward_geometry(sim, s_exact):
    """The split-bf16 forward on 128 x 512 tiles (tile_size = 256 and a padded gene count that is a multiple of 512: waves 0-3 stage
    the softmax operand, waves 4-7 only multiply) on the emulator -- the small CPU cases otherwise never take this geometry.
    Against the fp64 oracle, general and two-product path, ragged spots / genes / cells."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    import ctypes as ct
    C, K, V = 210, 300, 150
    data = orc.make_synthetic(C, K, V, seed=31)
    M0 = orc.reference_init_M(C, V, 8)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="bf16x3", lambdas=lam, tile_size=256, s_exact=s_exact)
    geo = (ct.c_int * 8)()
    assert e._lib.tg_debug_layout(ct.byref(e.cfg), geo) == 0 and geo[5] == 1, "this shape must take the wide forward geometry"
    h = e.new_history(3)
    e.step(3, 0.1, h)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(3, 0.1)
    assert np.abs(h.numpy()[:, 1] - np.array(ho["main_loss"])).max() <= 1e-5 and np.abs(h.numpy()[:, 0] - np.array(ho["total_loss"])).max() <= 1e-5
    assert np.abs(e.result().numpy() - Po).max() <= 2e-4
    Gh = e.project().numpy().astype(np.float64)
    ref = Po.T @ data["S"].astype(np.float64)
    assert np.linalg.norm(Gh - ref) / np.linalg.norm(ref) <= 1e-4
