"""Host pre-processing on the device (tangram_amd/preprocess.py -> tg_csr_gather_columns, tg_row_sums,
tg_cluster_aggregate) against the reference's NumPy formulas (tangram/mapping_utils.py:259-275, :88-89, :126-132):
the gather bit-exactly, the sums within 1 ulp of the exactly rounded result.  CPU: through the emulated C ABI; the same checks
run on the GPU in tests/test_gpu_parity.py::test_device_preprocessing."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from tests.hipsim.build_sim import build_sim
from tangram_amd.anndata_lite import AnnDataLite


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


def ulps(a, b):
    """distance in float32 units in the last place"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))


def make_counts(n, g, seed, normalised):
    rng = np.random.default_rng(seed)
    X = (rng.negative_binomial(2, 0.4, size=(n, g)) * (rng.random((n, g)) < 0.25)).astype(np.float32)
    if normalised:                                     # like sc.pp.normalize_total: non-integer values
        X = (X / np.maximum(X.sum(1, keepdims=True), 1) * 1e4).astype(np.float32)
    return X


def check_preprocessing(device):
    from tangram_amd import preprocess as pre
    for normalised in (False, True):
        X = make_counts(157, 300, 3, normalised)
        Xs = sp.csr_matrix(X)
        before = Xs.copy()
        cols = np.random.default_rng(1).permutation(300)[:77]
        # gather: bit-identical to adata[:, genes].X.toarray() (mapping_utils.py:259-262)
        S = pre.gather_training_genes(Xs, cols, device).cpu().numpy()
        assert S.dtype == np.float32 and np.array_equal(S, X[:, cols])
        assert (Xs != before).nnz == 0                                        # the caller's matrix is untouched
        # a CSR matrix with duplicate entries / unsorted indices is canonicalised on a copy
        Xd = sp.csr_matrix((np.r_[Xs.data, 1.0], np.r_[Xs.indices, Xs.indices[-1]], np.r_[Xs.indptr[:-1], Xs.indptr[-1] + 1]), shape=Xs.shape)
        want = X.copy(); want[-1, Xs.indices[-1]] += 1.0
        assert np.array_equal(pre.gather_training_genes(Xd, np.arange(300), device).cpu().numpy(), want)
        # density: rna_count_per_spot / np.sum(rna_count_per_spot)  (mapping_utils.py:88-89)
        exact = X.astype(np.float64).sum(1) / X.astype(np.float64).sum()
        ref = np.array(Xs.sum(axis=1)).squeeze(); ref = ref / np.sum(ref)      # the reference's own float32 arithmetic
        for M in (Xs, X):
            d = pre.rna_count_density(M, device).cpu().numpy()
            assert d.dtype == np.float32
            assert ulps(d, exact.astype(np.float32)).max() <= 1.0
            assert ulps(d, ref).max() <= 4.0                                   # (the reference's float32 total carries a few ulp itself)
        rs = pre.row_sums(Xs, device).cpu().numpy()
        assert ulps(rs, X.astype(np.float64).sum(1).astype(np.float32)).max() <= 0.5 + 1e-9
        # clusters: per-cluster sum / mean (mapping_utils.py:126-132)
        labels = np.random.default_rng(2).choice(["a", "b", "c", "d"], size=157)
        uniq = ["c", "a", "d", "b"]
        Sd = pre.gather_training_genes(Xs, np.arange(300), device)
        for scale in (True, False):
            got = pre.cluster_expression(Sd, labels, uniq, scale=scale).cpu().numpy()
            wantc = np.stack([(X[labels == l].astype(np.float64).sum(0) if scale else X[labels == l].astype(np.float64).mean(0)) for l in uniq])
            assert ulps(got, wantc.astype(np.float32)).max() <= 1.0
            refc = np.stack([(X[labels == l].sum(0) if scale else X[labels == l].mean(0)) for l in uniq])
            np.testing.assert_allclose(got, refc, rtol=2e-6)
    with pytest.raises(ValueError):
        pre.gather_training_genes(Xs, [0, 0], device)
    with pytest.raises(ValueError):
        pre.gather_training_genes(Xs, [300], device)


def test_device_preprocessing_emulated(sim):
    check_preprocessing("cpu")


def test_map_cells_to_space_from_sparse_anndata(sim):
    """`map_cells_to_space` with scipy-sparse `adata.X` (the usual AnnData storage): training genes gathered on the device,
    clusters aggregated on the device -- same result as the dense-input run."""
    import tangram_amd as tg
    from tests.test_map_cells_to_space import _adatas
    for mode, kw in (("cells", {}), ("clusters", dict(cluster_label="subclass_label")), ("constrained", dict(target_count=9))):
        ad_sc, ad_sp = _adatas(C=50, K=12, V=30)
        dense = tg.map_cells_to_space(ad_sc, ad_sp, mode=mode, device="cpu", num_epochs=4, random_state=42, verbose=False,
                                      gemm_precision="fp32", **kw)
        ad_sc2, ad_sp2 = _adatas(C=50, K=12, V=30)
        ad_sc2 = AnnDataLite(sp.csr_matrix(ad_sc2.X), obs=ad_sc2.obs, var=ad_sc2.var, uns=ad_sc2.uns)
        ad_sp2 = AnnDataLite(sp.csr_matrix(ad_sp2.X), obs=ad_sp2.obs, var=ad_sp2.var, uns=ad_sp2.uns)
        sparse = tg.map_cells_to_space(ad_sc2, ad_sp2, mode=mode, device="cpu", num_epochs=4, random_state=42, verbose=False,
                                       gemm_precision="fp32", **kw)
        np.testing.assert_allclose(sparse.X, dense.X, atol=1e-6, err_msg=mode)
        np.testing.assert_allclose(sparse.uns["train_genes_df"].sort_index().to_numpy(), dense.uns["train_genes_df"].sort_index().to_numpy(),
                                   atol=1e-5, err_msg=mode)
    # an all-zero training gene is still detected (mapping_utils.py:277)
    ad_sc, ad_sp = _adatas(C=50, K=12, V=30)
    Xz = ad_sc.X.copy(); Xz[:, 2] = 0
    ad_z = AnnDataLite(sp.csr_matrix(Xz), obs=ad_sc.obs, var=ad_sc.var, uns=ad_sc.uns)
    with pytest.raises(ValueError, match="all zero"):
        tg.map_cells_to_space(ad_z, ad_sp, device="cpu", num_epochs=1, verbose=False)


def test_density_priors_and_cluster_expression_helpers(sim):
    import tangram_amd as tg
    from tests.test_map_cells_to_space import _adatas
    ad_sc, ad_sp = _adatas(C=40, K=10, V=21)
    want = ad_sp.obs["rna_count_based_density"].to_numpy().copy()
    ad_sp.obs.drop(columns=["rna_count_based_density", "uniform_density"], inplace=True)
    tg.density_priors(ad_sp, device="cpu")
    np.testing.assert_allclose(ad_sp.obs["rna_count_based_density"].to_numpy(), want, rtol=3e-7)
    np.testing.assert_allclose(ad_sp.obs["uniform_density"].to_numpy(), 1.0 / 21)
    for scale in (True, False):
        host = tg.adata_to_cluster_expression(ad_sc, "subclass_label", scale=scale)
        dev = tg.adata_to_cluster_expression(ad_sc, "subclass_label", scale=scale, device="cpu")
        np.testing.assert_allclose(dev.X, host.X, rtol=2e-6)
        assert list(dev.obs["subclass_label"]) == list(host.obs["subclass_label"])
        np.testing.assert_allclose(dev.obs["cluster_density"].to_numpy(), host.obs["cluster_density"].to_numpy())
