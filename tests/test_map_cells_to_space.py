"""`map_cells_to_space` drop-in surface (reference tangram/mapping_utils.py:141-428) with duck-typed AnnData,
driven through the emulated C ABI on CPU.  Mirrors the reference's own tests (tests/tangram_test.py:67-152):
argument errors, clusters / cells / constrained modes, result contract, train-score consistency."""
import numpy as np
import pandas as pd
import pytest

from tests.hipsim.build_sim import build_sim
from tangram_amd.anndata_lite import AnnDataLite
from oracle import tangram_oracle as orc


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


def _adatas(C=60, K=12, V=25, seed=2, extra_genes=3):
    data = orc.make_synthetic(C, K + extra_genes, V, seed=seed)
    genes = [f"g{i}" for i in range(K + extra_genes)]
    rng = np.random.default_rng(seed)
    obs_sc = pd.DataFrame({"subclass_label": rng.choice(["a", "b", "c"], size=C)}, index=[f"c{i}" for i in range(C)])
    G = data["G"]
    obs_sp = pd.DataFrame({"rna_count_based_density": G.sum(1) / G.sum(), "uniform_density": np.ones(V) / V},
                          index=[f"s{i}" for i in range(V)])
    ad_sc = AnnDataLite(data["S"], obs=obs_sc, var=pd.DataFrame(index=genes))
    ad_sp = AnnDataLite(G, obs=obs_sp, var=pd.DataFrame(index=genes))
    train = genes[:K]
    for ad in (ad_sc, ad_sp):
        ad.uns["training_genes"] = train
        ad.uns["overlap_genes"] = genes
    return ad_sc, ad_sp


def test_invalid_arguments_raise_value_error(sim):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas()
    with pytest.raises(ValueError, match="lambda_g1 cannot be 0"):
        tg.map_cells_to_space(ad_sc, ad_sp, lambda_g1=0, device="cpu")
    with pytest.raises(ValueError, match='Argument "mode"'):
        tg.map_cells_to_space(ad_sc, ad_sp, mode="test", device="cpu")
    with pytest.raises(ValueError, match="cluster_label must be specified"):
        tg.map_cells_to_space(ad_sc, ad_sp, mode="clusters", device="cpu")
    with pytest.raises(ValueError, match="density_prior"):
        tg.map_cells_to_space(ad_sc, ad_sp, density_prior="bogus", device="cpu")
    with pytest.raises(ValueError, match="target_count"):
        tg.map_cells_to_space(ad_sc, ad_sp, mode="constrained", device="cpu")
    ad_bad, _ = _adatas()
    del ad_bad.uns["training_genes"]
    with pytest.raises(ValueError, match="pp_adatas"):
        tg.map_cells_to_space(ad_bad, ad_sp, device="cpu")


def test_cells_mode_contract_and_train_score(sim):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas()
    n = 6
    ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode="cells", device="cpu", num_epochs=n, random_state=42,
                                   verbose=False, gemm_precision="fp32")
    assert ad_map.X.shape == (60, 25) and ad_map.X.dtype == np.float32
    np.testing.assert_allclose(ad_map.X.sum(axis=1), 1.0, atol=1e-5)
    assert list(ad_map.obs.index) == list(ad_sc.obs.index) and list(ad_map.var.index) == list(ad_sp.obs.index)
    df = ad_map.uns["train_genes_df"]
    assert list(df.columns) == ["train_score", "sparsity_sc", "sparsity_sp", "sparsity_diff"]
    assert (np.diff(df["train_score"].to_numpy()) <= 1e-9).all()                    # sorted descending
    hist = ad_map.uns["training_history"]
    assert set(hist) >= {"total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"} and len(hist["main_loss"]) == n
    assert np.isnan(hist["vg_reg"]).all() and not np.isnan(hist["kl_reg"]).any()    # lambda_d forced to 1 (:214-215)
    # same trajectory as the oracle with the reference's RNG for M (random_state=42)
    train = ad_sc.uns["training_genes"]
    S = ad_sc[:, train].X; G = ad_sp[:, train].X
    o = orc.OracleMapper(S, G, d=ad_sp.obs["rna_count_based_density"].to_numpy(), lambda_d=1, random_state=42)
    Po, ho = o.train(n)
    np.testing.assert_allclose(hist["main_loss"], ho["main_loss"], atol=1e-5)
    np.testing.assert_allclose(ad_map.X, Po, atol=1e-5)
    # train-score consistency (reference tests/tangram_test.py:159-210)
    o2 = orc.OracleMapper(S, G, d=ad_sp.obs["rna_count_based_density"].to_numpy(), lambda_d=1, M0=o.M)
    terms, _ = o2.loss_and_grad()
    assert abs(df["train_score"].mean() - terms["main_loss"]) < 1e-4


def test_clusters_and_constrained_modes(sim):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas()
    ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode="clusters", cluster_label="subclass_label", device="cpu",
                                   num_epochs=4, random_state=42, verbose=False, gemm_precision="fp32")
    assert ad_map.X.shape == (3, 25) and "cluster_density" in ad_map.obs.columns
    # oracle on the aggregated matrix with d_source = cluster densities
    vc = ad_sc.obs["subclass_label"].value_counts(normalize=True)
    train = ad_sc.uns["training_genes"]
    S = np.stack([ad_sc[:, train].X[(ad_sc.obs["subclass_label"] == l).to_numpy()].sum(0) for l in vc.index])
    o = orc.OracleMapper(S, ad_sp[:, train].X, d=ad_sp.obs["rna_count_based_density"].to_numpy(),
                         d_source=vc.to_numpy(), lambda_d=1, random_state=42)
    Po, _ = o.train(4)
    np.testing.assert_allclose(ad_map.X, Po, atol=1e-5)

    ad_map2 = tg.map_cells_to_space(ad_sc, ad_sp, mode="constrained", target_count=10, device="cpu", num_epochs=4,
                                    random_state=42, verbose=False, gemm_precision="fp32", lambda_g2=1)
    assert "F_out" in ad_map2.obs.columns and ad_map2.X.shape == (60, 25)
    assert all(isinstance(x, str) for x in ad_map2.uns["training_history"]["main_loss"])
    oc = orc.OracleMapperConstrained(ad_sc[:, train].X, ad_sp[:, train].X, ad_sp.obs["rna_count_based_density"].to_numpy(),
                                     lambda_d=1, lambda_g2=1, target_count=10, random_state=42)
    Pc, Fc, _ = oc.train(4)
    np.testing.assert_allclose(ad_map2.X, Pc, atol=1e-5)
    np.testing.assert_allclose(ad_map2.obs["F_out"].to_numpy(), Fc, atol=1e-5)


def test_spatial_extension_terms_with_csr_graph(sim):
    """lambda_neighborhood_g1 + lambda_ct_islands through map_cells_to_space with obsp spot graphs (CSR), against
    the oracle fed with the dense matrices the reference would build (spatial_weights.py:5-29)."""
    import scipy.sparse as sp
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(C=50, K=10, V=25)
    W_bin = orc.grid_graph(25, standardized=False, self_inclusion=False)
    rng = np.random.default_rng(0)
    dist = W_bin * rng.uniform(0.5, 2.0, size=W_bin.shape).astype(np.float32)
    ad_sp.obsp["spatial_connectivities"] = sp.csr_matrix(W_bin)
    ad_sp.obsp["spatial_distances"] = sp.csr_matrix(dist)
    n = 5
    ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode="cells", cluster_label="subclass_label", device="cpu", num_epochs=n,
                                   random_state=42, verbose=False, gemm_precision="fp32",
                                   lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17)
    train = ad_sc.uns["training_genes"]
    Wd = dist / dist.sum(1, keepdims=True) + np.eye(25, dtype=np.float32)
    lab = ad_sc.obs["subclass_label"]
    E = np.stack([(lab == c).to_numpy().astype(np.float32) for c in lab.unique()], axis=1)
    o = orc.OracleMapper(ad_sc[:, train].X, ad_sp[:, train].X, d=ad_sp.obs["rna_count_based_density"].to_numpy(), lambda_d=1,
                         lambda_neighborhood_g1=0.96, voxel_weights=Wd, lambda_ct_islands=0.17,
                         neighborhood_filter=W_bin, ct_encode=E, random_state=42)
    Po, ho = o.train(n)
    hist = ad_map.uns["training_history"]
    np.testing.assert_allclose([float(x) for x in hist["total_loss"]], ho["total_loss"], atol=1e-5)
    np.testing.assert_allclose(ad_map.X, Po, atol=1e-5)


def test_project_genes_on_device(sim):
    """tangram_amd.project_genes (reference utils.py:338-375): all genes of adata_sc projected with the mapping resident
    on the device; result contract + values against adata_map.X.T @ adata_sc.X."""
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(extra_genes=7)
    ad_sc.var.index = [g.upper() for g in ad_sc.var.index]                 # the reference lower-cases (:351)
    ad_sc.X = ad_sc.X.copy(); ad_sc.X[:, -1] = 0                           # an all-zero gene is dropped (:357)
    ad_sc_train = AnnDataLite(ad_sc.X, obs=ad_sc.obs, var=pd.DataFrame(index=[g.lower() for g in ad_sc.var.index]), uns=ad_sc.uns)
    ad_map = tg.map_cells_to_space(ad_sc_train, ad_sp, mode="cells", device="cpu", num_epochs=3, random_state=42,
                                   verbose=False, gemm_precision="fp32", keep_mapper=True)
    # explicit `mapper=`: the projection uses the mapping that is still resident on the device
    ad_ge = tg.project_genes(ad_map, ad_sc, device="cpu", gemm_precision="fp32", mapper=ad_map._tangram_amd_mapper)
    keep = [g.lower() for g in ad_sc.var.index]
    assert ad_ge.X.shape == (25, 18) and list(ad_ge.var.index) == keep[:18]
    assert list(ad_ge.obs.index) == list(ad_sp.obs.index)
    assert ad_ge.var["is_training"].sum() == 12 and "n_cells" in ad_ge.var.columns
    want = ad_map.X.astype(np.float64).T @ ad_sc_train.X[:, :18].astype(np.float64)
    np.testing.assert_allclose(ad_ge.X, want, rtol=1e-5, atol=1e-6)
    # default (reference semantics, utils.py:366): the mapping matrix the caller passes, `adata_map.X`, is what gets projected,
    # even with a live mapper on the object -- an edited adata_map.X must not be silently replaced by the trained mapping
    ad_ge2 = tg.project_genes(ad_map, ad_sc, device="cpu", gemm_precision="fp32")
    np.testing.assert_allclose(ad_ge2.X, want, rtol=2e-5, atol=2e-6)
    X_saved = ad_map.X.copy()
    ad_map.X = ad_map.X[:, ::-1] * np.linspace(0.2, 1.5, ad_map.X.shape[0], dtype=np.float32)[:, None]   # edited: not even row-stochastic
    ad_ge3 = tg.project_genes(ad_map, ad_sc, device="cpu", gemm_precision="fp32")
    np.testing.assert_allclose(ad_ge3.X, ad_map.X.astype(np.float64).T @ ad_sc_train.X[:, :18].astype(np.float64), rtol=2e-5, atol=2e-6)
    ad_map.X = X_saved
    ad_map._tangram_amd_mapper.release()
    # by default the result owns no device memory (no mapper attached)
    ad_map_plain = tg.map_cells_to_space(ad_sc_train, ad_sp, mode="cells", device="cpu", num_epochs=1, random_state=42,
                                         verbose=False, gemm_precision="fp32")
    assert not hasattr(ad_map_plain, "_tangram_amd_mapper")
    # clusters: the single-cell matrix is aggregated the same way as for the mapping (:359-360)
    ad_map_c = tg.map_cells_to_space(ad_sc_train, ad_sp, mode="clusters", cluster_label="subclass_label", device="cpu",
                                     num_epochs=3, random_state=42, verbose=False, gemm_precision="fp32")
    ad_ge_c = tg.project_genes(ad_map_c, ad_sc, cluster_label="subclass_label", device="cpu", gemm_precision="fp32")
    agg = tg.adata_to_cluster_expression(ad_sc_train, "subclass_label", scale=True)
    np.testing.assert_allclose(ad_ge_c.X, ad_map_c.X.astype(np.float64).T @ agg.X[:, :18], rtol=1e-5, atol=1e-5)
    # mismatching cells raise like the reference (:362-363)
    bad = AnnDataLite(ad_sc.X[:-1], obs=ad_sc.obs.iloc[:-1], var=ad_sc.var.copy(), uns=ad_sc.uns)
    with pytest.raises(ValueError, match="same `obs` index"):
        tg.project_genes(ad_map, bad, device="cpu")
