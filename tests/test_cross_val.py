"""`cross_val` / `cv_data_gen` drop-ins (reference tangram/utils.py:462-668) on the emulated C ABI: the fold generator against
sklearn's splitters (what the reference calls), the batched cross-validation against the reference's own procedure spelled out
with this package's `map_cells_to_space` -- one fold after the other, host projection, cosine per held-out gene."""
import numpy as np
import pytest

from tests.hipsim.build_sim import build_sim
from tests.test_map_cells_to_space import _adatas


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


@pytest.mark.parametrize("K", [10, 13, 27])
def test_fold_generator_matches_sklearn(K):
    import tangram_amd as tg
    from sklearn.model_selection import KFold, LeaveOneOut
    ad_sc, ad_sp = _adatas(K=K)
    genes = np.array(ad_sp.uns["training_genes"])
    for mode, cv in (("loo", LeaveOneOut()), ("10fold", KFold(n_splits=10))):
        ours = list(tg.cv_data_gen(ad_sc, ad_sp, mode))
        ref = [(list(genes[a]), list(genes[b])) for a, b in cv.split(genes)]
        assert ours == ref
    bad, _ = _adatas(K=K)
    del bad.uns["training_genes"]
    with pytest.raises(ValueError, match="pp_adatas"):
        list(tg.cv_data_gen(bad, ad_sp))
    ad_sp.uns["training_genes"] = list(genes[::-1])
    with pytest.raises(ValueError, match="Unmatched training_genes"):
        list(tg.cv_data_gen(ad_sc, ad_sp))


def _sequential_reference_procedure(ad_sc, ad_sp, folds, mode, epochs, **kw):
    """utils.py:566-640 with this package's map_cells_to_space: one fold after the other, projection and scores on the host."""
    import tangram_amd as tg
    src = tg.adata_to_cluster_expression(ad_sc, kw["cluster_label"], True, device="cpu") if mode == "clusters" else ad_sc
    tests, trains, preds = [], [], []
    for train_genes, test_genes in folds:
        ad_map = tg.map_cells_to_space(ad_sc, ad_sp, cv_train_genes=train_genes, mode=mode, device="cpu", num_epochs=epochs,
                                       verbose=False, gemm_precision="fp32", **kw)
        pred = ad_map.X.T.astype(np.float64) @ np.asarray(src[:, test_genes].X, dtype=np.float64)
        g = np.asarray(ad_sp[:, test_genes].X, dtype=np.float64)
        score = (pred * g).sum(0) / (np.linalg.norm(pred, axis=0) * np.linalg.norm(g, axis=0))
        tests.append(score.mean())
        trains.append(float(list(ad_map.uns["training_history"]["main_loss"])[-1]))
        preds.append(pred)
    return np.array(tests), np.array(trains), preds


def test_leave_one_out_in_clusters_mode_matches_the_sequential_procedure(sim, capsys):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(C=60, K=7, V=40, seed=5)
    folds = list(tg.cv_data_gen(ad_sc, ad_sp, "loo"))
    kw = dict(cluster_label="subclass_label", random_state=3, density_prior="rna_count_based")
    t_ref, tr_ref, p_ref = _sequential_reference_procedure(ad_sc, ad_sp, folds, "clusters", 6, **kw)
    cv, ad_ge, df = tg.cross_val(ad_sc, ad_sp, mode="clusters", num_epochs=6, device="cpu", cv_mode="loo", return_gene_pred=True,
                                 verbose=True, gemm_precision="fp32", folds_per_launch=4, **kw)
    out = capsys.readouterr().out
    assert out.count("cv set:") == 7 and "cv avg test score" in out and "cv avg train score" in out
    assert set(cv) == {"avg_test_score", "avg_train_score"}
    np.testing.assert_allclose(cv["avg_test_score"], t_ref.mean(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(cv["avg_train_score"], tr_ref.mean(), rtol=0, atol=1e-7)     # same bits as training alone
    genes = list(ad_sp.uns["training_genes"])
    assert ad_ge.X.shape == (40, 7) and list(ad_ge.var.index) == genes and list(ad_ge.obs.index) == list(ad_sp.obs.index)
    np.testing.assert_allclose(ad_ge.var["test_score"].to_numpy(), t_ref, atol=2e-6)
    np.testing.assert_allclose(ad_ge.X, np.concatenate(p_ref, axis=1), rtol=2e-5, atol=1e-6)
    assert list(df.columns) == ["score", "is_training", "sparsity_sp", "sparsity_sc", "sparsity_diff"]
    assert list(df.index) == genes and not df["is_training"].any()
    np.testing.assert_allclose(df["sparsity_diff"], df["sparsity_sp"] - df["sparsity_sc"])
    assert (df["sparsity_sp"].to_numpy() == 1 - (np.asarray(ad_sp[:, genes].X) != 0).mean(0)).all()


@pytest.mark.parametrize("mode", ["cells", "constrained"])
def test_ten_fold_matches_the_sequential_procedure(sim, mode):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(C=40, K=12, V=30, seed=8)
    folds = list(tg.cv_data_gen(ad_sc, ad_sp, "10fold"))
    kw = dict(random_state=11, density_prior="uniform")
    if mode == "constrained":
        kw.update(target_count=12, lambda_count=1, lambda_f_reg=1)
    t_ref, tr_ref, _ = _sequential_reference_procedure(ad_sc, ad_sp, folds, mode, 5, **kw)
    cv = tg.cross_val(ad_sc, ad_sp, mode=mode, num_epochs=5, device="cpu", cv_mode="10fold", gemm_precision="fp32", **kw)
    np.testing.assert_allclose(cv["avg_test_score"], t_ref.mean(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(cv["avg_train_score"], tr_ref.mean(), rtol=0, atol=1e-7)


def test_scores_do_not_depend_on_how_many_folds_share_a_launch(sim):
    """`folds_per_launch` (and the cap `_folds_resident` puts on it by free device memory) decides how many folds share a `tg_batch`,
    never what a fold computes: the cross-validation scores and predictions for groups of 1, 3 and all folds are the same bits."""
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(C=60, K=7, V=40, seed=5)
    kw = dict(cluster_label="subclass_label", mode="clusters", num_epochs=5, device="cpu", cv_mode="loo", return_gene_pred=True,
              random_state=3, density_prior="rna_count_based", gemm_precision="fp32")
    runs = [tg.cross_val(ad_sc, ad_sp, folds_per_launch=n, **kw) for n in (1, 3, 16)]
    for cv, ad_ge, df in runs[1:]:
        assert cv == runs[0][0]
        np.testing.assert_array_equal(ad_ge.X, runs[0][1].X)
        np.testing.assert_array_equal(df["score"].to_numpy(), runs[0][2]["score"].to_numpy())


def test_fold_footprint_comes_from_the_library(sim):
    """The resident-fold cap is sized by `tg_query_sizes` for the fold's shape (round-4 advisor finding: it was a hand formula)."""
    import ctypes as ct
    from tangram_amd import _capi
    from tangram_amd.cross_validation import _fold_footprint
    cfg = _capi.TgConfig()
    cfg.abi_version, cfg.precision = _capi.TG_ABI_VERSION, _capi.PRECISIONS["bf16x3"]
    cfg.n_cells, cfg.n_genes, cfg.n_spots, cfg.n_spots_total = 18, 250, 9852, 9852
    cfg.has_density, cfg.lambda_g1, cfg.lambda_d, cfg.beta1, cfg.beta2, cfg.eps = 1, 1.0, 1.0, 0.9, 0.999, 1e-8
    sizes = _capi.TgSizes()
    assert _capi.lib().tg_query_sizes(ct.byref(cfg), ct.byref(sizes)) == 0
    fp = _fold_footprint(18, 9852, 250, "bf16x3", False)
    assert fp == sizes.state_bytes + sizes.workspace_bytes + 2 * 4 * 18 * 9852 and fp > 3 * 4 * 18 * 9852


def test_argument_errors_are_those_of_map_cells_to_space(sim):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas(K=10)
    with pytest.raises(ValueError, match="cluster_label must be specified"):
        tg.cross_val(ad_sc, ad_sp, device="cpu")                                   # mode defaults to 'clusters' (:507)
    with pytest.raises(ValueError, match="lambda_g1 cannot be 0"):
        tg.cross_val(ad_sc, ad_sp, mode="cells", lambda_g1=0, device="cpu")
    with pytest.raises(ValueError, match="define the density_prior"):
        tg.cross_val(ad_sc, ad_sp, mode="cells", lambda_d=1, device="cpu")
    with pytest.raises(ValueError, match="target_count"):
        tg.cross_val(ad_sc, ad_sp, mode="constrained", device="cpu")


def _cv_worker(rank, world, port, sim_path, outdir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        import tangram_amd as tg
        ad_sc, ad_sp = _adatas(C=50, K=9, V=33, seed=6)
        cv, ad_ge, df = tg.cross_val(ad_sc, ad_sp, cluster_label="subclass_label", mode="clusters", num_epochs=5, device="cpu",
                                     return_gene_pred=True, random_state=2, gemm_precision="fp32", folds_per_launch=3, distributed=True)
        np.savez(os.path.join(outdir, f"cv_{rank}.npz"), test=cv["avg_test_score"], train=cv["avg_train_score"], X=ad_ge.X,
                 score=df["score"].to_numpy(), genes=np.array(list(df.index)))
    finally:
        dist.destroy_process_group()


def test_folds_dealt_over_two_ranks_give_the_single_process_result(sim, tmp_path):
    """f-3 "across GPUs": world-2 gloo run, fold i on rank i mod 2, one all-gather of the per-fold records at the end."""
    import torch.multiprocessing as mp
    import tangram_amd as tg
    from tests.test_sharded_gloo import _free_port
    mp.spawn(_cv_worker, args=(2, _free_port(), sim, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "cv_0.npz"), np.load(tmp_path / "cv_1.npz")
    for k in z0.files:
        np.testing.assert_array_equal(z0[k], z1[k], err_msg=k)
    ad_sc, ad_sp = _adatas(C=50, K=9, V=33, seed=6)
    cv, ad_ge, df = tg.cross_val(ad_sc, ad_sp, cluster_label="subclass_label", mode="clusters", num_epochs=5, device="cpu",
                                 return_gene_pred=True, random_state=2, gemm_precision="fp32", folds_per_launch=3)
    assert float(z0["test"]) == cv["avg_test_score"] and float(z0["train"]) == cv["avg_train_score"]
    np.testing.assert_array_equal(z0["X"], ad_ge.X)
    np.testing.assert_array_equal(z0["score"], df["score"].to_numpy())
    assert list(z0["genes"]) == list(df.index)
    with pytest.raises(RuntimeError, match="initialised torch.distributed"):
        tg.cross_val(ad_sc, ad_sp, cluster_label="subclass_label", device="cpu", distributed=True)
