"""The reference's own test functions, replayed through the drop-in wrapper on the GPU (-m gpu).

tests/tangram_test.py:67-103 (`test_map_cells_to_space`: 9 parameter rows, mode='clusters', 500 epochs, random_state=42,
asserts round(ad_map.X[0, 0], 3)) and :159-210 (`test_train_score_match`: 6 rows, average training score of
`project_genes` + `compare_spatial_geneexp` == last `main_loss`, 3 decimals).  The reference's h5ad inputs are missing
from the checkout, so the cells are synthetic (oracle/gen_golden.py::cluster_inputs) and the expected values come from
runs of the UNMODIFIED reference optimizer on the same inputs (tests/golden/grid_*.npz)."""
import numpy as np
import pandas as pd
import pytest

from oracle.gen_golden import CASES, GRID_CELLS, cluster_inputs
from tangram_amd.anndata_lite import AnnDataLite
from tests import parity_common as pc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CL, K, V, SEED = 12, 60, 80, 11


def _fixture_name(lambda_g2, lambda_d, density_prior, scale):
    # clusters mode: lambda_d is forced to >= 1 and the prior falls back to uniform (mapping_utils.py:293-307)
    ld = lambda_d if lambda_d else 1
    prior = "rna" if density_prior == "rna_count_based" else "uniform"
    name = f"grid_g2_{lambda_g2}_d{ld}_{prior}_{'scaled' if scale else 'unscaled'}"
    assert name in CASES, name
    return name


def _adatas():
    ci = cluster_inputs(N_CL, K, V, SEED, True, "uniform")
    genes = [f"gene{i}" for i in range(K)]
    obs_sc = pd.DataFrame({"subclass_label": [f"ct{l:02d}" for l in ci["labels"]]}, index=[f"cell{i}" for i in range(GRID_CELLS)])
    G = ci["G"]
    obs_sp = pd.DataFrame({"rna_count_based_density": G.sum(1) / G.sum(), "uniform_density": np.ones(V) / V},
                          index=[f"spot{i}" for i in range(V)])
    ad_sc = AnnDataLite(ci["S_cells"].copy(), obs=obs_sc, var=pd.DataFrame(index=genes))
    ad_sp = AnnDataLite(G.copy(), obs=obs_sp, var=pd.DataFrame(index=genes))
    for ad in (ad_sc, ad_sp):                 # what pp_adatas leaves behind (mapping_utils.py:74-85)
        ad.uns["training_genes"] = genes
        ad.uns["overlap_genes"] = genes
    return ad_sc, ad_sp


@pytest.mark.parametrize("lambda_g1, lambda_g2, lambda_d, density_prior, scale", [
    (1, 0, 0, None, True), (1, 0, 0, None, False), (1, 1, 0, None, True), (1, 1, 0, None, False),
    (1, 1, 1, "uniform", True), (1, 1, 1, "uniform", False), (1, 0, 2, "uniform", True),
    (1, 0, 1, "rna_count_based", True), (1, 0, 1, "uniform", True),
])                                            # the 9 rows of tests/tangram_test.py:67-80
def test_map_cells_to_space(lambda_g1, lambda_g2, lambda_d, density_prior, scale):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas()
    ad_map = tg.map_cells_to_space(adata_sc=ad_sc, adata_sp=ad_sp, device=DEV, mode="clusters", cluster_label="subclass_label",
                                   lambda_g1=lambda_g1, lambda_g2=lambda_g2, lambda_d=lambda_d, density_prior=density_prior,
                                   scale=scale, random_state=42, num_epochs=500, verbose=False)
    z = pc.load_golden(_fixture_name(lambda_g2, lambda_d, density_prior, scale))
    # the reference's assertion (:103), against the reference's own fp32 result on these inputs
    assert round(float(ad_map.X[0, 0]), 3) == round(float(z["f32_P"][0, 0]), 3)
    # and the whole mapping: within 5x of what fp32 costs the reference itself over these 500 epochs
    bound = max(2e-4, 5.0 * float(np.abs(z["f32_P"] - z["f64_P"]).max()))
    assert float(np.abs(ad_map.X - z["f64_P"]).max()) <= bound
    assert list(ad_map.obs["cluster_density"].index) == list(ad_map.obs.index) and ad_map.X.shape == (N_CL, V)
    hist = ad_map.uns["training_history"]
    assert len(hist["main_loss"]) == 500
    np.testing.assert_allclose(np.array(hist["total_loss"], dtype=np.float64), z["f64_hist_total_loss"],
                               atol=max(1e-5, 5.0 * float(np.abs(z["f32_hist_total_loss"] - z["f64_hist_total_loss"]).max())))


@pytest.mark.parametrize("lambda_g1, lambda_g2, lambda_d, density_prior, scale", [
    (1, 0, 0, None, True), (1, 0, 0, None, False), (1, 1, 0, None, True), (1, 1, 0, None, False),
    (1, 0, 1, "uniform", True), (1, 0, 1, "rna_count_based", False),
])                                            # the 6 rows of tests/tangram_test.py:159-169
def test_train_score_match(lambda_g1, lambda_g2, lambda_d, density_prior, scale):
    import tangram_amd as tg
    ad_sc, ad_sp = _adatas()
    ad_map = tg.map_cells_to_space(adata_sc=ad_sc, adata_sp=ad_sp, device=DEV, mode="clusters", cluster_label="subclass_label",
                                   lambda_g1=lambda_g1, lambda_g2=lambda_g2, lambda_d=lambda_d, density_prior=density_prior,
                                   scale=scale, random_state=42, num_epochs=500, verbose=False)
    ad_ge = tg.project_genes(adata_map=ad_map, adata_sc=ad_sc, cluster_label="subclass_label", scale=scale, device=DEV)
    # compare_spatial_geneexp (utils.py:378-460): per-gene cosine similarity between predicted and measured spatial expression
    Gp, G = np.asarray(ad_ge.X, dtype=np.float64), np.asarray(ad_sp.X, dtype=np.float64)
    score = (Gp * G).sum(0) / (np.linalg.norm(Gp, axis=0) * np.linalg.norm(G, axis=0))
    is_training = ad_ge.var["is_training"].to_numpy()
    avg_score_df = round(float(score[is_training].mean()), 3)
    avg_score_train_hist = round(float(list(ad_map.uns["training_history"]["main_loss"])[-1]), 3)
    assert avg_score_df == pytest.approx(avg_score_train_hist, abs=1.001e-3)      # rounding boundary: the reference compares the two rounded values
