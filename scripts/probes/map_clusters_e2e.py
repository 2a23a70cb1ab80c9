"""Probe: `map_cells_to_space(mode='clusters')` end to end at the tutorial scale (20 000 cells in 18 clusters x 249 genes x 9 852 spots)."""
import cProfile, io, json, os, pstats, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd as tg
from tangram_amd.anndata_lite import AnnDataLite
from tangram_amd.synthetic import make_workload
dev = "cuda:0"
C, K, V = 20000, 249, 9852
w = make_workload(C, K, V, dev, seed=2)
S, G = w["S"].cpu().numpy(), w["G"].cpu().numpy()
genes = [f"g{i}" for i in range(K)]
rng = np.random.default_rng(0)
obs_sc = pd.DataFrame({"cluster": rng.integers(0, 18, C).astype(str)}, index=[f"c{i}" for i in range(C)])
obs_sp = pd.DataFrame({"rna_count_based_density": G.sum(1) / G.sum(), "uniform_density": np.ones(V) / V}, index=[f"s{i}" for i in range(V)])
ad_sc = AnnDataLite(S, obs=obs_sc, var=pd.DataFrame(index=genes))
ad_sp = AnnDataLite(G, obs=obs_sp, var=pd.DataFrame(index=genes))
for ad in (ad_sc, ad_sp):
    ad.uns["training_genes"] = genes; ad.uns["overlap_genes"] = genes
kw = dict(mode="clusters", cluster_label="cluster", device=dev, num_epochs=1000, random_state=1, verbose=False)
for _ in range(2):
    tg.map_cells_to_space(ad_sc, ad_sp, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for _ in range(5):
    ad_map = tg.map_cells_to_space(ad_sc, ad_sp, **kw)
pr.disable(); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(json.dumps({"map_cells_to_space_clusters_s": t})); print(s.getvalue()[:4500])
