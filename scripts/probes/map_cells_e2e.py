"""Probe: `map_cells_to_space` end to end at the tutorial scale (26 431 cells x 249 training genes x 9 852 spots, cells mode,
1 000 epochs) with a cProfile of the host side."""
import cProfile, io, json, os, pstats, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd as tg
from tangram_amd.anndata_lite import AnnDataLite
from tangram_amd.synthetic import make_workload
dev = "cuda:0"
C, K, V = 26431, 249, 9852
w = make_workload(C, K + 100, V, dev, seed=2)
S, G = w["S"].cpu().numpy(), w["G"].cpu().numpy()
genes = [f"g{i}" for i in range(K + 100)]
obs_sp = pd.DataFrame({"rna_count_based_density": G.sum(1) / G.sum(), "uniform_density": np.ones(V) / V}, index=[f"s{i}" for i in range(V)])
ad_sc = AnnDataLite(S, obs=pd.DataFrame(index=[f"c{i}" for i in range(C)]), var=pd.DataFrame(index=genes))
ad_sp = AnnDataLite(G, obs=obs_sp, var=pd.DataFrame(index=genes))
for ad in (ad_sc, ad_sp):
    ad.uns["training_genes"] = genes[:K]; ad.uns["overlap_genes"] = genes
tg.map_cells_to_space(ad_sc, ad_sp, mode="cells", device=dev, num_epochs=5, random_state=1, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode="cells", device=dev, num_epochs=1000, random_state=1, verbose=False)
pr.disable(); torch.cuda.synchronize(); t = time.perf_counter() - t0
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(json.dumps({"map_cells_to_space_s": t})); print(s.getvalue()[:5000])
