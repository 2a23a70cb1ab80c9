"""Probe: what building a Mapper costs at the tutorial scale (26 431 x 249 x 9 852) next to its training."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd.mapping_optimizer as mo
from tangram_amd import host_rng
from tangram_amd.synthetic import make_workload
dev = "cuda:0"
C, K, V = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (26431, 249, 9852)
w = make_workload(C, K, V, dev, seed=1)
S, G, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
out = {"shape": [C, K, V], "host_threads": len(os.sched_getaffinity(0))}
mo.Mapper(S=S, G=G, d=d, lambda_d=1, device=dev, random_state=1).train(num_epochs=3, print_each=None)
np.random.seed(1); t0 = time.perf_counter(); a = np.random.normal(0, 1, (C, V)).astype(np.float32); out["numpy_normal_s"] = time.perf_counter() - t0
np.random.seed(1); t0 = time.perf_counter(); b = host_rng.legacy_normal_f32((C, V)); out["helper_normal_s"] = time.perf_counter() - t0
out["same_bits"] = bool(np.array_equal(a, b))
torch.cuda.synchronize(); t0 = time.perf_counter()
m = mo.Mapper(S=S, G=G, d=d, lambda_d=1, device=dev, random_state=1)
torch.cuda.synchronize(); out["mapper_construction_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); m.train(num_epochs=1000, learning_rate=0.1, print_each=None); torch.cuda.synchronize(); out["train_1000_epochs_s"] = time.perf_counter() - t0
print(json.dumps(out))
