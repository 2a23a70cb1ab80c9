"""rocprofv3 target: B clusters-mode folds stepped as one tg_batch (per-kernel durations of the batched launches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tangram_amd.mapping_optimizer as mo
from tangram_amd.batched import MapperBatch
from tangram_amd.synthetic import make_workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
C, K, V = 18, 250, 9852
w = make_workload(C, K + B, V, dev, seed=1)
S_all, G_all, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
ds = np.full(C, 1.0 / C, np.float32)
ms = [mo.Mapper(S=S_all[:, [g for g in range(K + B) if g != i][:K]], G=G_all[:, [g for g in range(K + B) if g != i][:K]], d=d, d_source=ds,
                lambda_d=1, device=dev, random_state=i + 1) for i in range(B)]
if B > 1:
    b = MapperBatch(ms)
    b.step(300, 0.1)
else:
    ms[0]._engine.step(300, 0.1)
torch.cuda.synchronize()
