"""Probe: B independent cells-mode mappings of the cfg2 shape (30 000 x 1 000 x 10 000) stepped as a tg_batch (groups on streams of
their own: one mapping's HBM-bound update can run beside another's MFMA-bound GEMMs) vs one after the other.   usage: batch_large.py [C K V]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd.mapping_optimizer as mo  # noqa: E402
from tangram_amd.batched import MapperBatch  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402

dev = "cuda:0"
C, K, V = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (30000, 1000, 10000)
STEPS = 30
w = make_workload(C, K, V, dev, seed=1)
out = {"shape": [C, K, V]}


def build(i):
    return mo.Mapper(S=w["S"], G=w["G"], d=w["d"], lambda_d=1, device=dev, random_state=i + 1)


m = build(0)
m._engine.step(5, 0.1)
torch.cuda.synchronize()
t0 = time.perf_counter()
m._engine.step(STEPS, 0.1)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / STEPS
out["one_mapping_ms_per_iter"] = 1e3 * t1
m.release()
for B in (2, 3, 4):
    ms = [build(i) for i in range(B)]
    b = MapperBatch(ms)
    b.step(5, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.step(STEPS, 0.1)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / STEPS
    out[f"batch_{B}"] = {"ms_per_batch_iter": 1e3 * tb, "ms_per_mapping_iter": 1e3 * tb / B, "vs_one_after_the_other": B * t1 / tb}
    b.close()
    for x in ms:
        x.release()
print(json.dumps(out))
