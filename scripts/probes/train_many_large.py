"""Probe: three independent cells-mode mappings of a GPU-filling size (tutorial scale 26 431 x 249 x 9 852) through train_many:
one after the other, batched='auto' (not batched above 2^25 cells x spots), batched=False (a stream + host thread per mapping)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd.mapping_optimizer as mo  # noqa: E402
from tangram_amd.batched import train_many  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402

dev = "cuda:0"
C, K, V = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (26431, 249, 9852)
EPOCHS, B = 200, 3
w = make_workload(C, K, V, dev, seed=1)
S, G, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
builders = [(lambda i=i: mo.Mapper(S=S, G=G, d=d, lambda_d=1, device=dev, random_state=i + 1)) for i in range(B)]
out = {"shape": [C, K, V], "mappings": B, "epochs": EPOCHS}
builders[0]().train(num_epochs=5, learning_rate=0.1, print_each=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in builders:
    m = b()
    m.train(num_epochs=EPOCHS, learning_rate=0.1, print_each=None)
    m.release()
torch.cuda.synchronize()
out["one_after_the_other_s"] = time.perf_counter() - t0
for label, kw in (("auto", dict(batched="auto")), ("streams", dict(batched=False, max_concurrent=3))):
    t0 = time.perf_counter()
    res, ms = train_many(builders, EPOCHS, 0.1, device=dev, **kw)
    torch.cuda.synchronize()
    out[label + "_s"] = time.perf_counter() - t0
    for m in ms:
        m.release()
print(json.dumps(out))
