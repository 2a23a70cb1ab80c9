"""N independent clusters-mode mappings (18 clusters x 250 genes x 9852 spots, 1000 epochs: the tutorial's
cross-validation unit) one after the other vs side by side on one GPU (tangram_amd.batched.train_many)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tangram_amd.mapping_optimizer as mo  # noqa: E402
from tangram_amd.batched import train_many  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402


def main():
    dev = "cuda:0"
    C, K, V, N, EPOCHS = 18, 250, 9852, 16, 1000
    w = make_workload(C, K + N, V, dev, seed=1)
    S_all, G_all, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
    ds = np.full(C, 1.0 / C, np.float32)

    def builder(i):          # leave-one-gene-out fold i (cross_val, utils.py:576-600)
        keep = [g for g in range(K + N) if g != i][:K]
        return lambda: mo.Mapper(S=S_all[:, keep], G=G_all[:, keep], d=d, d_source=ds, lambda_d=1, device=dev, random_state=i + 1)     # (0 would mean "unseeded", like the reference)

    builders = [builder(i) for i in range(N)]
    builders[0]().train(num_epochs=50, learning_rate=0.1, print_each=None)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = [b().train(num_epochs=EPOCHS, learning_rate=0.1, print_each=None) for b in builders]
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    out = {"mappings": N, "epochs": EPOCHS, "shape": [C, K, V], "sequential_s": t_seq, "sequential_us_per_iter": 1e6 * t_seq / (N * EPOCHS)}
    for conc in (2, 4, 8, 16):
        t0 = time.perf_counter()
        res, _ = train_many(builders, EPOCHS, 0.1, max_concurrent=conc, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        same = all(np.array_equal(a[0], b[0]) for a, b in zip(seq, res))
        out[f"concurrent_{conc}"] = {"seconds": t, "us_per_iter": 1e6 * t / (N * EPOCHS), "speedup": t_seq / t, "bit_identical": bool(same)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
