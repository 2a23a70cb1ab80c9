"""Reference point: what the vendor GEMM (hipBLASLt / rocBLAS behind torch.matmul) reaches on this box for the shapes of the two
GEMMs of one iteration (bf16 operands), next to the dense peak -- a yardstick for the hand-written kernels' MFMA utilisation."""
import json
import time

import torch

dev = "cuda:0"
out = {}
for name, (m, n, k) in {"backward X = S dGhat^T (30000 x 10000 x 1024)": (30000, 10000, 1024),
                        "forward Ghat = P^T S (10000 x 1024 x 30000)": (10000, 1024, 30000),
                        "square 8192^3": (8192, 8192, 8192)}.items():
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    out[name] = {"ms": 1e3 * dt, "TFLOPs": 2.0 * m * n * k / dt / 1e12}
print(json.dumps(out))
