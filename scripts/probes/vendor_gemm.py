"""Yardstick: the two GEMM shapes of cfg2 (30 000 cells x 1 000 genes x 10 000 spots) through the vendor library
(torch.matmul -> hipBLASLt / rocBLAS) in bf16 and fp32, next to this library's own kernels (bench.py reports those).
Plain GEMMs only: no softmax in the operand path, no statistics, no row dots -- a ceiling for what a library call could give."""
import json
import time

import torch

dev = "cuda:0"
C, K, V = 30000, 1000, 10000
out = {}
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    P = torch.rand(C, V, device=dev, dtype=dt)
    S = torch.rand(C, K, device=dev, dtype=dt)
    dG = torch.rand(V, K, device=dev, dtype=dt)
    for label, fn in (("forward  Ghat = P^T S   [V x C] x [C x K]", lambda: P.t() @ S), ("backward X = S dGhat^T [C x K] x [K x V]", lambda: S @ dG.t())):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[f"{name} {label}"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * C * K * V / ms / 1e9, 1)}
print(json.dumps(out, indent=1))
