"""Context number: the reference's op sequence (oracle/torch_port.py: softmax, matmul, cosine_similarity, KLDivLoss, autograd,
torch.optim.Adam -- what `tg.map_cells_to_space(device='cuda')` executes) run by PyTorch-ROCm ON THE SAME MI355X at the
BASELINE shape.  Measurement tooling only (like bench.py's cpu_baseline leg); the product never imports the oracle."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_port import TorchPortMapper  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    C, K, V = 30000, 1000, 10000
    w = make_workload(C, K, V, dev, seed=0)
    m = TorchPortMapper(w["S"].cpu().numpy()[:8], w["G"].cpu().numpy()[:8], d=w["d"].cpu().numpy()[:8], lambda_g1=1, lambda_d=1, random_state=1)
    m.S, m.G, m.d = w["S"], w["G"], w["d"]                         # full-size tensors, already on the GPU
    m.M = torch.randn((C, V), device=dev, dtype=torch.float32, requires_grad=True)
    opt = torch.optim.Adam([m.M], lr=0.1)

    def steps(k):                                                  # the body of the reference's train() (mapping_optimizer.py:382-396)
        for _ in range(k):
            total, terms = m.loss()                                # includes the reference's .tolist() host syncs
            opt.zero_grad()
            total.backward()
            opt.step()

    steps(3)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    steps(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"torch_port_on_gpu_fp32": {"ms_per_iter": 1e3 * dt, "iters_per_s": 1.0 / dt,
                                                  "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}}))


if __name__ == "__main__":
    main()
