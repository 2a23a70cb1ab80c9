"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Run in the authoring container (the reference does not travel to the GPU box):

    python oracle/measure_port_ratio.py

Times the UNMODIFIED reference optimizer (/root/reference/tangram/mapping_optimizer.py, loaded standalone like
oracle/gen_golden.py does) and oracle/torch_port.py (the port that bench.py's `cpu_baseline` leg times on the GPU
box) on the SAME inputs, same thread count, interleaved, and writes oracle/port_vs_reference.json.  bench.py reports
the ratio beside `"kind": "port"` so that the port's number can be read as the reference's.
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.tangram_oracle import make_synthetic  # noqa: E402
from oracle.torch_port import TorchPortMapper  # noqa: E402

REF = "/root/reference/tangram/mapping_optimizer.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_mo", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    shapes = [(6000, 500, 2000), (12000, 1000, 4000)]
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    ref = load_ref()
    rows = []
    for C, K, V in shapes:
        d = make_synthetic(C, K, V, seed=0)
        mr = ref.Mapper(S=d["S"], G=d["G"], d=d["d"], lambda_d=1, lambda_g1=1, device="cpu", random_state=42)
        mp = TorchPortMapper(d["S"], d["G"], d=d["d"], lambda_g1=1, lambda_d=1, random_state=42)
        mr.train(num_epochs=1, learning_rate=0.1, print_each=None)          # cold iterations
        mp.train(1, 0.1)
        tr, tp = [], []
        for _ in range(3):                                                  # interleaved rounds
            t0 = time.perf_counter(); mr.train(num_epochs=2, learning_rate=0.1, print_each=None); tr.append((time.perf_counter() - t0) / 2)
            t0 = time.perf_counter(); mp.train(2, 0.1); tp.append((time.perf_counter() - t0) / 2)
        r, p = float(np.median(tr)), float(np.median(tp))
        rows.append({"shape": [C, K, V], "threads": threads, "reference_s_per_iter": r, "port_s_per_iter": p,
                     "port_over_reference": p / r})
        print(rows[-1], flush=True)
    out = {"host": "authoring container, %d vCPU" % threads, "torch": torch.__version__, "rows": rows,
           "port_over_reference_time": float(np.median([x["port_over_reference"] for x in rows])),
           "note": "reference = unmodified tangram/mapping_optimizer.py Mapper.train (note: every train() call rebuilds the "
                   "optimizer like the reference does, mapping_optimizer.py:373); port = oracle/torch_port.py, same inputs"}
    json.dump(out, open(os.path.join(HERE, "port_vs_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
