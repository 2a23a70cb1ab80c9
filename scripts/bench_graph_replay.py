#!/usr/bin/env python3
"""Eager tg_mapper_step vs the same steps replayed from a HIP graph, at the cross-validation unit of the tutorial (clusters mode,
18 x 250 x 9 852) and at a mid-size cells-mode problem: does taking the host launches out change the iteration time?
  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tangram_amd.engine import HipMapperEngine  # noqa: E402
from tangram_amd.synthetic import make_workload, init_logits  # noqa: E402


def run(C, K, V, n=400):
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(s):
        w = make_workload(C, K, V, dev, seed=0)
        e = HipMapperEngine(w["S"], w["G"], init_logits(C, V, dev, seed=42), d=w["d"], device=dev, precision="bf16x3",
                            lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
        h = e.new_history(n)
        e.step(20, 0.1)
        s.synchronize()
        t0 = time.perf_counter(); e.step(n, 0.1, h); s.synchronize()
        out["eager_us_per_iter"] = 1e6 * (time.perf_counter() - t0) / n
        e.set_step(0)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            e.step(n, 0.1, h)
        e.set_step(0)
        g.replay(); s.synchronize()
        t0 = time.perf_counter(); g.replay(); s.synchronize()
        out["graph_replay_us_per_iter"] = 1e6 * (time.perf_counter() - t0) / n
        out["last_total_loss"] = float(h[-1, 0].item())
        e.release()
    return out


if __name__ == "__main__":
    res = {"clusters_18x250x9852": run(18, 250, 9852), "cells_4200x1000x1500": run(4200, 1000, 1500, 200)}
    print(json.dumps(res))
