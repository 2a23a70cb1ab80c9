"""Probe: B folds as ONE tg_batch vs TWO tg_batches of B/2 on two HIP streams, each driven by its own host thread
(do the forward workgroups of one half fill the matrix-core gaps of the other half's backward workgroups?).
18 x 250 x 9 852, engine steps only.   usage: two_stream_batches.py [B]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tangram_amd.mapping_optimizer as mo  # noqa: E402
from tangram_amd.batched import MapperBatch  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C, K, V, EPOCHS = 18, 250, 9852, 1000
w = make_workload(C, K + B, V, dev, seed=1)
S_all, G_all, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
ds = np.full(C, 1.0 / C, np.float32)


def build(i):
    keep = [g for g in range(K + B) if g != i][:K]
    return mo.Mapper(S=S_all[:, keep], G=G_all[:, keep], d=d, d_source=ds, lambda_d=1, device=dev, random_state=i + 1)


def timed(batches, streams):
    def run(b, s, n):
        with torch.cuda.stream(s):
            b.step(n, 0.1)
    for b, s in zip(batches, streams):
        run(b, s, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(b, s, EPOCHS)) for b, s in zip(batches, streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / EPOCHS


out = {}
s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    ms = [build(i) for i in range(B)]
s0.synchronize()
one = MapperBatch(ms)
out["one_batch_us"] = 1e6 * timed([one], [s0])
one.close()
for m in ms:
    m.release()
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    groups = []
    for p, s in enumerate(streams):
        with torch.cuda.stream(s):
            groups.append([build(i) for i in range(p * B // parts, (p + 1) * B // parts)])
        s.synchronize()
    batches = [MapperBatch(g) for g in groups]
    out[f"{parts}_batches_us"] = 1e6 * timed(batches, streams)
    for b in batches:
        b.close()
    for g in groups:
        for m in g:
            m.release()
out["fold_iters_per_s"] = {k: B / (v * 1e-6) for k, v in out.items()}
print(out)
