"""B independent clusters-mode mappings (18 clusters x 250 genes x 9852 spots: the tutorial's cross-validation
unit, utils.py:576-600) -- one fold alone vs B folds in ONE launch per kernel (tg_batch, blockIdx.z = fold) vs one HIP stream +
host thread per fold.  Timed on engine steps (construction / result copies excluded), then end to end through train_many."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tangram_amd.mapping_optimizer as mo  # noqa: E402
from tangram_amd.batched import train_many, MapperBatch  # noqa: E402
from tangram_amd.synthetic import make_workload  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,4,8,16", help="batch sizes of the stepping-rate part")
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--no-e2e", action="store_true", help="skip the train_many part (e.g. under rocprofv3)")
    ap.add_argument("--no-single", action="store_true", help="skip the one-fold stepping rate (a profile of one batch size only)")
    ap.add_argument("--constrained", action="store_true", help="MapperConstrained folds instead of Mapper")
    opt = ap.parse_args()
    dev = "cuda:0"
    C, K, V, N, EPOCHS = 18, 250, 9852, 16, opt.epochs
    w = make_workload(C, K + N, V, dev, seed=1)
    S_all, G_all, d = w["S"].cpu().numpy(), w["G"].cpu().numpy(), w["d"].cpu().numpy()
    ds = np.full(C, 1.0 / C, np.float32)

    def builder(i):          # leave-one-gene-out fold i (cross_val, utils.py:576-600)
        keep = [g for g in range(K + N) if g != i][:K]
        if opt.constrained:
            return lambda: mo.MapperConstrained(S=S_all[:, keep], G=G_all[:, keep], d=d, lambda_d=1, lambda_count=1, lambda_f_reg=1, target_count=C // 2,
                                                device=dev, random_state=i + 1)
        return lambda: mo.Mapper(S=S_all[:, keep], G=G_all[:, keep], d=d, d_source=ds, lambda_d=1, device=dev, random_state=i + 1)     # (0 would mean "unseeded", like the reference)

    out = {"epochs": EPOCHS, "shape": [C, K, V]}
    # --- pure stepping rate: one fold alone, then B folds per launch
    t1 = float("nan")
    if not opt.no_single:
        m1 = builder(0)()
        m1._engine.step(100, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m1._engine.step(EPOCHS, 0.1)
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / EPOCHS
        out["one_fold_us_per_iter"] = 1e6 * t1
    for B in [int(x) for x in opt.batches.split(",") if x]:
        ms = [builder(i)() for i in range(B)]
        batch = MapperBatch(ms)
        batch.step(100, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.step(EPOCHS, 0.1)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / EPOCHS
        out[f"batch_{B}"] = {"us_per_batch_iter": 1e6 * tb, "fold_iters_per_s": B / tb, "speedup_vs_one_fold": B * t1 / tb}
        batch.close()
        for m in ms:
            m.release()
    if opt.no_e2e:
        print(json.dumps(out))
        return
    # --- end to end through train_many (construction, training, result copies), 16 folds
    builders = [builder(i) for i in range(N)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = [b().train(num_epochs=EPOCHS, learning_rate=0.1, print_each=None) for b in builders]
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    out["train_sequential_s"] = t_seq
    for label, kw in (("train_many_batched", dict(batched="auto")), ("train_many_streams_4", dict(batched=False, max_concurrent=4))):
        t0 = time.perf_counter()
        res, ms = train_many(builders, EPOCHS, 0.1, device=dev, **kw)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        same = all(np.array_equal(a[0], b[0]) and list(a[1]["main_loss"]) == list(b[1]["main_loss"]) for a, b in zip(seq, res))
        out[label] = {"seconds": t, "speedup": t_seq / t, "bit_identical_to_sequential": bool(same)}
        for m in ms:
            m.release()
    # --- the reference's tuning caller (mapping_parameter_tuning.py:110-129): three seeds of ONE problem with a validation split and
    # val_each = 1, one after the other (what the reference does) vs in one tg_batch that pauses at every validation epoch
    tr, va = np.arange(0, K - 10), np.arange(K - 10, K)
    def seed_builder(seed):
        return lambda: mo.Mapper(S=S_all[:, :K], G=G_all[:, :K], d=d, d_source=ds, lambda_d=1, train_genes_idx=tr, val_genes_idx=va, device=dev, random_state=seed)
    sb = [seed_builder(s) for s in (1, 2, 3)]
    ep = min(EPOCHS, 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = [b().train(num_epochs=ep, learning_rate=0.1, print_each=None, val_each=1) for b in sb]
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    res, ms = train_many(sb, ep, 0.1, device=dev, val_each=1)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    same = all(np.array_equal(a[0], b[0]) and all(list(map(float, a[1][k])) == list(map(float, b[1][k])) for k in ("main_loss", "val_gene_sim")) for a, b in zip(seq, res))
    out["tuning_three_seeds_val_each_1"] = {"epochs": ep, "sequential_s": t_seq, "train_many_batched_s": t, "speedup": t_seq / t, "bit_identical_to_sequential": bool(same)}
    for m in ms:
        m.release()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
