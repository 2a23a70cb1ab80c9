#!/usr/bin/env python3
"""PROBE (a record, not a test): the UNMODIFIED reference against the library at the FULL cfg2 shape over a LONGER horizon than the
8 epochs of tests/test_gpu_live_reference.py (the round-4 review: "the full-size live comparisons are 8 epochs long; the long-horizon
evidence is only at <= 2 600 cells").

The reference `Mapper` as shipped (oracle/_ref, fp32, torch CPU on the box's host cores, its own seeded logits) trains `--epochs`
epochs at 30 000 x 1 000 x 10 000 (~4.5 s each); the library trains from the same logits on the GPU in split-bf16 (the default
precision).  Recorded every `--every` epochs: the loss terms of both, and at the end the logits (max |dM|, fraction further apart than
1e-3), the mapping (relative Frobenius) and the projection.  The reference cannot be stopped and resumed, so the snapshots along the way
are losses only; the state comparison is at the end.

    python scripts/probes/full_size_long_horizon.py --epochs 50 > gpurun_out/long_horizon.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--shape", default="30000,1000,10000")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--constrained", action="store_true", help="MapperConstrained (BASELINE config 5a) instead of Mapper")
    args = ap.parse_args()
    from oracle import make_ref
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    from tangram_amd.engine import HipMapperEngine
    C, K, V = (int(x) for x in args.shape.split(","))
    n = args.epochs
    ref_mo = make_ref.load()
    torch.set_num_threads(args.threads)
    data = orc.make_synthetic(C, K, V, seed=2)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    t0 = time.perf_counter()
    if args.constrained:
        lam.update(lambda_count=1.0, lambda_f_reg=1.0)
        tc = float(V // 2)
        m = ref_mo.MapperConstrained(S=data["S"], G=data["G"], d=data["d"], device="cpu", random_state=42, target_count=tc, **lam)
        M0, F0 = m.M.detach().numpy().copy(), m.F.detach().numpy().copy()
        t1 = time.perf_counter()
        P_ref, F_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
        ref = {}                               # (stringified with 4 decimals, mapping_optimizer.py:630: the state carries the precision)
    else:
        m = ref_mo.Mapper(S=data["S"], G=data["G"], d=data["d"], device="cpu", random_state=42, **lam)
        M0 = m.M.detach().numpy().copy()
        t1 = time.perf_counter()
        P_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
        ref = {k: np.array([float(x) for x in v], dtype=np.float64) for k, v in hist.items() if len(v)}
    t2 = time.perf_counter()
    M_ref = m.M.detach().numpy()
    if args.constrained:
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device="cuda:0", precision="bf16x3", lambdas=lam,
                            target_count=tc)
    else:
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision="bf16x3", lambdas=lam)
    h = e.new_history(n)
    e.step(n, 0.1, h)
    torch.cuda.synchronize()
    hh = h.cpu().numpy().astype(np.float64)
    cols = dict(total_loss=_capi.H_TOTAL, main_loss=_capi.H_MAIN, vg_reg=_capi.H_VG, kl_reg=_capi.H_KL)
    per_term = {k: float(np.abs(hh[:, j] - ref[k]).max()) for k, j in cols.items() if k in ref}
    by_epoch = {str(ep): {k: float(abs(hh[ep - 1, j] - ref[k][ep - 1])) for k, j in cols.items() if k in ref}
                for ep in sorted(set([1, 2, 4, 8] + list(range(10, n + 1, 10)) + [n])) if ep <= n}
    dM = np.abs(e.logits()[0][:, :V].cpu().numpy() - M_ref)
    res = e.result(with_filter=args.constrained)
    P = (res[0] if args.constrained else res).cpu().numpy()
    S_eff = data["S"] * np.asarray(F_ref)[:, None] if args.constrained else data["S"]
    with torch.no_grad():
        proj_ref = (torch.from_numpy(P_ref).T @ torch.from_numpy(np.ascontiguousarray(S_eff.astype(np.float32)))).numpy()
    proj = e.project().cpu().numpy()
    out = dict(probe="full_size_long_horizon", shape=[C, K, V], epochs=n, precision="bf16x3", terms=lam,
               reference_s_per_epoch=(t2 - t1) / n, reference_init_s=t1 - t0, host_threads=args.threads,
               max_abs_loss_difference_over_all_epochs=per_term, loss_difference_by_epoch=by_epoch,
               final_main_loss=dict(reference=float(ref["main_loss"][-1]) if ref else None, library=float(hh[-1, _capi.H_MAIN])),
               mapper="MapperConstrained" if args.constrained else "Mapper",
               filter_max_abs_difference=float(np.abs(res[1].cpu().numpy() - np.asarray(F_ref)).max()) if args.constrained else None,
               logits=dict(max_dM=float(dM.max()), frac_beyond_1e_3=float((dM > 1e-3).mean()), frac_beyond_1e_4=float((dM > 1e-4).mean()),
                           rms_dM=float(np.sqrt((dM.astype(np.float64) ** 2).mean()))),
               mapping_rel_fro=float(np.linalg.norm((P - P_ref).astype(np.float64)) / np.linalg.norm(P_ref.astype(np.float64))),
               projection_rel_fro=float(np.linalg.norm((proj - proj_ref).astype(np.float64)) / np.linalg.norm(proj_ref.astype(np.float64))),
               argmax_agreement=float((P.argmax(1) == P_ref.argmax(1)).mean()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
