#!/usr/bin/env python3
"""PROBE (a record, not a test): BASELINE config 3 as deployed -- 30 000 x 1 000 x 10 000 on EIGHT ranks of 1 250 spots, one process per
rank, exchanges over the peer-memory transport -- against the UNMODIFIED reference's single-process run, over a longer horizon than the
suite's 8 epochs.  The one substitution a 1-GPU box forces: the eight processes share cuda:0 (their mailboxes still cross process
boundaries as hipIpc handles).  Recorded: the global history of rank 0 against the reference's, every rank's history equal to rank 0's,
the assembled logits and mapping against the reference's.

    python scripts/probes/full_size_shards_long_horizon.py --epochs 30 > gpurun_out/shards_long_horizon.json
"""
import argparse
import json
import os
import socket
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SHAPE = (30000, 1000, 10000)
LAM = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)


def worker(rank, world, port, outdir, n, m0_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TG_PEER_TIMEOUT_MS"] = "30000"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import tangram_oracle as orc
        from tangram_amd.sharded import make_sharded
        C, K, V = SHAPE
        data = orc.make_synthetic(C, K, V, seed=2)
        M0 = np.load(m0_path, mmap_mode="r")
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision="bf16x3", lambdas=LAM, transport="peer")
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist, 0)
        torch.cuda.synchronize()
        sh.peer_check()
        P, (lo, hi) = sh.result_local()
        np.savez(os.path.join(outdir, f"rank_{rank}.npz"), hist=hist.cpu().numpy(), M=sh.eng.logits()[0][:, : sh.eng.V].cpu().numpy(),
                 P=P.cpu().numpy(), lo=lo, hi=hi)
        sh.release()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--threads", type=int, default=32)
    args = ap.parse_args()
    import torch
    import torch.multiprocessing as mp
    from oracle import make_ref
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    C, K, V = SHAPE
    n, world = args.epochs, args.world
    torch.set_num_threads(args.threads)
    ref_mo = make_ref.load()
    data = orc.make_synthetic(C, K, V, seed=2)
    m = ref_mo.Mapper(S=data["S"], G=data["G"], d=data["d"], device="cpu", random_state=42, **LAM)
    M0 = m.M.detach().numpy().copy()
    t0 = time.perf_counter()
    P_ref, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
    t_ref = (time.perf_counter() - t0) / n
    M_ref = m.M.detach().numpy()
    ref = {k: np.array([float(x) for x in v], dtype=np.float64) for k, v in hist.items() if len(v)}
    with tempfile.TemporaryDirectory() as tmp:
        m0_path = os.path.join(tmp, "M0.npy")
        np.save(m0_path, M0)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        t0 = time.perf_counter()
        mp.spawn(worker, args=(world, port, tmp, n, m0_path), nprocs=world, join=True)
        t_sh = time.perf_counter() - t0
        z = [np.load(os.path.join(tmp, f"rank_{r}.npz")) for r in range(world)]
        same_hist = all(np.array_equal(z[r]["hist"], z[0]["hist"], equal_nan=True) for r in range(1, world))      # (disabled terms are NaN, like the reference)
        hh = z[0]["hist"].astype(np.float64)
        M = np.concatenate([z[r]["M"] for r in range(world)], axis=1)
        P = np.concatenate([z[r]["P"] for r in range(world)], axis=1)
        ranges = [(int(z[r]["lo"]), int(z[r]["hi"])) for r in range(world)]
    cols = dict(total_loss=_capi.H_TOTAL, main_loss=_capi.H_MAIN, vg_reg=_capi.H_VG, kl_reg=_capi.H_KL)
    dM = np.abs(M - M_ref)
    out = dict(probe="full_size_shards_long_horizon", shape=[C, K, V], world=world, epochs=n, transport="peer (hipIpc, processes sharing cuda:0)",
               reference_s_per_epoch=t_ref, sharded_wall_s_including_process_start=t_sh, spot_ranges=ranges,
               global_history_identical_on_every_rank=bool(same_hist),
               max_abs_loss_difference_over_all_epochs={k: float(np.abs(hh[:, j] - ref[k]).max()) for k, j in cols.items() if k in ref},
               logits=dict(max_dM=float(dM.max()), frac_beyond_1e_3=float((dM > 1e-3).mean()), frac_beyond_1e_4=float((dM > 1e-4).mean()),
                           rms_dM=float(np.sqrt((dM.astype(np.float64) ** 2).mean()))),
               mapping_rel_fro=float(np.linalg.norm((P - P_ref).astype(np.float64)) / np.linalg.norm(P_ref.astype(np.float64))),
               argmax_agreement=float((P.argmax(1) == P_ref.argmax(1)).mean()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
