"""Diagnostic: the same 2-rank sharded run over the peer transport with the exchanges inside the kernels (round 6) and with round 5's
exchange kernels (TG_PEER_FUSED=0); two processes sharing cuda:0.  Prints where logits / history differ."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, port, outdir, fused, n):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TG_PEER_FUSED"] = fused
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tangram_amd.sharded import make_sharded
    from tests.test_gpu_zz_peer_processes import _peer_problem
    data, M0, kw, lam = _peer_problem(False, None)
    sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision="bf16x3", lambdas=lam, transport="peer", **kw)
    hist = sh.eng.new_history(n)
    outs = {}
    for i in range(n):
        sh.run(1, 0.1, hist, i)
        torch.cuda.synchronize()
        ws = sh.eng
        outs[f"M{i}"] = ws.logits()[0][:, : ws.V].cpu().numpy()
    sh.peer_check()
    np.savez(os.path.join(outdir, f"f{fused}_{rank}.npz"), hist=hist.cpu().numpy(), **outs)
    sh.release()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    n, world = 3, 2
    d = tempfile.mkdtemp()
    for fused in ("1", "0"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(worker, args=(world, port, d, fused, n), nprocs=world, join=True)
    for r in range(world):
        a, b = np.load(os.path.join(d, f"f1_{r}.npz")), np.load(os.path.join(d, f"f0_{r}.npz"))
        print("rank", r, "hist equal", np.array_equal(a["hist"], b["hist"], equal_nan=True))
        for i in range(n):
            x, y = a[f"M{i}"], b[f"M{i}"]
            bad = x != y
            rows = np.where(bad.any(1))[0]
            print(f"  step {i}: {bad.sum()} of {bad.size} logits differ, in {len(rows)} of {x.shape[0]} rows (first rows {rows[:8].tolist()}), max |d| {np.abs(x - y).max():.3e}")
