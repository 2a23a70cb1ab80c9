"""Per-rank cost of the spot-sharded multi-GPU step on ONE GPU: rank 0 of an 8-way split of cfg2 (30k x 1k x 1250 of
10 000 spots) driven through the real ShardedMapperEngine (the C library issues kernels + RCCL collectives on a 1-rank
group), so the host-side enqueue cost and the kernel time of a shard are both visible."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tangram_amd.sharded import ShardedMapperEngine  # noqa: E402
from tangram_amd.synthetic import make_workload, init_logits  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    # (transport "peer": the library's own one-hop exchange kernels; on ONE rank they push to and read from the rank's own mailbox,
    #  i.e. the kernel's fixed cost without a link -- as the 1-rank RCCL collectives show RCCL's enqueue + local copy)
    # "peer": the exchanges inside the kernels (round 6); "peer_kernels": round 5's form, one exchange kernel between the library's
    # kernels (TG_PEER_FUSED=0)
    cases = ((30000, 1000, 10000, 8, "bf16x3", "rccl"), (30000, 1000, 10000, 8, "bf16x3", "peer"),
             (30000, 1000, 10000, 8, "bf16x3", "peer_kernels"),
             (30000, 1000, 10000, 4, "bf16x3", "rccl"), (30000, 1000, 10000, 4, "bf16x3", "peer"),
             (30000, 1000, 10000, 2, "bf16x3", "rccl"), (30000, 1000, 10000, 2, "bf16x3", "peer"),
             (30000, 1000, 10000, 8, "bf16", "rccl"), (30000, 1000, 10000, 8, "bf16", "peer"))
    # experiments (not in the default list), as arguments "parts,precision,transport+fsN" (the forward cut into N stream-K pieces per gene
    # tile: fwd_splits = -N) or "...+eqN" (N equal ranges per spot tile: fwd_splits = N)
    # "cfg4": BASELINE config 4 as one of its eight ranks sees it -- 200 000 x 2 000 x 6 250 of 50 000 spots, bf16 (logits drawn for the shard only)
    if "cfg4" in sys.argv[1:]:
        cases = ((200000, 2000, 50000, 8, "bf16", "rccl"), (200000, 2000, 50000, 8, "bf16", "peer"))
    exp = [a.split(",") for a in sys.argv[1:] if a.count(",") == 2]
    shp = tuple(int(x) for x in os.environ.get("PROXY_SHAPE", "30000,1000,10000").split(","))      # (C, K, V_total of the experiment cases)
    cases = cases + tuple((shp[0], shp[1], shp[2], int(p_), pr, tn) for p_, pr, tn in exp)
    only = [a for a in sys.argv[1:] if not a.startswith("-") and a.count(",") != 2 and a != "cfg4"]
    if exp and not only:
        only = ["(experiments only)"]
    for (C, K, V, parts, prec, tname) in cases:
        if only and not any(o in f"{parts}_{prec}_{tname}" for o in only) and [str(parts), prec, tname] not in exp:
            continue
        transport = "peer" if tname.startswith("peer") else tname.split("+")[0]
        fs = -int(tname.split("+fs")[1]) if "+fs" in tname else (int(tname.split("+eq")[1]) if "+eq" in tname else 0)
        os.environ["TG_PEER_FUSED"] = "0" if tname == "peer_kernels" else "1"
        Vl = V // parts
        w = make_workload(C, K, V, dev, seed=0)
        M0 = init_logits(C, V, dev, seed=42)[:, :Vl].contiguous() if C * V <= (1 << 29) else init_logits(C, Vl, dev, seed=42)
        e = ShardedMapperEngine(w["S"], w["G"][:Vl].contiguous(), M0, w["d"][:Vl].contiguous(), n_spots_total=V,
                                device=dev, precision=prec, lambdas=dict(lambda_g1=1.0, lambda_d=1.0), transport=transport, fwd_splits=fs)
        n = 100 if C * Vl <= (1 << 29) else 20
        hist = e.eng.new_history(n)
        e.run(10, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(n, 0.1, hist)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        e.eng.profile(True)
        e.run(20, 0.1)
        torch.cuda.synchronize()
        kern = {name: round(1e3 * ms / max(cnt, 1), 1) for name, ms, cnt in e.eng.profile_read()}
        e.eng.profile(False)
        e.peer_check()
        out[f"{C}x{K}x{Vl}_of_{parts}_{prec}_{tname}"] = dict(ms_per_step=1e3 * t_all / n, host_enqueue_ms_per_step=1e3 * t_enq / n,
                                                      main_loss=float(hist[-1, 1]), kernels_us=kern)
        del e, w, M0
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
