#!/usr/bin/env python
"""Benchmark of the Tangram mapping hot path on MI355X (BASELINE.json metric: mapping iterations/s and
cell*spot*gene/s at 30k cells x 1k genes x 10k spots).

    python bench.py --gpus N --steps K --warmup W [--precision bf16x3|bf16|fp32]

One "step" = one full mapping iteration (softmax, P^T S, cosine + density loss, backward, Adam) on synthetic
inputs of the named shape that are resident in HBM before the timed region starts.  N > 1 is launched by
torch.distributed.run (one rank per GPU, RCCL): the spots are sharded over the ranks (strong scaling: the
problem is fixed, `value` is whole-job iterations/s).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12                      # B/s   (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
# dense matrix-core peak for ONE algorithmic product: bf16 2.5 PFLOP/s; split-bf16 x3 issues three bf16 MFMAs per product, i.e.
# 833 TFLOP/s effective (the figure SURVEY 8(d) prices the bf16x3 roofline with); exact f32 MFMA 157.3 TFLOP/s
MFMA_PEAK = {"bf16": 2.5e15, "bf16x3": 2.5e15 / 3.0, "fp32": 157.3e12}
DTYPE_NAME = {"bf16x3": "f32 (split-bf16 x3 MFMA operands, f32 accumulate and state: fp32-parity)",
              "bf16": "bf16 MFMA operands, f32 accumulate and state", "fp32": "f32 (exact f32 MFMA)"}
WORKLOADS = {"cfg2": (30000, 1000, 10000)}


def kernel_model(name, C, K, V):
    """Algorithmic bytes / flops of ONE launch (per-unit figures of SURVEY 8d)."""
    gemm = 2.0 * C * V * K
    small = 4.0 * (C * K + V * K)
    if name == "tg_fwd_kernel":
        return 4.0 * C * V + small, gemm            # read M once, S once, write Ghat
    if name == "tg_bwd_kernel":
        return 4.0 * C * V + small, gemm            # S and dGhat once, one C x V fp32 plane (X out; M in on the sharded path)
    if name in ("tg_adam_rowpass", "tg_adam_update"):
        return 24.0 * C * V, 0.0                    # read+write M, Adam m, Adam v (the X read is implementation traffic)
    return None, None


def pmc_traffic(kernel, precision):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, plus WRITE_SIZE), or None."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return tab[precision][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(C, K, V, seed=0):
    """The reference's CPU path (PyTorch port in oracle/torch_port.py, same op sequence) on a bounded sample."""
    from oracle.torch_port import TorchPortMapper
    from tangram_amd.synthetic import make_workload
    Cs, Vs = max(C // 5, 64), max(V // 5, 64)              # 1/25 of the C*V plane, all K genes
    torch.set_num_threads(os.cpu_count() or 1)
    w = make_workload(Cs, K, Vs, "cpu", seed=seed)
    m = TorchPortMapper(w["S"].numpy(), w["G"].numpy(), d=w["d"].numpy(), lambda_g1=1, lambda_d=1, random_state=42)
    m.train(1, 0.1)                                        # warm-up (cold first iteration)
    n = 3
    t0 = time.perf_counter()
    m.train(n, 0.1)
    dt = (time.perf_counter() - t0) / n
    csg = Cs * K * Vs / dt
    return {"value": csg / (float(C) * K * V), "unit": "iters/s (cell*spot*gene/s of the sample / C*K*V of the workload)",
            "cell_spot_gene_per_s": csg, "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} iterations of oracle/torch_port.py (PyTorch-CPU port of the reference loop, fp32) at "
                      f"{Cs}x{K}x{Vs} = 1/25 of the cell x spot plane, {dt:.2f} s/iter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "fp32"])
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--shape", default=None, help="override C,K,V (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short runs of the other GEMM precisions")
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0, help="force the GEMM tile edge (128 or 256); 0 = automatic")
    ap.add_argument("--bands", type=int, default=0, help="cell bands of the opt-in 3-stream pipeline (0/1 = sequential schedule, the default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("TG_BENCH_BACKEND", "nccl")       # "gloo": plumbing smoke test of the N > 1 path on a 1-GPU box
    if world > 1 and backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks need {world} GPUs (found {ndev}); RCCL cannot share a device between ranks")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import ShardedMapperEngine, shard_bounds
    from tangram_amd.synthetic import make_workload, init_logits

    C, K, V = WORKLOADS[args.workload] if not args.shape else tuple(int(x) for x in args.shape.split(","))
    lam = dict(lambda_g1=1.0, lambda_d=1.0)        # mode='cells' defaults as resolved by mapping_utils.py:214-215
    w = make_workload(C, K, V, device, seed=0)
    lr = 0.1
    if world == 1:
        M0 = init_logits(C, V, device, seed=42)
        eng = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=device, precision=args.precision, lambdas=lam,
                              fwd_splits=args.splits, tile_size=args.tile, pipeline_bands=args.bands)
        del M0
        run = lambda n: eng.step(n, lr)
        core = eng
    else:
        lo, hi = shard_bounds(V, world, rank)
        M0 = init_logits(C, hi - lo, device, seed=42 + rank)
        sh = ShardedMapperEngine(w["S"], w["G"][lo:hi].contiguous(), M0, w["d"][lo:hi].contiguous(), n_spots_total=V,
                                 device=device, precision=args.precision, lambdas=lam, fwd_splits=args.splits, tile_size=args.tile)
        del M0
        run = lambda n: sh.run(n, lr)
        core = sh.eng
    torch.cuda.empty_cache()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)                                 # timed region: the product schedule
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel durations: HIP events after every kernel on the kernel's stream.  Event-bracketing needs the kernels on
    # ONE stream, so this pass runs the sequential schedule (same kernels, same launches, no overlap between them).
    nprof = max(4, min(args.steps, 20))
    core.profile(True)
    t1 = time.perf_counter()
    run(nprof)
    fence()
    seq_elapsed = time.perf_counter() - t1
    prof = core.profile_read()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the loss of the last step is finite (nothing was skipped)
    hist = core.new_history(1)
    if world == 1:
        eng.step(1, lr, hist)
    else:
        sh.run(1, lr, hist)
    torch.cuda.synchronize(device)
    main_loss = float(hist[0, 1].item())

    # the other GEMM precisions on the same inputs (reported beside the headline, never as `value`)
    alt = {}
    if world == 1 and not args.no_alt:
        del eng, core
        torch.cuda.empty_cache()
        for prec in [p for p in ("bf16x3", "bf16", "fp32") if p != args.precision]:
            M0 = init_logits(C, V, device, seed=42)
            e2 = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=device, precision=prec, lambdas=lam)
            del M0
            n2 = max(4, args.steps // 4)
            e2.step(2, lr)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            e2.step(n2, lr)
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t1) / n2
            alt[prec] = {"value": 1.0 / dt, "unit": "iters/s", "ms_per_step": 1e3 * dt, "steps": n2, "dtype": DTYPE_NAME[prec]}
            e2.close()
            del e2
            torch.cuda.empty_cache()
    del w

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        its = args.steps / elapsed
        Vl = V if world == 1 else (shard_bounds(V, world, 0)[1])
        kern = []
        for name, ms, cnt in prof:
            b, f = kernel_model(name, C, K, Vl)
            kern.append({"name": name, "avg_ms": ms / max(cnt, 1), "launches": cnt,
                         "alg_GB": None if b is None else b / 1e9, "alg_GFLOP": None if f is None else f / 1e9})
        dom = max((k for k in kern if k["alg_GB"] is not None), key=lambda k: k["avg_ms"] * k["launches"], default=None)
        roof = None
        # the committed PMC table was collected on the full single-GPU cfg2 launch: it does not describe a shard or another shape
        traffic_of = (lambda k: pmc_traffic(k, args.precision)) if (world == 1 and not args.shape) else (lambda k: None)
        if dom is not None:
            t = dom["avg_ms"] * 1e-3
            t_h = dom["alg_GB"] * 1e9 / HBM_PEAK
            t_m = dom["alg_GFLOP"] * 1e9 / MFMA_PEAK[args.precision]
            if t_h >= t_m:
                roof = {"kernel": dom["name"], "bound": "hbm", "achieved": dom["alg_GB"] / t / 1e3, "peak": HBM_PEAK / 1e12,
                        "unit": "TB/s", "frac": (dom["alg_GB"] * 1e9 / t) / HBM_PEAK, "traffic": traffic_of(dom["name"])}
            else:
                roof = {"kernel": dom["name"], "bound": "mfma", "achieved": dom["alg_GFLOP"] / t / 1e3,
                        "peak": MFMA_PEAK[args.precision] / 1e12, "unit": "TFLOP/s",
                        "frac": (dom["alg_GFLOP"] * 1e9 / t) / MFMA_PEAK[args.precision], "traffic": traffic_of(dom["name"])}
        bytes_alg = 24.0 * C * V + 8.0 * (C * K + V * K)          # SURVEY 8(d), whole iteration, all GPUs
        flops_alg = 4.0 * C * V * K
        out = {
            "metric": "mapping iterations/s at 30k cells x 1k genes x 10k spots (mode='cells', lambda_g1=1, lambda_d=1)",
            "value": its, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {C} cells x {K} genes x {V} spots, planted-mapping synthetic "
                                   f"counts, Adam lr=0.1", "gemm_precision": args.precision,
                       "parallelism": "single GPU" if world == 1 else f"spots sharded over {world} GPUs, 3 small RCCL exchanges/step"},
            "cell_spot_gene_per_s": its * C * K * V,
            "last_main_loss": main_loss,
            "roofline": roof,
            "iteration_roofline": {"bytes_alg": bytes_alg, "flops_alg": flops_alg,
                                   "hbm_frac": bytes_alg * its / HBM_PEAK / world,
                                   "mfma_frac": flops_alg * its / MFMA_PEAK[args.precision] / world},
            "kernels": kern,
            "kernels_pass": {"schedule": "sequential (one stream, HIP event after every kernel)", "steps": nprof,
                             "ms_per_step": 1e3 * seq_elapsed / nprof, "value": nprof / seq_elapsed},
            "alt_precisions": alt,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(C, K, V)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
