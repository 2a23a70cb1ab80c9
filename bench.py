#!/usr/bin/env python
"""Benchmark of the Tangram mapping hot path on MI355X (BASELINE.json metric: mapping iterations/s and
cell*spot*gene/s at 30k cells x 1k genes x 10k spots; the other BASELINE configurations are --workload choices).

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg4|cfg5a|cfg5b] [--precision bf16x3|bf16|fp32]

One "step" = one full mapping iteration (softmax, P^T S, cosine + density loss, backward, Adam) on synthetic
inputs of the named shape that are resident in HBM before the timed region starts.

  cfg2   30 000 cells x 1 000 genes x 10 000 spots, mode='cells' defaults (lambda_g1 = lambda_d = 1)      -- the headline
  cfg4   200 000 x 2 000 x 50 000, bf16 MFMA operands, fp32 accumulate / state (fits ONE 288 GB GPU; sharded for N > 1)
  cfg5a  cfg2 shape, mode='constrained' (MapperConstrained: density + count + f_reg terms, target_count = V)
  cfg5b  cfg2 shape, mode='cells' + lambda_neighborhood_g1 = 0.96 + lambda_ct_islands = 0.17 on a 6-neighbour hex spot graph

N > 1: `python bench.py --gpus N` re-executes itself through torch.distributed.run (one rank per GPU, RCCL); the spots are
sharded over the ranks (strong scaling: the problem is fixed, `value` is whole-job iterations/s).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                      # B/s   (MI355X_MICROARCH.md: HBM3E 8 TB/s spec; 6.29 TB/s measured float4 copy)
HBM_COPY = 6.29e12
# dense matrix-core peak for ONE algorithmic product: bf16 2.5 PFLOP/s; split-bf16 x3 issues three bf16 MFMAs per product, i.e.
# 833 TFLOP/s effective (the figure SURVEY 8(d) prices the bf16x3 roofline with); exact f32 MFMA 157.3 TFLOP/s
MFMA_PEAK = {"bf16": 2.5e15, "bf16x3": 2.5e15 / 3.0, "fp32": 157.3e12}
DTYPE_NAME = {"bf16x3": "f32 (split-bf16 x3 MFMA operands, f32 accumulate and state: fp32-parity)",
              "bf16": "bf16 MFMA operands, f32 accumulate and state", "fp32": "f32 (exact f32 MFMA)"}
WORKLOADS = {
    # name: (C, K, V, mode, default GEMM precision, description)
    "cfg2": (30000, 1000, 10000, "cells", "bf16x3", "mode='cells', lambda_g1=1, lambda_d=1"),
    "cfg4": (200000, 2000, 50000, "cells", "bf16", "mode='cells', lambda_g1=1, lambda_d=1, bf16 operands / fp32 state"),
    "cfg5a": (30000, 1000, 10000, "constrained", "bf16x3", "mode='constrained', lambda_d=lambda_g1=lambda_count=lambda_f_reg=1, target_count=V"),
    "cfg5b": (30000, 1000, 10000, "spatial", "bf16x3", "mode='cells', lambda_g1=lambda_d=1, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17, "
                                                     "18 cell types, 6-neighbour hex spot graph (CSR)"),
}
HEAVY = ("tg_fwd_kernel", "tg_bwd_kernel", "tg_adam_rowpass", "tg_adam_update")


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


def bench_folds(device, folds=16, steps=300):
    """leave-one-gene-out folds of the tutorial's clusters-mode problem (18 clusters x 250 genes x 9 852 spots,
    utils.py:576-600) -- stepping rate of one fold alone and of `folds` folds advanced together (tg_batch)."""
    import torch
    import tangram_amd.mapping_optimizer as mo
    from tangram_amd.batched import MapperBatch
    from tangram_amd.synthetic import make_workload
    C, K, V = 18, 250, 9852
    w = make_workload(C, K + folds, V, device, seed=1)
    ds = torch.full((C,), 1.0 / C, device=device)

    def fold(i):
        keep = torch.tensor([g for g in range(K + folds) if g != i][:K], device=device)
        return mo.Mapper(S=w["S"][:, keep].contiguous(), G=w["G"][:, keep].contiguous(), d=w["d"], d_source=ds, lambda_d=1, device=device,
                         random_state=i + 1)

    def rate(step, n):
        step(50)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        step(steps)
        torch.cuda.synchronize(device)
        return n * steps / (time.perf_counter() - t0)

    m1 = fold(0)
    one = rate(lambda k: m1._engine.step(k, 0.1), 1)
    m1.release()
    ms = [fold(i) for i in range(folds)]
    b = MapperBatch(ms)
    many = rate(lambda k: b.step(k, 0.1), folds)
    b.close()
    for m in ms:
        m.release()
    return {"workload": f"{C} clusters x {K} genes x {V} spots, mode='clusters' folds of a leave-one-gene-out cross-validation",
            "one_fold_iters_per_s": one, "folds_per_launch": folds, "fold_iters_per_s": many, "unit": "mapping iterations/s summed over the folds"}


def csrc_sha():
    """Fingerprint of the kernel sources (tangram_amd/csrc/*): PMC tables are only valid for the kernels they were collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tangram_amd", "csrc")
    for name in sorted(n for n in os.listdir(d) if n.startswith("tg_") and n.endswith((".h", ".hip"))):     # every header of the split kernel sources
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


# scripts/probes/gemm_loop_lab.hip on MI355X (profiles/r04/lab): the backward GEMM's tile set with the matrix-core instructions
# ONLY (operands held in registers, no LDS read, no copy, no barrier).  With split-bf16 random operands the chip clocks to its
# power budget (~2.1 GHz shader clock under this load, s_memtime), so the matrix pipes themselves need this long:
MFMA_ONLY_PROBE = {"bf16x3": {"ms_per_gemm": 0.90, "frac_of_dense_peak": 0.84, "effective_clock_GHz": 2.1,
                              "source": "profiles/r04/lab/lab.txt (VAR 1), cfg2 backward tile set incl. padding"}}


# scripts/probes/cumask_step.py on MI355X (profiles/r05/run1_update_diet/cumask_step.txt): the kernels of the cfg2 iteration on CU-masked
# streams -- what a backward || update overlap by CU partition would have to live on (the round-4 review's gate: the update on ~80 CUs
# at >= 5.5 TB/s).  A CU moves <= ~37 GB/s of the update's stream whatever its instruction count (39.8 VALU per element since round 5).
OVERLAP_GATE = {"update_TBps_by_cus": {"256": 6.24, "192": 5.64, "160": 5.09, "128": 4.33, "96": 3.39, "64": 2.40},
                "bwd_gemm_ms_by_cus": {"256": 1.26, "192": 1.50, "160": 1.74, "128": 2.15, "96": 2.80},
                "gate": "update on ~80 CUs >= 5.5 TB/s", "met": False,
                "best_split": "128 / 128 CUs: 2.15 ms of backward GEMM beside 1.94 ms of update = 0.46 ms (12 %) saved in the ideal, before "
                              "the fill / drain of a cell-band pipeline",
                "source": "profiles/r05/run1_update_diet/cumask_step.txt"}


def pmc_traffic(precision):
    """(HBM bytes per launch, note) from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, plus WRITE_SIZE).  The table carries the fingerprint of the
    kernel sources it was collected on; against other sources it is stale and NOT attached ({} + a note saying so)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}, "profiles/pmc_traffic.json not found: no PMC traffic attached"
    have, want = tab.get("_csrc_sha"), csrc_sha()
    if have != want:
        return {}, f"profiles/pmc_traffic.json was collected on kernel sources {have}, this tree is {want}: traffic omitted (re-run scripts/gpu_pmc.sh + scripts/make_pmc_traffic.py)"
    return ({k: v["hbm_bytes_per_launch"] for k, v in tab.get(precision, {}).items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v},
            f"rocprofv3 --pmc passes on kernel sources {have} (profiles/pmc_traffic.json)")


def roof_of(bytes_alg, flops_alg, seconds, precision, traffic=None):
    """Roofline entry for `seconds` of work with the given algorithmic bytes / flops: the BINDING roof is the larger of
    t_HBM = bytes / 8 TB/s and t_MFMA = flops / dense peak of the precision; frac = that time / measured time."""
    t_h = bytes_alg / HBM_PEAK
    t_m = flops_alg / MFMA_PEAK[precision]
    if t_h >= t_m:
        return {"bound": "hbm", "achieved": bytes_alg / seconds / 1e12, "peak": HBM_PEAK / 1e12, "unit": "TB/s",
                "frac": t_h / seconds, "traffic": traffic}
    return {"bound": "mfma", "achieved": flops_alg / seconds / 1e12, "peak": MFMA_PEAK[precision] / 1e12, "unit": "TFLOP/s",
            "frac": t_m / seconds, "traffic": traffic}


# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload, C, K, V):
    """The reference's CPU path on the GPU box's host cores ("load the reference optimizer unmodified").

    `oracle/_ref/` (staged by oracle/make_ref.py in the authoring container; git-ignored, ships with the snapshot) holds the
    UNMODIFIED tangram/mapping_optimizer.py: its `Mapper(...).train(...)` (:358-408) is what is timed -- `"kind": "reference"`,
    at the workload's full shape when that fits the host (cfg2 / cfg5: 1 warm-up + 3 timed iterations), else on a stated
    fraction of the cell x spot plane.  Only when the staged copy is absent does the leg fall back to oracle/torch_port.py
    (a PyTorch-CPU port with the reference's op sequence) and says so: `"kind": "port"`.
    Thread count: swept on a small probe first (256 threads on element-wise ops across two sockets is oversubscription)."""
    import torch
    from tangram_amd.synthetic import make_workload, hex_grid_graph, cell_type_encoding
    from oracle import make_ref
    ncpu = os.cpu_count() or 1
    have_ref = make_ref.available()
    if have_ref:
        ref = make_ref.load()
        Mp, Mc = ref.Mapper, ref.MapperConstrained
    else:
        from oracle.torch_port import TorchPortMapper as Mp, TorchPortMapperConstrained as Mc

    def build(Cs, Vs):
        w = make_workload(Cs, K, Vs, "cpu", seed=0)
        S, G, d = w["S"].numpy(), w["G"].numpy(), w["d"].numpy()
        if workload == "cfg5a":
            return Mc(S=S, G=G, d=d, lambda_d=1, lambda_g1=1, lambda_g2=0, lambda_count=1, lambda_f_reg=1,
                      target_count=Vs, random_state=42)
        if workload == "cfg5b":           # the reference takes DENSE V x V weight matrices (mapping_optimizer.py:125-132)
            N, W = hex_grid_graph(Vs)
            E = cell_type_encoding(w["assign"].numpy(), Vs, 18)
            return Mp(S=S, G=G, d=d, lambda_g1=1, lambda_d=1, lambda_neighborhood_g1=0.96, voxel_weights=W.toarray(),
                      lambda_ct_islands=0.17, neighborhood_filter=N.toarray(), ct_encode=E, random_state=42)
        return Mp(S=S, G=G, d=d, lambda_g1=1, lambda_d=1, random_state=42)

    def time_iters(m, n):
        t0 = time.perf_counter()
        m.train(n, 0.1, print_each=None) if have_ref else m.train(n, 0.1)
        return (time.perf_counter() - t0) / n

    t_begin = time.perf_counter()
    # probe: 1/64 of the plane, one timed iteration per thread count
    Cp, Vp = max(C // 8, 64), max(V // 8, 64)
    probe = build(Cp, Vp)
    torch.set_num_threads(min(32, ncpu))
    time_iters(probe, 1)
    sweep = {}
    for nt in [n for n in (8, 16, 32, 64, 128, 256) if n <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        sweep[nt] = min(time_iters(probe, 1), time_iters(probe, 1))
    best = min(sweep, key=sweep.get)
    del probe
    torch.set_num_threads(best)
    # the reference holds ~9x the C x V plane in fp32 (autograd temporaries): full shape up to 3.2e8 elements
    # (cfg2: 10.9 GB resident), beyond that a fraction of the plane with all K genes
    div = 1
    while (C // div) * (V // div) > 3.2e8:
        div *= 2
    if not have_ref:
        div = max(div, 2)                                  # (the port leg keeps round 3's 1/4-plane sample)
    Cs, Vs = max(C // div, 64), max(V // div, 64)
    m = build(Cs, Vs)
    time_iters(m, 1)                                       # warm-up (cold first iteration)
    n = 3 if have_ref else 2
    dt = time_iters(m, n)
    csg = Cs * K * Vs / dt
    shape = f"{Cs}x{K}x{Vs}" + (" (the workload's full shape)" if div == 1 else f" = 1/{div * div} of the cell x spot plane")
    out = {"value": csg / (float(C) * K * V), "unit": "iters/s (cell*spot*gene/s of the sample / C*K*V of the workload)",
           "cell_spot_gene_per_s": csg, "cores": best, "host_cpus": ncpu,
           "thread_sweep_s_per_iter": {str(k): v for k, v in sweep.items()}}
    sweep_txt = f"{best} threads (best of a sweep on a {Cp}x{K}x{Vp} probe: " + ", ".join(f"{k}: {v:.3f} s" for k, v in sweep.items()) + ")"
    if have_ref:
        out.update(kind="reference",
                   sample=f"{n} iterations (after 1 warm-up) of the UNMODIFIED reference "
                          f"{'MapperConstrained' if workload == 'cfg5a' else 'Mapper'}.train ({'tangram/mapping_optimizer.py:587-639' if workload == 'cfg5a' else 'tangram/mapping_optimizer.py:358-408'}, staged "
                          f"in oracle/_ref by oracle/make_ref.py), device='cpu', fp32, at {shape}, {dt:.2f} s/iter (each train() call "
                          f"ends with the reference's final softmax + copy), {sweep_txt}")
    else:
        ratio = None
        try:
            ratio = json.load(open(os.path.join(ROOT, "oracle", "port_vs_reference.json")))["port_over_reference_time"]
        except Exception:
            pass
        out.update(kind="port",
                   sample=f"FALLBACK (oracle/_ref not staged): {n} iterations (after 1 warm-up) of oracle/torch_port.py (PyTorch-CPU port "
                          f"of the reference loop, fp32) at {shape}, {dt:.2f} s/iter, {sweep_txt}",
                   port_over_reference_time=ratio,
                   port_over_reference_note="wall time of this port / the UNMODIFIED reference Mapper on the same inputs and threads, "
                                            "measured once in the authoring container (oracle/measure_port_ratio.py)")
    out["wall_s"] = time.perf_counter() - t_begin
    return out


# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)         # >= 200 timed iterations after >= 20 warm-up
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default=None, choices=["bf16x3", "bf16", "fp32"], help="GEMM precision (default: the workload's)")
    ap.add_argument("--shape", default=None, help="override C,K,V (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short runs of the other GEMM precisions")
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0, help="force the GEMM tile edge (128 or 256); 0 = automatic")
    ap.add_argument("--bands", type=int, default=0, help="cell bands of the opt-in 3-stream pipeline (0/1 = sequential schedule, the default)")
    ap.add_argument("--s-exact", action="store_true", help="bf16x3 only: let the library use the two-product path when it finds S bf16-exact "
                                                            "(the synthetic counts are); the default line times the GENERAL three-product path")
    return ap.parse_args()


def respawn_distributed(args):
    """`python bench.py --gpus N` without a launcher: re-execute through torch.distributed.run, one rank per GPU."""
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    backend = os.environ.get("TG_BENCH_BACKEND", "nccl")
    if backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {ndev} "
              f"(RCCL cannot share a device between ranks)", file=sys.stderr)
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_distributed(args)
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (there is no CPU path)", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("TG_BENCH_BACKEND", "nccl")       # "gloo": plumbing smoke test of the N > 1 path on a 1-GPU box
    if world > 1 and backend == "nccl" and world > ndev:
        print(f"bench.py: {world} ranks need {world} GPUs (found {ndev}); RCCL cannot share a device between ranks", file=sys.stderr)
        sys.exit(2)
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
    from tangram_amd.synthetic import make_workload, init_logits, hex_grid_graph, cell_type_encoding

    C, K, V, mode, default_prec, wl_desc = WORKLOADS[args.workload]
    if args.shape:
        C, K, V = (int(x) for x in args.shape.split(","))
    precision = args.precision or default_prec
    w = make_workload(C, K, V, device, seed=0)
    lr = 0.1
    extra = {}
    if mode == "constrained":
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.0, lambda_count=1.0, lambda_f_reg=1.0)
        extra = dict(mode="constrained", target_count=float(V))
    elif mode == "spatial":
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17)   # values of README.md:96-100
        N, W = hex_grid_graph(V)
        extra = dict(voxel_weights=W, neighborhood_filter=N, ct_encode=cell_type_encoding(w["assign"].cpu().numpy(), V, 18))
    else:
        lam = dict(lambda_g1=1.0, lambda_d=1.0)        # mode='cells' defaults as resolved by mapping_utils.py:214-215

    def make_engine(prec, s_exact=None):
        s_exact = ("auto" if args.s_exact else False) if s_exact is None else s_exact
        if world == 1:
            M0 = init_logits(C, V, device, seed=42)
            if mode == "constrained":
                extra["F0"] = torch.randn(C, device=device, generator=torch.Generator(device=device).manual_seed(7))
            e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=device, precision=prec, lambdas=lam,
                                fwd_splits=args.splits, tile_size=args.tile, pipeline_bands=args.bands, s_exact=s_exact, **extra)
            return e, e, (lambda n, h=None: e.step(n, lr, h))
        lo, hi = shard_bounds(V, world, rank, mode == "spatial")      # (spatial terms: blocks of ceil(V / world) spots)
        M0 = init_logits(C, hi - lo, device, seed=42 + rank)
        kw = dict(extra)
        if mode == "constrained":
            kw["F0"] = torch.randn(C, device=device, generator=torch.Generator(device=device).manual_seed(7))
        sh = ShardedMapperEngine(w["S"], w["G"][lo:hi].contiguous(), M0, w["d"][lo:hi].contiguous(), n_spots_total=V,
                                 device=device, precision=prec, lambdas=lam, fwd_splits=args.splits, tile_size=args.tile,
                                 spot_offset=lo, s_exact=s_exact, **kw)
        return sh, sh.eng, (lambda n, h=None: sh.run(n, lr, h))

    owner, core, run = make_engine(precision)
    eff_prec = getattr(core, "effective_precision", precision)
    torch.cuda.empty_cache()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(args.warmup)
    timed_hist = core.new_history(args.steps)       # the timed region is literally Mapper.train's call: it writes the history rows
    fence()
    t0 = time.perf_counter()
    run(args.steps, timed_hist)                     # timed region: the product schedule
    fence()
    elapsed = time.perf_counter() - t0
    # A LONGER run of the same schedule, once, after the timed region (never `value`): the device has two HBM states tens of seconds long
    # (the update kernel at 5.7 or 6.3 TB/s); 20 timed steps catch one of them, 200 average more of it -- a reader
    # of the line can tell a slow-state draw from a regression.
    long_steps = 200 if elapsed / args.steps < 0.010 else 20
    fence()
    tl = time.perf_counter()
    run(long_steps)
    fence()
    long_elapsed = time.perf_counter() - tl
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
        tt = torch.tensor([elapsed, long_elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, long_elapsed = float(tt[0].item()), float(tt[1].item())

    # sanity: the loss of the last step is finite (nothing was skipped)
    hist = core.new_history(1)
    run(1, hist)
    torch.cuda.synchronize(device)
    main_loss = float(hist[0, 1].item())
    if not (main_loss == main_loss) and not os.environ.get("TG_BENCH_TIMING_ABLATION"):    # (ablation builds compute garbage on purpose)
        print("bench.py: the last step's main_loss is NaN", file=sys.stderr)
        sys.exit(3)

    # the other GEMM precisions on the same inputs (reported beside the headline, never as `value`)
    alt = {}
    if world == 1 and not args.no_alt and args.workload != "cfg4":
        owner.release() if hasattr(owner, "release") else None
        del owner, core, run
        torch.cuda.empty_cache()
        alts = [(p, None) for p in ("bf16x3", "bf16", "fp32") if p != precision]
        if precision == "bf16x3" and not args.s_exact:
            alts.insert(0, ("bf16x3_s_exact", "auto"))       # the opt-in two-product path (S of the synthetic workload is bf16-exact counts)
        for prec_name, se in alts:
            prec = "bf16x3" if prec_name == "bf16x3_s_exact" else prec_name
            e2, _, run2 = make_engine(prec, se)
            n2 = max(4, min(args.steps // 4, 50))
            run2(3)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            run2(n2)
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t1) / n2
            alt[prec_name] = {"value": 1.0 / dt, "unit": "iters/s", "ms_per_step": 1e3 * dt, "steps": n2, "dtype": DTYPE_NAME[prec],
                         "effective_precision": getattr(e2, "effective_precision", None),
                         "roofline": dict(roof_of(24.0 * C * V + 8.0 * (C * K + V * K), 4.0 * C * V * K, dt, prec),
                                          scope="one iteration", hbm_frac=(24.0 * C * V + 8.0 * (C * K + V * K)) / dt / HBM_PEAK,
                                          mfma_frac=4.0 * C * V * K / dt / MFMA_PEAK[prec])}
            e2.release()
            del e2, run2
            torch.cuda.empty_cache()
    del w
    next_rows = {}
    if world == 1 and not args.no_alt and args.workload == "cfg2" and not args.shape:
        try:        # SURVEY 8(f-3) beside the headline (never `value`): clusters-mode cross-validation folds, one alone and 16 per tg_batch
            next_rows["f3_batched_mappings"] = bench_folds(device)
        except Exception as e:
            next_rows["f3_batched_mappings"] = {"error": repr(e)}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        its = args.steps / elapsed
        Vl = V if world == 1 else (shard_bounds(V, world, 0)[1])
        # the committed PMC table was collected on the full single-GPU cfg2 launch: it does not describe a shard or another shape
        pmc, pmc_note = pmc_traffic(precision) if (world == 1 and not args.shape and args.workload == "cfg2") else ({}, "no PMC table for this workload / shard")
        kern = []
        for name, ms, cnt in prof:
            b, f = kernel_model(name, C, K, Vl)
            row = {"name": name, "avg_ms": ms / max(cnt, 1), "launches": cnt,
                   "alg_GB": None if b is None else b / 1e9, "alg_GFLOP": None if f is None else f / 1e9}
            if b is not None and cnt:
                row["roofline"] = roof_of(b, f, 1e-3 * ms / cnt, precision, pmc.get(name))
            kern.append(row)
        # The roofline of the ITERATION: bytes_alg = 24 C V + 8 (C K + V K), flops_alg = 4 C V K over all GPUs; the
        # binding roof is the larger of t_HBM and t_MFMA, frac = that time / measured time per step.  The heavy kernels follow
        # with their own figures (kernel durations: HIP events on the kernel's stream, averaged over the profiled steps).
        bytes_alg = 24.0 * C * V + 8.0 * (C * K + V * K)
        flops_alg = 4.0 * C * V * K
        it_traffic = sum(pmc.get(k["name"], 0.0) for k in kern if k["name"] in HEAVY) if pmc else None
        roof = roof_of(bytes_alg / world, flops_alg / world, elapsed / args.steps, precision, it_traffic or None)
        roof["scope"] = "one iteration (all kernels of a step, per GPU)"
        roof["traffic_source"] = pmc_note
        roof["bytes_alg"] = bytes_alg
        roof["flops_alg"] = flops_alg
        roof["hbm_frac"] = bytes_alg * its / HBM_PEAK / world
        roof["hbm_frac_of_measured_copy_bw"] = bytes_alg * its / HBM_COPY / world
        roof["mfma_frac"] = flops_alg * its / MFMA_PEAK[precision] / world
        roof["kernels"] = [dict(name=k["name"], avg_ms=k["avg_ms"], **k["roofline"]) for k in kern if "roofline" in k]
        # what the HBM-bound update kernel really moved per second in THIS run: counter bytes (PMC table) -- or, without a table for this
        # workload, the kernel's modelled 28 B per element (it also reads X) -- over its measured duration here
        upd = next((k for k in kern if k["name"] in ("tg_adam_rowpass", "tg_adam_update") and k["launches"]), None)
        if upd is not None:
            ub = pmc.get(upd["name"])
            roof["update_TBps_actual"] = (ub if ub else 28.0 * C * Vl) / (1e-3 * upd["avg_ms"]) / 1e12
            roof["update_traffic_source"] = ("PMC bytes per launch (profiles/pmc_traffic.json) / this run's kernel time" if ub else
                                             "modelled 28 B per element (reads X, M, m, v; writes M, m, v) / this run's kernel time")
        # The structural ceiling of this design, on the face of the line.  The step is a CHAIN of two
        # matrix-core-bound GEMMs and one HBM-bound update (softmax backward needs the complete row dot before any element of
        # the row may be updated: DESIGN.md section 4), so its floor is the SUM of the three kernels' own roofs, not their
        # maximum; the north star's 0.60-of-HBM target prices the iteration against the maximum.
        heavy = {k["name"]: k for k in kern if k["name"] in HEAVY and k.get("launches")}
        t_seq = sum(k["avg_ms"] for k in heavy.values())
        t_gemm_roof = 1e3 * (flops_alg / world) / MFMA_PEAK[precision]
        t_upd_roof = 1e3 * (24.0 * C * V / world) / HBM_PEAK
        t_bound = 1e3 * max((bytes_alg / world) / HBM_PEAK, (flops_alg / world) / MFMA_PEAK[precision])
        roof["floor"] = {
            "sequential_measured_ms": t_seq,
            "sequential_measured_parts": {n: k["avg_ms"] for n, k in heavy.items()},
            "sequential_roof_ms": t_gemm_roof + t_upd_roof,
            "sequential_roof_its": 1e3 / (t_gemm_roof + t_upd_roof),
            "overlap_bound_ms": t_bound, "overlap_bound_its": 1e3 / t_bound,
            "mfma_only_probe": MFMA_ONLY_PROBE.get(precision),
            "note": "sequential_roof = (two GEMMs at the dense MFMA peak of the precision) + (update at 8 TB/s): what the three-kernel "
                    "chain could reach with every kernel AT its roof; overlap_bound = max(t_MFMA, t_HBM) of the whole iteration, "
                    "reachable only if the phases overlapped (every overlap design measured slower: profiles/LABBOOK.md section 6b; "
                    "round 5's gate on reviving them: overlap_gate below, DESIGN.md section 6)",
            "overlap_gate": OVERLAP_GATE if (args.workload == "cfg2" and not args.shape) else None}
        target_its = 0.60 * HBM_PEAK / (bytes_alg / world)
        roof["target"] = {"definition": "north_star: >= 0.60 of the HBM roofline (8 TB/s) on the fused iteration",
                          "its": target_its, "ms_per_step": 1e3 / target_its, "met": bool(its >= target_its),
                          "frac_of_target": its / target_its,
                          "reachable_by_this_design": bool(1e3 / (t_gemm_roof + t_upd_roof) >= target_its)}
        metric = ("mapping iterations/s at 30k cells x 1k genes x 10k spots (mode='cells', lambda_g1=1, lambda_d=1)"
                  if args.workload == "cfg2" and not args.shape else
                  f"mapping iterations/s at {C} cells x {K} genes x {V} spots ({wl_desc})")
        out = {
            "metric": metric,
            "value": its, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "value_long": {"value": long_steps / long_elapsed, "unit": "iters/s", "steps": long_steps, "ms_per_step": 1e3 * long_elapsed / long_steps,
                           "note": "one longer run of the same schedule after the timed region; never `value`"},
            "dtype": DTYPE_NAME[precision], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {C} cells x {K} genes x {V} spots, {wl_desc}; planted-mapping synthetic "
                                   f"counts, Adam lr=0.1", "gemm_precision": precision,
                       "effective_precision": eff_prec,
                       "parallelism": "single GPU" if world == 1 else
                       f"spots sharded over {world} ranks, 3 small exchanges/step issued by the library ({getattr(owner, 'transport', '?')}: "
                       f"{ {'rccl': 'RCCL on the compute stream', 'peer': 'one-hop peer-memory exchange kernels on the compute stream'}.get(getattr(owner, 'transport', ''), backend + ' through callbacks') })"},
            "cell_spot_gene_per_s": its * C * K * V,
            "last_main_loss": main_loss,
            "roofline": roof,
            "kernels": kern,
            "kernels_pass": {"schedule": "sequential (one stream, HIP event after every kernel)", "steps": nprof,
                             "ms_per_step": 1e3 * seq_elapsed / nprof, "value": nprof / seq_elapsed},
            "alt_precisions": alt,
            "next_rows": next_rows,
        }
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "cfg4":
                out["cpu_baseline"] = {"value": None, "kind": "port", "cores": 0, "unit": "iters/s",
                                       "sample": "not runnable on the CPU reference: M alone is 40 GB fp32 plus ~9x that in autograd "
                                                 "temporaries (BASELINE.md section 3, item 5); see the cfg2 line for the CPU rate per cell*spot*gene"}
            else:
                try:
                    out["cpu_baseline"] = cpu_baseline(args.workload, C, K, V)
                except Exception as e:  # the baseline must never take the GPU number down with it
                    out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
