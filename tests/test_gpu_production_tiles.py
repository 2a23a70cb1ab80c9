"""GPU parity at PRODUCTION tile counts (-m gpu), closing the size hole of the golden fixtures (<= 300 x 60 x 120):

 (i)  4 200 x 1 000 x 1 500 against the fp64 NumPy oracle: K = 1 000 => 4 gene tiles of 256 and 32 contraction steps in the
      backward GEMM, 17 cell tiles x 6 spot tiles, several forward splits -- both tile geometries, all three GEMM precisions,
      5 epochs of trajectory + the first-step gradient (recovered from Adam's first moment);
 (ii) the full BASELINE shape 30 000 x 1 000 x 10 000 against the REFERENCE'S OWN OP SEQUENCE (oracle/torch_port.py: softmax,
      matmul, cosine_similarity, KLDivLoss, autograd, torch.optim.Adam -- what the reference executes with device='cuda')
      run by PyTorch-ROCm in fp32 on the same GPU: first-step gradient and a 3-step loss trajectory, cells and constrained.

The checker is the oracle; the thing under test is the C-ABI library.  Tolerances: the stated fp32 tolerances of
tests/parity_common.py (loss 1e-5, gradient rel 1e-5 vs fp64 / 1e-4 vs the fp32 torch run, whose own round-off is ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from tests import parity_common as pc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BETA1 = 0.9

C1, K1, V1, N1 = 4200, 1000, 1500, 5


@pytest.fixture(scope="module")
def oracle_k1000():
    """fp64 oracle run shared by every parametrisation below (about 10 s of host time)."""
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(C1, K1, V1, seed=21)
    M0 = orc.reference_init_M(C1, V1, 5)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    _, dM = o.loss_and_grad()
    Po, ho = o.train(N1, 0.1)
    Gh = Po.T @ data["S"].astype(np.float64)
    return dict(data=data, M0=M0, lam=lam, dM=dM, P=Po, hist=ho, Ghat=Gh)


@pytest.mark.parametrize("tile", [256, 128])
@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16"])
def test_k1000_multi_tile_against_oracle_fp64(oracle_k1000, precision, tile):
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    o = oracle_k1000
    data = o["data"]
    e = HipMapperEngine(data["S"], data["G"], o["M0"], d=data["d"], device=DEV, precision=precision, lambdas=o["lam"],
                        tile_size=tile)
    hist = e.new_history(N1)
    e.step(1, 0.1, hist, 0)
    _, m1, _, _ = e.logits()
    g = (m1[:, :V1] / (1.0 - BETA1)).cpu().numpy().astype(np.float64)        # exp_avg after one step = (1 - beta1) * grad
    rel = np.linalg.norm(g - o["dM"]) / np.linalg.norm(o["dM"])
    assert rel <= (1e-5 if precision != "bf16" else 1e-2), f"first-step gradient rel err {rel:.3e}"
    e.step(N1 - 1, 0.1, hist, 1)
    h = hist.cpu().numpy().astype(np.float64)
    tol = pc.TOL[precision]
    for k, col in (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("vg_reg", _capi.H_VG), ("kl_reg", _capi.H_KL)):
        err = float(np.abs(h[:, col] - np.asarray(o["hist"][k], dtype=np.float64)).max())
        assert err <= tol["loss"], f"{k}: max per-epoch |delta| {err:.3e}"
    P = e.result().cpu().numpy()
    assert float(np.abs(P - o["P"]).max()) <= tol["P"]
    Gh = e.project().cpu().numpy()
    relg = np.linalg.norm(Gh - o["Ghat"]) / np.linalg.norm(o["Ghat"])
    assert relg <= tol["ghat"], f"relFro(P^T S) {relg:.3e}"
    e.release()


EDGE_SHAPES = [
    # C, K, V, expected tile edge, expected wide forward (bf16x3)   -- boundaries of tg_make_layout's rules
    (4096, 1, 448, 256, 0),        # smallest problem on the 256 layout; a single gene (+ the density column)
    (4097, 511, 449, 256, 1),      # one past every multiple: 17 cell tiles, 2 spot tiles, 512 padded gene columns -> wide forward
    (4100, 512, 1025, 256, 0),     # 513 columns pad to 768: not a multiple of 512 -> the 256^2 forward
    (4095, 40, 3000, 128, 0),      # one cell short of the 256 layout
    (5000, 30, 600, 128, 0),       # spots would pad 640 -> 768 (+20 %): stays on 128
    (9000, 70, 257, 128, 0),       # V < 448
]


@pytest.mark.parametrize("mode", ["mapper", "constrained"])
@pytest.mark.parametrize("C,K,V,tile,wide", EDGE_SHAPES)
def test_layout_boundaries_against_oracle_fp64(C, K, V, tile, wide, mode):
    """Ragged shapes at the boundaries of the geometry rules (layout 128 / 256, wide forward, dense tile map with bands of unequal
    height, a single gene) against the fp64 oracle: 3 epochs of history, the mapping, the filter and the projection."""
    import ctypes as ct
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    from oracle import tangram_oracle as orc
    n = 3
    data = orc.make_synthetic(C, K, V, seed=C + K + V)
    if mode == "constrained":
        M0, F0 = orc.reference_init_MF_constrained(C, V, 11)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.4, lambda_r=1e-4, lambda_count=0.9, lambda_f_reg=1.1)
        kw = dict(F0=F0, mode="constrained", target_count=0.4 * V)
        o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, target_count=0.4 * V, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
    else:
        M0 = orc.reference_init_M(C, V, 11)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.4, lambda_l2=1e-7)
        kw, Fo = {}, None
        o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam, **kw)
    geo = (ct.c_int * 8)()
    assert e._lib.tg_debug_layout(ct.byref(e.cfg), geo) == 0
    assert (geo[0], geo[5]) == (tile, wide)
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    h = hist.cpu().numpy().astype(np.float64)
    for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss"), (_capi.H_VG, "vg_reg"), (_capi.H_KL, "kl_reg")):
        ref = np.array([float(x) for x in ho[k]])
        err = np.abs(h[:, col] - ref).max()
        assert err <= 1e-5 * max(1.0, np.abs(ref).max()), (k, err)
    if mode == "constrained":
        P, F = e.result(with_filter=True)
        assert np.abs(F.cpu().numpy() - Fo).max() <= 2e-5
    else:
        P = e.result()
    assert np.abs(P.cpu().numpy() - Po).max() <= 2e-4
    Gh = e.project().cpu().numpy()
    want = (Po * Fo[:, None]).T @ data["S"].astype(np.float64) if mode == "constrained" else Po.T @ data["S"].astype(np.float64)
    assert np.linalg.norm(Gh - want) / np.linalg.norm(want) <= 1e-4
    e.release()


def test_backward_tile_choice_does_not_change_the_result():
    """Under the 256 layout the backward GEMM runs on 256^2 or 128^2 tiles (a fixed rule of the shape; `bwd_tile` pins it): every
    X element is the same k-ordered sum, so mapping and history are bit-identical; and a 1-rank spot shard (row-dot epilogue) on
    either geometry matches the fp64 oracle."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from tangram_amd import _capi
    from tests.local_comm import run_ranks
    from oracle import tangram_oracle as orc
    C, K, V, n = 6100, 48, 2900, 3
    data = orc.make_synthetic(C, K, V, seed=33)
    M0 = orc.reference_init_M(C, V, 3)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-4)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)

    def alone(bwd_tile=0):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam, bwd_tile=bwd_tile)
        assert e.cfg.tile_size == 0
        hist = e.new_history(n)
        e.step(n, 0.1, hist)
        out = hist.cpu().numpy(), e.result().cpu().numpy()
        e.release()
        return out

    def check(hist, P, what):
        for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss"), (_capi.H_KL, "kl_reg"), (_capi.H_ENTROPY, "entropy_reg")):
            ref = np.array([float(x) for x in ho[k]])
            err = np.abs(hist[:, col].astype(np.float64) - ref).max()
            assert err <= 1e-5 * max(1.0, np.abs(ref).max()), (what, k, err)
        assert np.abs(P - Po).max() <= 2e-4, what

    h_auto, P_auto = alone()
    check(h_auto, P_auto, "auto")
    for pin in (256, 128):
        h, P = alone(pin)
        np.testing.assert_array_equal(P, P_auto)
        np.testing.assert_array_equal(h, h_auto)

        def rank_fn(comm):
            sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam, comm=comm,
                              bwd_tile=pin)
            hh = sh.eng.new_history(n)
            sh.run(n, 0.1, hh)
            out = hh.cpu().numpy(), sh.result_full().cpu().numpy()
            sh.release()
            return out

        (h1, P1), = run_ranks(1, rank_fn)
        check(h1, P1, f"1-rank shard, backward tiles {pin}")


def _torch_reference(w, mode, M0, F0, steps):
    """The reference's op sequence in fp32 on the GPU (oracle/torch_port.py with device tensors): per-step history, the
    first-step gradient of M (and F)."""
    from oracle.torch_port import TorchPortMapper, TorchPortMapperConstrained
    tiny = lambda x: x[:4].cpu().numpy()
    if mode == "constrained":
        V = w["G"].shape[0]
        m = TorchPortMapperConstrained(tiny(w["S"]), tiny(w["G"]), tiny(w["d"]), lambda_d=1, lambda_g1=1, lambda_g2=0, lambda_count=1,
                                       lambda_f_reg=1, target_count=V, M0=np.zeros((4, 4)), F0=np.zeros(4))
        m.F = F0.clone().requires_grad_(True)
        params = None
    else:
        m = TorchPortMapper(tiny(w["S"]), tiny(w["G"]), d=tiny(w["d"]), lambda_g1=1, lambda_d=1, M0=np.zeros((4, 4)))
    m.S, m.G, m.d = w["S"], w["G"], w["d"]                          # full-size tensors, already on the GPU
    m.M = M0.clone().requires_grad_(True)
    params = [m.M, m.F] if mode == "constrained" else [m.M]
    opt = torch.optim.Adam(params, lr=0.1)
    hist, grad = [], None
    for i in range(steps):
        total, terms = m.loss()
        opt.zero_grad()
        total.backward()
        if i == 0:
            grad = m.M.grad.detach().clone()
        opt.step()
        hist.append(terms)
    del opt
    return hist, grad


@pytest.mark.parametrize("mode", ["cells", "constrained"])
def test_full_size_cfg2_against_reference_op_sequence(mode):
    """30k x 1k x 10k (118 x 40 tiles of 256^2, 4 gene tiles, 3 forward splits): gradient and 3-step trajectory of the HIP
    path (bf16x3, the default) vs the reference's op sequence executed by PyTorch-ROCm fp32 on the same GPU."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    from tangram_amd import _capi
    C, K, V = 30000, 1000, 10000
    w = make_workload(C, K, V, DEV, seed=0)
    M0 = init_logits(C, V, DEV, seed=42)
    F0 = torch.randn(C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7)) if mode == "constrained" else None
    n = 3
    ref_hist, ref_grad = _torch_reference(w, mode, M0, F0, n)
    torch.cuda.empty_cache()
    if mode == "constrained":
        e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], F0=F0, mode="constrained", device=DEV, precision="bf16x3",
                            lambdas=dict(lambda_g1=1, lambda_d=1, lambda_g2=0, lambda_count=1, lambda_f_reg=1), target_count=float(V))
    else:
        e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    del M0
    hist = e.new_history(n)
    e.step(1, 0.1, hist, 0)
    _, m1, _, _ = e.logits()
    g = m1[:, :V] / (1.0 - BETA1)
    # The fp32 autograd gradient itself carries ~1e-6 of round-off; entries of dM span 10 orders of magnitude (P underflows), so
    # the comparison is in the Frobenius norm, plus a per-row check that no row is off (a wrong tile would be a whole block).
    rel = float(torch.linalg.norm(g - ref_grad) / torch.linalg.norm(ref_grad))
    assert rel <= 1e-4, f"first-step gradient rel err {rel:.3e}"
    row_rel = torch.linalg.norm(g - ref_grad, dim=1) / torch.linalg.norm(ref_grad, dim=1).clamp_min(1e-30)
    assert float(row_rel.max()) <= 1e-3, f"worst row {int(row_rel.argmax())}: {float(row_rel.max()):.3e}"
    col_rel = torch.linalg.norm(g - ref_grad, dim=0) / torch.linalg.norm(ref_grad, dim=0).clamp_min(1e-30)
    assert float(col_rel.max()) <= 1e-3, f"worst spot column {int(col_rel.argmax())}: {float(col_rel.max()):.3e}"
    del g, ref_grad, row_rel, col_rel
    e.step(n - 1, 0.1, hist, 1)
    h = hist.cpu().numpy().astype(np.float64)
    for k, col in (("main_loss", _capi.H_MAIN), ("kl_reg", _capi.H_KL)):
        ref = np.array([float(t[k]) for t in ref_hist])
        err = float(np.abs(h[:, col] - ref).max())
        assert err <= 1e-5, f"{k}: {h[:, col]} vs {ref}"
    ref_total = np.array([float(t["total_loss"]) for t in ref_hist])
    scale = max(1.0, float(np.abs(ref_total).max()))        # constrained: |sum f - target| ~ 1e4 dominates the total
    assert float(np.abs(h[:, _capi.H_TOTAL] - ref_total).max()) <= 2e-5 * scale, (h[:, _capi.H_TOTAL], ref_total)
    if mode == "constrained":
        for k, col in (("count_reg", _capi.H_COUNT), ("lambda_f_reg", _capi.H_FREG)):
            ref = np.array([float(t[k]) for t in ref_hist])
            assert float(np.abs(h[:, col] - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), (k, h[:, col], ref)
    e.release()


def test_rccl_one_rank_group_runs_the_sharded_step():
    """RCCL itself on a real device: a 1-rank `nccl` process group drives the spot-sharded step (all three exchanges are
    issued through RCCL) and must reproduce the single-engine run on the same inputs."""
    import os
    import torch.distributed as dist
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import ShardedMapperEngine
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        C, K, V = 900, 150, 520
        data = orc.make_synthetic(C, K, V, seed=2)
        M0 = orc.reference_init_M(C, V, 8)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.3, lambda_r=1e-3, lambda_l2=1e-6)
        n = 6
        sh = ShardedMapperEngine(data["S"], data["G"], M0, data["d"], n_spots_total=V, device=dev, precision="bf16x3", lambdas=lam)
        assert sh.transport == "rccl"          # the C library bound librccl.so itself and issues the collectives on its stream
        hs = sh.eng.new_history(n)
        sh.run(n, 0.1, hs)
        hs = hs.cpu().numpy()
        Ps = sh.result_full().cpu().numpy()
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=dev, precision="bf16x3", lambdas=lam)
        h1 = e.new_history(n)
        e.step(n, 0.1, h1)
        h1 = h1.cpu().numpy()
        for col in (0, 1, 2, 3, 4, 6):
            # (sums over cells / spots are associated differently on the two paths: fp32 round-off relative to the term's size)
            np.testing.assert_allclose(hs[:, col], h1[:, col], rtol=5e-7, atol=2e-6, err_msg=f"history column {col}")
        assert float(np.abs(Ps - e.result().cpu().numpy()).max()) <= 1e-5     # (summation orders differ: 2e-6 observed)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile", [256, 128])
def test_two_product_path_equals_the_general_path_at_production_tiles(oracle_k1000, tile):
    """`s_exact="auto"` (PrecBF16x2S: S is bf16-exact count data, the product a_hi * S_lo adds exact zeros and is skipped) at K = 1 000:
    4 gene tiles, 32 backward contraction steps, wide forward tiles, several splits, compact 64-byte S tile rows in LDS.  EQUAL to
    the three-product path (history, mapping, first Adam moment), and within the fp32 tolerances of the fp64 oracle."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    o = oracle_k1000
    data = o["data"]
    outs = []
    for se in (False, "auto"):
        e = HipMapperEngine(data["S"], data["G"], o["M0"], d=data["d"], device=DEV, precision="bf16x3", lambdas=o["lam"], tile_size=tile, s_exact=se)
        hist = e.new_history(N1)
        e.step(N1, 0.1, hist)
        outs.append(dict(h=hist.cpu().numpy(), P=e.result().cpu().numpy(), m1=e.logits()[1].cpu().numpy(), eff=e.effective_precision))
        e.release()
    assert outs[0]["eff"] == "bf16x3" and outs[1]["eff"].startswith("bf16x3 (S exact")
    for k in ("h", "P", "m1"):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    h = outs[1]["h"].astype(np.float64)
    for k, col in (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("vg_reg", _capi.H_VG), ("kl_reg", _capi.H_KL)):
        assert float(np.abs(h[:, col] - np.asarray(o["hist"][k], dtype=np.float64)).max()) <= pc.TOL["bf16x3"]["loss"], k
    assert float(np.abs(outs[1]["P"] - o["P"]).max()) <= pc.TOL["bf16x3"]["P"]


def test_two_product_path_full_size_cfg2_first_steps_equal():
    """The BASELINE shape 30 000 x 1 000 x 10 000 (synthetic counts: bf16-exact S): three steps of the two-product path equal the
    general path's (history bits; mapping bits on a sample of rows)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    C, K, V = 30000, 1000, 10000
    w = make_workload(C, K, V, DEV, seed=0)
    M0 = init_logits(C, V, DEV, seed=42)
    res = []
    for se in (False, "auto"):
        e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_g1=1.0, lambda_d=1.0), s_exact=se)
        hist = e.new_history(3)
        e.step(3, 0.1, hist)
        P = e.result()
        res.append((hist.cpu().numpy(), P[::997].cpu().numpy(), e.effective_precision))
        e.release()
        del e, P
        torch.cuda.empty_cache()
    assert res[1][2].startswith("bf16x3 (S exact")
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True)
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("seed", range(int(os.environ.get("TG_FUZZ_SEEDS", "24"))))      # (TG_FUZZ_SEEDS=200: a wider one-off sweep)
def test_random_production_geometries_against_oracle_fp64(seed):
    """Seeded random problems LARGE enough for the production geometries -- 256^2 tiles, the 128 x 512 forward (padded gene counts
    that are multiples of 512), several gene tiles, stream-K and forced piece counts of the forward, both backward tile sizes, the
    two-product path on bf16-exact S -- with ragged cell / gene / spot counts, both mapper classes, 3 epochs against the fp64
    oracle (the small random cases of tests/test_gpu_parity.py never leave the 128^2 geometry)."""
    import torch
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    from tangram_amd.engine import HipMapperEngine
    from tests import parity_common as pc
    rng = np.random.default_rng(7000 + seed)
    C = int(rng.integers(300, 5000))
    K = int(rng.choice([int(rng.integers(1, 1600)), int(rng.integers(769, 1023)), int(rng.integers(1281, 1535)), 511, 1023]))
    V = int(rng.integers(200, 3000))
    prec = ["bf16x3", "bf16x3", "fp32", "bf16"][int(rng.integers(4))]
    constrained = bool(rng.integers(3) == 0)
    tile = int(rng.choice([0, 128, 256, 256]))
    forced = int(rng.integers(3))
    fwd_splits = 0 if forced == 0 else (int(rng.integers(1, 6)) if forced == 1 else -int(rng.integers(2, 300)))
    bwd_tile = int(rng.choice([0, 0, 128, 256])) if tile != 128 else 0
    s_exact = "auto" if (prec == "bf16x3" and rng.integers(2)) else False
    data = orc.make_synthetic(C, K, V, seed=seed)
    n = 3
    pick = lambda vals: float(rng.choice(vals))
    lam = dict(lambda_g1=1.0, lambda_d=pick([0.5, 1.0]), lambda_g2=pick([0.0, 0.5]), lambda_r=pick([0.0, 1e-3]))
    kw = dict(device="cuda:0", precision=prec, tile_size=tile, fwd_splits=fwd_splits, bwd_tile=bwd_tile, s_exact=s_exact)
    if constrained:
        lam.update(lambda_count=1.0, lambda_f_reg=1.0)
        tc = float(max(1, C // 3))
        M0, F0 = orc.reference_init_MF_constrained(C, V, seed)
        o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", lambdas=lam, target_count=tc, **kw)
    else:
        lam.update(lambda_l2=pick([0.0, 1e-5]))
        M0 = orc.reference_init_M(C, V, seed)
        o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], lambdas=lam, **kw)
    what = (seed, C, K, V, prec, constrained, tile, fwd_splits, bwd_tile, s_exact, e.effective_precision)
    h = e.new_history(n)
    e.step(n, 0.1, h)
    torch.cuda.synchronize()
    hh = h.cpu().numpy().astype(np.float64)
    tol = pc.TOL[prec]
    for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss")):
        ref = np.array([float(x) for x in ho[k]])
        assert np.abs(hh[:, col] - ref).max() <= 3 * tol["loss"] * max(1.0, np.abs(ref).max()), (what, k)
    out = e.result(with_filter=constrained)
    P = (out[0] if constrained else out).cpu().numpy()
    assert np.abs(P - Po).max() <= tol["P"], what
    rel = np.linalg.norm(P - Po) / np.linalg.norm(Po)
    assert rel <= (1e-4 if prec != "bf16" else 2e-2), (what, rel)
    if constrained:
        assert np.abs(out[1].cpu().numpy() - Fo).max() <= (2e-5 if prec != "bf16" else 5e-3), what
    Gh = e.project().cpu().numpy().astype(np.float64)
    S_eff = data["S"].astype(np.float64) * (Fo[:, None] if constrained else 1.0)
    ref = Po.T @ S_eff
    assert np.linalg.norm(Gh - ref) / np.linalg.norm(ref) <= tol["ghat"], what


@pytest.mark.parametrize("seed", range(int(os.environ.get("TG_FUZZ_SHARD_SEEDS", "10"))))
def test_random_spot_shards_against_oracle_fp64(seed):
    """The same kind of draw for the SHARDED step: 2 - 5 spot shards (threads of this process on one GPU, tests/local_comm.py) of a
    random problem in the production geometries -- uneven shard widths, the row-dot backward on either tile size, both mapper
    classes -- 3 epochs against the fp64 oracle of the unsharded problem; the global history bit-identical on every rank."""
    from oracle import tangram_oracle as orc
    from tangram_amd import _capi
    from tangram_amd.sharded import make_sharded
    from tests.local_comm import run_ranks
    rng = np.random.default_rng(9000 + seed)
    world = int(rng.integers(2, 6))
    C = int(rng.integers(300, 4000))
    K = int(rng.choice([int(rng.integers(1, 1300)), int(rng.integers(769, 1023)), 1000]))
    V = int(rng.integers(world * 40, 2600))
    prec = ["bf16x3", "bf16x3", "fp32", "bf16"][int(rng.integers(4))]
    constrained = bool(rng.integers(3) == 0)
    bwd_tile = int(rng.choice([0, 0, 128, 256]))
    data = orc.make_synthetic(C, K, V, seed=100 + seed)
    n = 3
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=float(rng.choice([0.0, 0.5])), lambda_r=float(rng.choice([0.0, 1e-3])))
    kw = {}
    if constrained:
        lam.update(lambda_count=1.0, lambda_f_reg=1.0)
        tc = float(max(1, C // 3))
        M0, F0 = orc.reference_init_MF_constrained(C, V, seed)
        o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(n, 0.1)
        kw = dict(F0=F0, mode="constrained", target_count=tc)
    else:
        M0 = orc.reference_init_M(C, V, seed)
        o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(n, 0.1)
    what = (seed, world, C, K, V, prec, constrained, bwd_tile)

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision=prec, lambdas=lam, comm=comm, bwd_tile=bwd_tile, **kw)
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist, 0)
        res = sh.result_local(with_filter=constrained)
        out = dict(hist=hist.cpu().numpy(), P=res[0].cpu().numpy(), range=res[1], F=res[2].cpu().numpy() if constrained else None)
        sh.release()
        return out

    res = run_ranks(world, rank_fn)
    for x in res[1:]:
        np.testing.assert_array_equal(x["hist"], res[0]["hist"])
    tol = pc.TOL[prec]
    hh = res[0]["hist"].astype(np.float64)
    for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss")):
        ref = np.array([float(x) for x in ho[k]])
        assert np.abs(hh[:, col] - ref).max() <= 3 * tol["loss"] * max(1.0, np.abs(ref).max()), (what, k)
    assert res[0]["range"][0] == 0 and res[-1]["range"][1] == V
    P = np.concatenate([x["P"] for x in res], axis=1)
    assert np.abs(P - Po).max() <= tol["P"], what
    assert np.linalg.norm(P - Po) / np.linalg.norm(Po) <= (1e-4 if prec != "bf16" else 2e-2), what
    if constrained:
        for x in res:
            assert np.abs(x["F"] - Fo).max() <= (2e-5 if prec != "bf16" else 5e-3), what
