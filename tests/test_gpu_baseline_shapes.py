"""GPU parity (-m gpu) at the tile counts of the BASELINE configurations that the golden fixtures and the K <= 1000 / single-engine
production tests do not reach ("untested at production size"):

 cfg3  the spot-shard path (row-dot backward GEMM + tg_rowsum_parts + tg_adam_update + the three exchanges) at K = 1000:
       the 1/8 shard SHAPE of cfg2, 30 000 x 1 000 x 1 250, driven by a 1-rank RCCL group, and 4 200 x 1 000 x 1 500 split into
       2 / 3 in-process shards on both backward tile geometries -- against the fp64 oracle.
 cfg5b the spatial terms at the size `bench.py --workload cfg5b` times (30 000 x 1 000 x 10 000, 6-neighbour hex CSR graph with
       ~60k non-zeros, 18 cell types) against the reference's op sequence with DENSE V x V operators (oracle/torch_port.py run by
       PyTorch-ROCm fp32 on the same GPU), plus tg_spmm at gene counts 257 / 1003 (not multiples of 4) against the fp64 oracle.
 cfg4  K = 2 000 (8 gene tiles, 4 wide forward tiles, 63 backward contraction steps) in bf16 and bf16x3 against the fp64 oracle,
       and one problem whose C x V arrays exceed 2^32 elements (70 000 x 8 x 70 000) against the reference's op sequence on the
       GPU, sampled where a 32-bit element index would have wrapped.

The checker is the oracle; the thing under test is the C-ABI library.  Tolerances: tests/parity_common.py (fp32 path: loss 1e-5,
gradient rel 1e-5 vs fp64, 1e-4 vs the fp32 torch run)."""
import os

import numpy as np
import pytest
import torch

from tests import parity_common as pc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BETA1 = 0.9


def _grad_from_first_moment(eng, V):
    _, m1, _, _ = eng.logits()
    return m1[:, :V] / (1.0 - BETA1)            # exp_avg after one step = (1 - beta1) * grad


def _check_hist(h, ref, cols, tol=1e-5):
    from tangram_amd import _capi
    for k, col in cols:
        r = np.asarray([float(x) for x in ref[k]], dtype=np.float64)
        err = float(np.abs(h[:len(r), col].astype(np.float64) - r).max())
        assert err <= tol * max(1.0, float(np.abs(r).max())), f"{k}: max per-epoch |delta| {err:.3e}"


# ------------------------------------------------------------------------------------------------------------------------
# cfg3: spot shards at K = 1000
# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_shard_k1000():
    from oracle import tangram_oracle as orc
    C, K, V, n = 4200, 1000, 1500, 5
    data = orc.make_synthetic(C, K, V, seed=21)
    M0 = orc.reference_init_M(C, V, 5)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-4)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    _, dM = o.loss_and_grad()
    Po, ho = o.train(n, 0.1)
    return dict(C=C, K=K, V=V, n=n, data=data, M0=M0, lam=lam, dM=dM, P=Po, hist=ho)


@pytest.mark.parametrize("world,tile", [(2, 256), (3, 256), (2, 128), (3, 128)])
def test_spot_shards_k1000_against_oracle_fp64(oracle_shard_k1000, world, tile):
    """4 200 x 1 000 x 1 500 as 2 / 3 shards of 750 / 500 spots (threads of one process, tests/local_comm.py): the row-dot
    backward GEMM with 32 contraction steps on 256^2 and on 128^2 tiles, tg_rowsum_parts, tg_adam_update and the exchanges."""
    from tangram_amd.sharded import make_sharded
    from tangram_amd import _capi
    from tests.local_comm import run_ranks
    o = oracle_shard_k1000
    data, V, n = o["data"], o["V"], o["n"]

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], o["M0"], d=data["d"], device=DEV, precision="bf16x3", lambdas=o["lam"], comm=comm,
                          bwd_tile=tile)
        hist = sh.eng.new_history(n)
        sh.run(1, 0.1, hist, 0)
        g = _grad_from_first_moment(sh.eng, sh.eng.V).cpu().numpy().astype(np.float64)
        sh.run(n - 1, 0.1, hist, 1)
        out = hist.cpu().numpy(), sh.result_full().cpu().numpy(), g
        sh.release()
        return out

    res = run_ranks(world, rank_fn)
    g = np.concatenate([r[2] for r in res], axis=1)
    rel = np.linalg.norm(g - o["dM"]) / np.linalg.norm(o["dM"])
    assert rel <= 1e-5, f"first-step gradient rel err {rel:.3e}"
    cols = (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("vg_reg", _capi.H_VG), ("kl_reg", _capi.H_KL),
            ("entropy_reg", _capi.H_ENTROPY))
    for hist, P, _ in res:
        _check_hist(hist, o["hist"], cols)
        assert float(np.abs(P - o["P"]).max()) <= 2e-4
        np.testing.assert_array_equal(hist, res[0][0])        # every rank holds the same global history


def test_cfg3_shard_shape_under_one_rank_rccl_against_oracle_fp64():
    """The shape one rank of the 8-GPU cfg3 run steps: 30 000 cells x 1 000 genes x 1 250 spots (118 x 5 backward tiles of 256^2
    = the thin-grid rule's 128^2 tiles, 32 contraction steps, 12 forward splits), through the real sharded step with RCCL
    (1-rank group: all three exchanges are issued) -- first-step gradient and a 5-epoch history against the fp64 oracle."""
    import torch.distributed as dist
    from oracle import tangram_oracle as orc
    from tangram_amd.sharded import ShardedMapperEngine
    from tangram_amd import _capi
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    C, K, V, n = 30000, 1000, 1250, 5
    data = orc.make_synthetic(C, K, V, seed=4)
    M0 = orc.reference_init_M(C, V, 9)
    lam = dict(lambda_g1=1.0, lambda_d=1.0)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    _, dM = o.loss_and_grad()
    Po, ho = o.train(n, 0.1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sh = ShardedMapperEngine(data["S"], data["G"], M0, data["d"], n_spots_total=V, device=dev, precision="bf16x3", lambdas=lam)
        assert sh.transport == "rccl"
        hist = sh.eng.new_history(n)
        sh.run(1, 0.1, hist, 0)
        g = _grad_from_first_moment(sh.eng, V).cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(g - dM) / np.linalg.norm(dM)
        assert rel <= 1e-5, f"first-step gradient rel err {rel:.3e}"
        row_rel = np.linalg.norm(g - dM, axis=1) / np.maximum(np.linalg.norm(dM, axis=1), 1e-300)
        assert float(row_rel.max()) <= 1e-4, f"worst cell row {int(row_rel.argmax())}: {float(row_rel.max()):.3e}"
        sh.run(n - 1, 0.1, hist, 1)
        _check_hist(hist.cpu().numpy(), ho, (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("kl_reg", _capi.H_KL)))
        assert float(np.abs(sh.result_full().cpu().numpy() - Po).max()) <= 2e-4
        sh.release()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------
# cfg5b: spatial terms at the size the bench times
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [257, 1003])
def test_csr_spatial_terms_ragged_gene_counts_against_oracle_fp64(K):
    """tg_spmm (4 genes per thread) at gene counts that are not multiples of 4, on a 3 000-spot hex graph (~18k non-zeros):
    neighbourhood + cell-type-island terms against the fp64 oracle with dense operators -- gradient and 3 epochs."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import hex_grid_graph
    from tangram_amd import _capi
    C, V, T, n = 500, 3000, 7, 3
    data = orc.make_synthetic(C, K, V, seed=K, n_types=T)
    M0 = orc.reference_init_M(C, V, 13)
    N, W = hex_grid_graph(V)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, voxel_weights=W.toarray(),
                         neighborhood_filter=N.toarray(), ct_encode=data["ct_encode"], **lam)
    _, dM = o.loss_and_grad()
    Po, ho = o.train(n, 0.1)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam,
                        voxel_weights=W, neighborhood_filter=N, ct_encode=data["ct_encode"])
    hist = e.new_history(n)
    e.step(1, 0.1, hist, 0)
    g = _grad_from_first_moment(e, V).cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(g - dM) / np.linalg.norm(dM)
    assert rel <= 1e-5, f"first-step gradient rel err {rel:.3e}"
    e.step(n - 1, 0.1, hist, 1)
    _check_hist(hist.cpu().numpy(), ho, (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("kl_reg", _capi.H_KL)))
    assert float(np.abs(e.result().cpu().numpy() - Po).max()) <= 2e-4
    e.release()


def test_cfg5b_full_size_against_reference_op_sequence():
    """The exact inputs of `bench.py --workload cfg5b` (30 000 x 1 000 x 10 000, hex CSR graph, 18 cell types,
    lambda_neighborhood_g1 = 0.96, lambda_ct_islands = 0.17) against the reference's op sequence with the reference's DENSE
    10 000 x 10 000 W and N (mapping_optimizer.py:234-248), fp32 on the same GPU: first-step gradient and 3 steps of total_loss."""
    from oracle.torch_port import TorchPortMapper
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits, hex_grid_graph, cell_type_encoding
    from tangram_amd import _capi
    C, K, V, T, n = 30000, 1000, 10000, 18, 3
    w = make_workload(C, K, V, DEV, seed=0)
    M0 = init_logits(C, V, DEV, seed=42)
    N, W = hex_grid_graph(V)
    E = cell_type_encoding(w["assign"].cpu().numpy(), V, T)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17)
    tiny = lambda x: x[:4].cpu().numpy()
    m = TorchPortMapper(tiny(w["S"]), tiny(w["G"]), d=tiny(w["d"]), lambda_g1=1, lambda_d=1, lambda_neighborhood_g1=0.96,
                        voxel_weights=np.eye(4, dtype=np.float32), lambda_ct_islands=0.17, neighborhood_filter=np.eye(4, dtype=np.float32),
                        ct_encode=np.zeros((4, T), np.float32), M0=np.zeros((4, 4)))
    m.S, m.G, m.d = w["S"], w["G"], w["d"]                          # full-size tensors, already on the GPU
    m.W = torch.as_tensor(W.toarray(), device=DEV)
    m.N = torch.as_tensor(N.toarray(), device=DEV)
    m.E = torch.as_tensor(E, device=DEV)
    m.M = M0.clone().requires_grad_(True)
    opt = torch.optim.Adam([m.M], lr=0.1)
    zero = torch.zeros(1, device=DEV)
    ref_hist, ref_grad = [], None
    for i in range(n):
        # (TorchPortMapper.loss builds its zero for torch.max on the host; same arithmetic with the device tensors here)
        P = torch.softmax(m.M, dim=1)
        Gp = P.t() @ m.S
        gv = torch.nn.functional.cosine_similarity(Gp, m.G, dim=0).mean()
        dens = torch.nn.KLDivLoss(reduction="sum")(torch.log(P.sum(dim=0) / C), m.d)
        nb = 0.96 * torch.nn.functional.cosine_similarity(m.W @ Gp, m.W @ m.G, dim=0).mean()
        cm = P.t() @ m.E
        ct = 0.17 * torch.max(cm - m.N @ cm, zero).mean()
        total = -gv + dens + ct - nb
        opt.zero_grad()
        total.backward()
        if i == 0:
            ref_grad = m.M.grad.detach().clone()
        opt.step()
        ref_hist.append(dict(total_loss=float(total), main_loss=float(gv), kl_reg=float(dens)))
        del P, Gp, cm, total
    del opt, m
    torch.cuda.empty_cache()
    e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=lam,
                        voxel_weights=W, neighborhood_filter=N, ct_encode=E)
    del M0
    hist = e.new_history(n)
    e.step(1, 0.1, hist, 0)
    g = _grad_from_first_moment(e, V)
    rel = float(torch.linalg.norm(g - ref_grad) / torch.linalg.norm(ref_grad))
    assert rel <= 1e-4, f"first-step gradient rel err {rel:.3e}"
    col_rel = torch.linalg.norm(g - ref_grad, dim=0) / torch.linalg.norm(ref_grad, dim=0).clamp_min(1e-30)
    assert float(col_rel.max()) <= 1e-3, f"worst spot column {int(col_rel.argmax())}: {float(col_rel.max()):.3e}"
    del g, ref_grad, col_rel
    e.step(n - 1, 0.1, hist, 1)
    h = hist.cpu().numpy().astype(np.float64)
    for k, col, tol in (("total_loss", _capi.H_TOTAL, 2e-5), ("main_loss", _capi.H_MAIN, 1e-5), ("kl_reg", _capi.H_KL, 1e-5)):
        ref = np.array([t[k] for t in ref_hist])
        assert float(np.abs(h[:, col] - ref).max()) <= tol * max(1.0, float(np.abs(ref).max())), (k, h[:, col], ref)
    e.release()


# ------------------------------------------------------------------------------------------------------------------------
# cfg4: K = 2000, and arrays beyond 2^32 elements
# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_k2000():
    from oracle import tangram_oracle as orc
    C, K, V, n = 4200, 2000, 1500, 4
    data = orc.make_synthetic(C, K, V, seed=77)
    M0 = orc.reference_init_M(C, V, 6)
    lam = dict(lambda_g1=1.0, lambda_d=1.0)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    _, dM = o.loss_and_grad()
    Po, ho = o.train(n, 0.1)
    return dict(C=C, K=K, V=V, n=n, data=data, M0=M0, lam=lam, dM=dM, P=Po, hist=ho, Ghat=Po.T @ data["S"].astype(np.float64))


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_k2000_against_oracle_fp64(oracle_k2000, precision):
    """cfg4's gene count: 2 001 operand columns pad to 2 048 = 8 gene tiles of 256 (4 wide forward tiles), 64 (bf16x3) / 32 (bf16)
    backward contraction steps."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    o = oracle_k2000
    data, V, n = o["data"], o["V"], o["n"]
    e = HipMapperEngine(data["S"], data["G"], o["M0"], d=data["d"], device=DEV, precision=precision, lambdas=o["lam"])
    hist = e.new_history(n)
    e.step(1, 0.1, hist, 0)
    g = _grad_from_first_moment(e, V).cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(g - o["dM"]) / np.linalg.norm(o["dM"])
    assert rel <= (1e-5 if precision != "bf16" else 1e-2), f"first-step gradient rel err {rel:.3e}"
    e.step(n - 1, 0.1, hist, 1)
    tol = pc.TOL[precision]
    _check_hist(hist.cpu().numpy(), o["hist"], (("total_loss", _capi.H_TOTAL), ("main_loss", _capi.H_MAIN), ("kl_reg", _capi.H_KL)),
                tol=tol["loss"])
    assert float(np.abs(e.result().cpu().numpy() - o["P"]).max()) <= tol["P"]
    Gh = e.project().cpu().numpy()
    assert np.linalg.norm(Gh - o["Ghat"]) / np.linalg.norm(o["Ghat"]) <= tol["ghat"]
    e.release()


def test_arrays_beyond_2_pow_32_elements_against_reference_op_sequence():
    """70 000 cells x 8 genes x 70 000 spots: M, both Adam moments and X hold 4.9e9 elements each (19.6 GB; cfg4 has 1e10), so
    every element offset past cell 61 356 exceeds 2^32.  Against the reference's op sequence in fp32 on the same GPU: loss terms
    of 2 steps, and the first-step gradient on 256 rows sampled from the LAST 1 000 cells (plus 64 from the first 1 000)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    from tangram_amd import _capi
    free, _ = torch.cuda.mem_get_info()
    if free < 230 * 2**30:
        pytest.skip("needs ~220 GB of free HBM")
    C, K, V, n = 70000, 8, 70000, 2
    assert C * ((V + 63) // 64 * 64) > 2**32
    w = make_workload(C, K, V, DEV, seed=5)
    M0 = init_logits(C, V, DEV, seed=11)
    gen = torch.Generator(device="cpu").manual_seed(1)
    rows = torch.cat([C - 1000 + torch.randperm(1000, generator=gen)[:256], torch.randperm(1000, generator=gen)[:64]]).to(DEV)
    # reference op sequence, fp32 on the GPU (softmax, matmul, cosine_similarity, KLDivLoss, autograd, Adam)
    M = M0.clone().requires_grad_(True)
    opt = torch.optim.Adam([M], lr=0.1)
    ref_hist, ref_grad = [], None
    for i in range(n):
        P = torch.softmax(M, dim=1)
        Gp = P.t() @ w["S"]
        gv = torch.nn.functional.cosine_similarity(Gp, w["G"], dim=0).mean()
        dens = torch.nn.KLDivLoss(reduction="sum")(torch.log(P.sum(dim=0) / C), w["d"])
        total = -gv + dens
        del P
        opt.zero_grad()
        total.backward()
        if i == 0:
            ref_grad = M.grad[rows].detach().clone()
        opt.step()
        ref_hist.append(dict(total_loss=float(total), main_loss=float(gv), kl_reg=float(dens)))
        del Gp, total
    del opt, M
    torch.cuda.empty_cache()
    e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    del M0
    torch.cuda.empty_cache()
    hist = e.new_history(n)
    e.step(1, 0.1, hist, 0)
    _, m1, _, _ = e.logits()
    g = m1[rows][:, :V] / (1.0 - BETA1)
    rel = float(torch.linalg.norm(g - ref_grad) / torch.linalg.norm(ref_grad))
    assert rel <= 1e-4, f"first-step gradient on the sampled rows: rel err {rel:.3e}"
    row_rel = torch.linalg.norm(g - ref_grad, dim=1) / torch.linalg.norm(ref_grad, dim=1).clamp_min(1e-30)
    assert float(row_rel.max()) <= 1e-3, f"worst sampled row {int(rows[int(row_rel.argmax())])}: {float(row_rel.max()):.3e}"
    e.step(n - 1, 0.1, hist, 1)
    h = hist.cpu().numpy().astype(np.float64)
    for k, col in (("main_loss", _capi.H_MAIN), ("kl_reg", _capi.H_KL), ("total_loss", _capi.H_TOTAL)):
        ref = np.array([t[k] for t in ref_hist])
        assert float(np.abs(h[:, col] - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max())), (k, h[:, col], ref)
    # the mapping of the last cells (softmax of rows stored past the 2^32-element mark) is a distribution over the spots
    P_tail = torch.softmax(e.logits()[0][C - 8:, :V], dim=1)
    assert torch.allclose(P_tail.sum(dim=1), torch.ones(8, device=DEV), atol=1e-5)
    e.release()


def test_full_size_properties_cfg4():
    """BASELINE config 4 at FULL size on one GPU (200 000 x 2 000 x 50 000, bf16 GEMM operands, fp32 state: 120 GB of logits and Adam
    moments, every C x V array beyond 2^32 elements), started from the seam's device-side initialiser (`Mapper(init="device")`'s
    generator -- the reference's host draw of this plane is 80 GB of float64).  Neither the reference nor the oracle can run this
    size: size-independent invariants (mapping_optimizer.py:201-221, tests/tangram_test.py:159-210)."""
    import gc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.device_init import device_normal
    from tangram_amd.synthetic import make_workload
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip(f"needs ~230 GB of free HBM, {free >> 30} GB free")
    C, K, V = 200000, 2000, 50000
    w = make_workload(C, K, V, DEV, seed=0)
    M0 = device_normal(C, V, DEV, seed=42)
    first = M0[:2].clone(), M0[-2:].clone()
    e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16", lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
    del M0
    torch.cuda.empty_cache()
    M, m1, m2, _ = e.logits()
    assert M.shape[0] * M.shape[1] > 2 ** 32
    assert torch.equal(M[:2, :V], first[0]) and torch.equal(M[-2:, :V], first[1])          # the logits landed where they belong, also past 2^32
    n = 6
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    h = hist.cpu().numpy()
    assert np.isfinite(h[:, [0, 1, 3]]).all()
    assert (np.diff(h[:, 1]) > 0).all(), "gene-voxel score must increase in the first epochs"
    assert (np.diff(h[:, 3]) < 0).all(), "KL density term must decrease"
    P = e.result()                                                                         # 40 GB
    rs = P.sum(dim=1)
    assert float((rs - 1).abs().max()) < 1e-4 and float(P.min()) >= 0.0
    # every cell row moved, in particular those whose elements sit beyond a 32-bit index (cells >= 2^32 / pitch = 85 8xx)
    moved = (m1[:, :V].abs().amax(dim=1) > 0)
    assert bool(moved.all()), f"{int((~moved).sum())} cell rows were never updated"
    del rs, moved
    # train-score invariant: the gene score of the NEXT step is the cosine recomputed from P (bf16 operands in the library's GEMM)
    Gp = torch.empty((V, K), dtype=torch.float32, device=DEV)
    for k0 in range(0, K, 500):                                                            # (fp32 P^T S in gene blocks)
        Gp[:, k0:k0 + 500] = P.t() @ w["S"][:, k0:k0 + 500]
    cos = torch.nn.functional.cosine_similarity(Gp, w["G"], dim=0).mean().item()
    colsum = P.sum(dim=0)
    kl = float((torch.xlogy(w["d"], w["d"]) - w["d"] * torch.log(colsum / C)).sum().item())        # KLDivLoss(reduction="sum"), :218
    del Gp, colsum
    hist2 = e.new_history(1)
    e.step(1, 0.1, hist2)
    assert abs(cos - float(hist2[0, 1].item())) < 1e-3, (cos, float(hist2[0, 1].item()))
    assert abs(kl - float(hist2[0, 3].item())) < 1e-4 * max(1.0, abs(kl)), (kl, float(hist2[0, 3].item()))
    # softmax statistics carried from step to step equal a from-scratch softmax: first rows and rows past the 32-bit mark
    del P
    torch.cuda.empty_cache()
    P_now = e.result()
    for rows in (slice(0, 1024), slice(C - 1024, C), slice(86000, 87024)):
        ref = torch.softmax(M[rows, :V], dim=1)
        assert float((P_now[rows] - ref).abs().max()) < 1e-6
    assert e.logits()[3] == n + 1
    e.release()
    del P_now, M, m1, m2, w
    gc.collect()
    torch.cuda.empty_cache()


def test_full_size_cfg4_as_eight_spot_shards():
    """BASELINE config 4 in ITS decomposition -- 200 000 x 2 000 x 50 000, bf16 operands, the spots as 8 shards of 6 250 -- at full
    size: the eight shards are threads of this process on ONE GPU (tests/local_comm.py; their exchanges sum in rank order) stepping
    through the sharded C schedule from the seam's device-side initialiser (`make_sharded(..., device_init_seed=)`: every shard
    generates exactly its columns of the 10^10-element plane, global offsets beyond 2^32).  Checked against the UNSHARDED run of the
    same problem on the same GPU: the shards' logits are the unsharded run's columns (initially bit for bit, after the steps within
    the bf16 bound of the full-size live-reference cases), and every rank holds the same global history, equal to the unsharded one."""
    import gc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.device_init import device_normal
    from tangram_amd.sharded import make_sharded, shard_bounds
    from tangram_amd.synthetic import make_workload
    from tests.local_comm import run_ranks
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 250 * (1 << 30):
        pytest.skip(f"needs ~230 GB of free HBM, {free >> 30} GB free")
    C, K, V, world, n = 200000, 2000, 50000, 8, 4
    lam = dict(lambda_g1=1.0, lambda_d=1.0)
    w = make_workload(C, K, V, DEV, seed=0)
    bounds = [shard_bounds(V, world, r) for r in range(world)]
    cols = torch.tensor(sorted({c for lo, hi in bounds for c in (lo, lo + 1, (lo + hi) // 2, hi - 2, hi - 1)}), device=DEV)
    # ---- the unsharded run: history + the sampled columns of the logits before and after
    M0 = device_normal(C, V, DEV, seed=42)
    e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16", lambdas=lam)
    del M0
    torch.cuda.empty_cache()
    init_cols = e.logits()[0][:, cols].clone()
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    ref_hist = hist.cpu().numpy().astype(np.float64)
    ref_cols = e.logits()[0][:, cols].clone()
    e.release()
    del e, hist
    gc.collect()
    torch.cuda.empty_cache()

    # ---- the same problem as 8 spot shards
    def rank_fn(comm):
        sh = make_sharded(w["S"], w["G"], None, d=w["d"], device=DEV, precision="bf16", lambdas=lam, comm=comm, device_init_seed=42)
        lo, hi = bounds[comm.rank]
        mine = ((cols >= lo) & (cols < hi)).nonzero().flatten()
        M = sh.eng.logits()[0]
        assert M.shape[0] == C and sh.eng.V == hi - lo
        first = M[:, cols[mine] - lo].clone()
        h = sh.eng.new_history(n)
        sh.run(n, 0.1, h, 0)
        out = dict(hist=h.cpu().numpy(), idx=mine.cpu().numpy(), first=first, last=M[:, cols[mine] - lo].clone(), range=sh.result_local()[1])
        sh.release()
        return out

    res = run_ranks(world, rank_fn)
    torch.cuda.synchronize()
    assert [x["range"] for x in res] == bounds and (C * V) > 2 ** 32
    for x in res[1:]:
        np.testing.assert_array_equal(x["hist"], res[0]["hist"])                 # the global history, identical on every rank
    hh = res[0]["hist"].astype(np.float64)
    live = [0, 1, 3]                                                             # total, gene score, KL
    assert np.isfinite(hh[:, live]).all()
    assert np.abs(hh[:, live] - ref_hist[:, live]).max() <= 1e-4 * max(1.0, np.abs(ref_hist[:, live]).max()), (hh[:, live], ref_hist[:, live])
    for x in res:
        idx = torch.as_tensor(x["idx"], device=DEV)
        assert torch.equal(x["first"], init_cols[:, idx])                        # each shard generated exactly its columns
        d = (x["last"] - ref_cols[:, idx]).abs()
        assert float((d > 1e-3).float().mean()) <= 1e-4 and float(d.max()) < 1.0, (float(d.max()), float((d > 1e-3).float().mean()))
    del res, w, init_cols, ref_cols
    gc.collect()
    torch.cuda.empty_cache()
