"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against
 (i) fixtures generated from the unmodified reference (tests/golden, fp64 ground truth),
 (ii) the oracle on seeded inputs at sizes it finishes in seconds,
 (iii) size-independent properties at the BASELINE.json shape (30k x 1k x 10k)."""
import numpy as np
import pytest
import torch

from tests import parity_common as pc
from oracle.gen_golden import CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MAPPER_CASES = [n for n, c in CASES.items() if c[5] in ("cells", "clusters", "constrained", "spatial", "autocorr", "grid")]


def test_native_library_is_the_one_running():
    from tangram_amd import _capi
    assert torch.cuda.is_available()
    assert not _capi.is_emulated()
    assert _capi.lib().tg_abi_version() == _capi.TG_ABI_VERSION == 6


@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16"])
@pytest.mark.parametrize("name", MAPPER_CASES)
def test_golden_reference_trajectories(name, precision):
    res = pc.run_case(name, DEV, precision)
    pc.check_against_golden(res, precision, full_length=True)
    if CASES[name][0] <= 32:      # at most 32 cells: the run above took the clusters-mode kernels; the GEMM kernels on the same case
        res = pc.run_case(name, DEV, precision, pin_gemm=True)
        pc.check_against_golden(res, precision, full_length=True)


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_medium_problem_against_oracle_fp64(precision):
    """SURVEY 8c calibration size: 1500 x 120 x 400, 100 epochs, planted-assignment data."""
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper
    C, K, V = 1500, 120, 400
    data = orc.make_synthetic(C, K, V, seed=0)
    M0 = orc.reference_init_M(C, V, 42)
    n = 100
    m = Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device=DEV, gemm_precision=precision, M_init=M0)
    P, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], lambda_d=1, M0=M0, dtype=np.float64)
    Po, ho = o.train(n, 0.1)
    for k in ("main_loss", "kl_reg", "total_loss"):
        err = np.abs(np.array([float(x) for x in hist[k]]) - np.array(ho[k])).max()
        assert err <= 1e-5, (k, err)
    assert np.abs(P - Po).max() <= 2e-4
    Gh = m.project_genes_device().cpu().numpy()
    ref = Po.astype(np.float64).T @ data["S"].astype(np.float64)
    assert np.linalg.norm(Gh - ref) / np.linalg.norm(ref) <= 1e-4
    assert (P.argmax(1) == Po.argmax(1)).mean() > 0.99


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_banded_tile_order_against_oracle_fp64(precision):
    """A shape with >= 16 cell tiles and >= 16 spot tiles, so that the XCD-banded supertile order is active."""
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper
    C, K, V = 2500, 70, 2200
    data = orc.make_synthetic(C, K, V, seed=12)
    M0 = orc.reference_init_M(C, V, 11)
    n = 25
    lam = dict(lambda_d=1, lambda_g1=1, lambda_g2=0.5)
    m = Mapper(data["S"], data["G"], d=data["d"], device=DEV, gemm_precision=precision, M_init=M0, **lam)
    P, hist = m.train(num_epochs=n, learning_rate=0.1, print_each=None)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)
    tol = pc.TOL[precision]
    for k in ("main_loss", "vg_reg", "kl_reg", "total_loss"):
        err = np.abs(np.array([float(x) for x in hist[k]]) - np.array(ho[k])).max()
        assert err <= tol["loss"], (k, err)
    assert np.abs(P - Po).max() <= tol["P"]


def test_single_step_gradient_fp32_path():
    """One Adam step from zero moments moves every logit by -lr*sign(g) (|g| >> eps): compare the implied
    gradient signs and, through a tiny-lr second run, magnitudes against the oracle's analytic gradient."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 700, 90, 333
    data = orc.make_synthetic(C, K, V, seed=4)
    M0 = orc.reference_init_M(C, V, 9)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.7)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="fp32", lambdas=lam)
    e.step(1, 0.1)
    M1, m1, m2, st = e.logits()
    g = (m1[:, :V] / 0.1).cpu().numpy()                  # exp_avg after one step = (1-beta1) * grad
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    _, dM = o.loss_and_grad()
    rel = np.linalg.norm(g - dM) / np.linalg.norm(dM)
    assert rel <= 1e-5, rel


def test_forward_splits_and_determinism():
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 2000, 200, 700
    data = orc.make_synthetic(C, K, V, seed=8)
    M0 = orc.reference_init_M(C, V, 3)
    outs = []
    for splits in (1, 4, 4):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3",
                            lambdas=dict(lambda_d=1.0), fwd_splits=splits)
        e.step(5, 0.1)
        outs.append(e.result().cpu().numpy())
    assert np.array_equal(outs[1], outs[2]), "same configuration must be bit-reproducible (no float atomics)"
    assert np.abs(outs[0] - outs[1]).max() < 1e-6


def test_full_size_properties_cfg2():
    """BASELINE.json shape, 1 GPU: size-independent invariants (the oracle cannot run this size)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    C, K, V = 30000, 1000, 10000
    w = make_workload(C, K, V, DEV, seed=0)
    M0 = init_logits(C, V, DEV, seed=42)
    e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_d=1.0))
    del M0
    n = 12
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    h = hist.cpu().numpy()
    assert np.isfinite(h[:, :4][:, [0, 1, 3]]).all()
    assert (np.diff(h[:, 1]) > 0).all(), "gene-voxel score must increase monotonically in the first epochs"
    assert (np.diff(h[:, 3]) < 0).all(), "KL density term must decrease"
    P = e.result()
    rs = P.sum(dim=1)
    assert float((rs - 1).abs().max()) < 1e-4 and float(P.min()) >= 0.0
    # train-score invariant (reference tests/tangram_test.py:159-210): recompute the gene score from P after one more step
    hist2 = e.new_history(1)
    Gp = P.t() @ w["S"]
    cos = torch.nn.functional.cosine_similarity(Gp, w["G"], dim=0).mean().item()
    e.step(1, 0.1, hist2)
    assert abs(cos - float(hist2[0, 1].item())) < 1e-4
    # softmax statistics carried across iterations equal a from-scratch recomputation
    M, m1, m2, step = e.logits()
    P2 = torch.softmax(M[:, :V], dim=1)
    assert float((e.result() - P2).abs().max()) < 1e-6
    assert step == n + 1


def test_checkpoint_resume_on_gpu():
    """tg_mapper_state / tg_mapper_set_step at production tiles: logits, both Adam moments and the step counter copied into a fresh
    handle continue the run (same trajectory up to the rounding of the rebuilt softmax normaliser)."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 4500, 60, 1300
    data = orc.make_synthetic(C, K, V, seed=12)
    M0 = orc.reference_init_M(C, V, 2)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3)
    mk = lambda M: HipMapperEngine(data["S"], data["G"], M, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam)
    a = mk(M0)
    ha = a.new_history(9)
    a.step(9, 0.1, ha)
    b = mk(M0)
    b.step(4, 0.1, b.new_history(4))
    Mb, m1b, m2b, step = b.logits()
    c = mk(np.zeros_like(M0))
    Mc, m1c, m2c, _ = c.logits()
    Mc.copy_(Mb); m1c.copy_(m1b); m2c.copy_(m2b)
    c.set_step(step)
    hc = c.new_history(9)
    c.step(5, 0.1, hc, 4)
    np.testing.assert_allclose(hc.cpu().numpy()[4:, :5], ha.cpu().numpy()[4:, :5], rtol=5e-6, atol=1e-7)
    assert float((c.result() - a.result()).abs().max()) <= 5e-7
    assert c.logits()[3] == 9
    for e in (a, b, c):
        e.release()


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_pipelined_schedule_matches_sequential(precision):
    """3-stream cell-band pipeline (backward | Adam | next forward overlap) vs the one-stream schedule on real hardware:
    any missing event dependency shows up here as a difference."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = 5000, 100, 1500
    data = orc.make_synthetic(C, K, V, seed=31)
    M0 = orc.reference_init_M(C, V, 8)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.3, lambda_r=1e-3)
    outs = []
    for bands in (1, 5, 5):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision=precision, lambdas=lam,
                            pipeline_bands=bands)
        hist = e.new_history(30)
        e.step(17, 0.1, hist, 0)
        e.step(13, 0.1, hist, 17)
        outs.append((e.result().cpu().numpy(), hist.cpu().numpy()))
    assert np.array_equal(outs[1][0], outs[2][0]) and np.array_equal(outs[1][1][:, :5], outs[2][1][:, :5]), "pipelined runs must be bit-reproducible"
    np.testing.assert_allclose(outs[0][1][:, :5], outs[1][1][:, :5], atol=5e-6 if precision != "bf16" else 1e-4, rtol=1e-6)
    assert np.abs(outs[0][0] - outs[1][0]).max() < (1e-4 if precision != "bf16" else 5e-3)


@pytest.mark.parametrize("shape", [(2, 3, 4), (5, 1, 7), (3, 2, 1), (1, 4, 6), (17, 130, 9), (18, 250, 9852), (70000, 8, 70),
                                   (18, 60, 20000), (5, 8, 70000)])
def test_degenerate_and_extreme_shapes(shape):
    """Single cell / gene / spot, a clusters-mode shape (18 x 250 x 9852), a tall shape (grid limits) and clusters mode on
    rows beyond the one-kernel update (the long-row path on a handful of rows: 1 024-thread `tg_adam_update`, `tg_gene_reduce_tall`)."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    C, K, V = shape
    rng = np.random.default_rng(C * 100 + K * 10 + V)
    S = rng.integers(1, 5, size=(C, K)).astype(np.float32)
    G = rng.integers(1, 5, size=(V, K)).astype(np.float32)
    d = (G.sum(1) / G.sum()).astype(np.float32)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    for prec in ("fp32", "bf16x3"):
        e = HipMapperEngine(S, G, M0, d=d, device=DEV, precision=prec, lambdas=lam)
        hist = e.new_history(3)
        e.step(3, 0.1, hist)
        o = orc.OracleMapper(S, G, d=d, M0=M0, dtype=np.float64, **lam)
        Po, ho = o.train(3, 0.1)
        np.testing.assert_allclose(hist[:, _capi.H_TOTAL].cpu().numpy(), np.array(ho["total_loss"]), atol=2e-5, err_msg=prec)
        np.testing.assert_allclose(e.result().cpu().numpy(), Po, atol=5e-5, err_msg=prec)
        np.testing.assert_allclose(e.project().cpu().numpy(), Po.T @ S.astype(np.float64), rtol=2e-4, atol=1e-5, err_msg=prec)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)])
def test_project_genes_all_genes(precision, tol):
    """tg_mapper_project_genes (reference utils.py:366-368, `adata_map.X.T @ adata_sc.X`): several gene blocks with a
    ragged tail and a padded row pitch, against the float64 product of the SAME mapping; stated tolerance = max abs
    error relative to the largest entry (fp32 GEMM round-off / split-bf16 / plain bf16 operands)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    C, K, V, K_all = 3000, 200, 1100, 730
    w = make_workload(C, K, V, DEV, seed=3)
    e = HipMapperEngine(w["S"], w["G"], init_logits(C, V, DEV, seed=1), d=w["d"], device=DEV, precision=precision,
                        lambdas=dict(lambda_d=1.0))
    e.step(5, 0.1, e.new_history(5))
    g = torch.Generator(device="cpu").manual_seed(0)
    wide = (torch.rand((C, K_all + 6), generator=g) * (torch.rand((C, K_all + 6), generator=g) < 0.3)).to(DEV) * 7.0
    S_all = wide[:, 3:3 + K_all]
    out = e.project_genes(S_all)
    want = e.result().double().t() @ S_all.double()
    assert out.shape == (V, K_all)
    assert float((out.double() - want).abs().max()) <= tol * float(want.abs().max())
    # training genes through the same entry point == the training-time projection kernel output
    np.testing.assert_allclose(e.project_genes(w["S"]).cpu().numpy(), e.project().cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_project_genes_full_size_linearity_cfg2():
    """BASELINE.json shape: 2 500 genes (2.5 blocks) projected from the resident mapping.  Size-independent checks:
    linearity in S (a gene scaled by 3 and a sum of two genes), column sums (sum_v Ghat[v,k] = sum_c S[c,k] because
    rows of P sum to 1), and block-boundary consistency (a gene's projection does not depend on its block)."""
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    C, K, V = 30000, 1000, 10000
    w = make_workload(C, K, V, DEV, seed=0)
    e = HipMapperEngine(w["S"], w["G"], init_logits(C, V, DEV, seed=42), d=w["d"], device=DEV, precision="bf16x3",
                        lambdas=dict(lambda_d=1.0))
    e.step(3, 0.1, e.new_history(3))
    g = torch.Generator(device="cpu").manual_seed(1)
    S_all = (torch.rand((C, 2500), generator=g) * (torch.rand((C, 2500), generator=g) < 0.3)).to(DEV) * 5.0
    S_all[:, 2400] = 3.0 * S_all[:, 10]
    S_all[:, 2401] = S_all[:, 20] + S_all[:, 1500]
    S_all[:, 999] = S_all[:, 1000]                       # same gene on both sides of a block boundary
    out = e.project_genes(S_all)
    scale = float(out.abs().max())
    assert float((out[:, 2400] - 3.0 * out[:, 10]).abs().max()) <= 2e-5 * scale
    assert float((out[:, 2401] - (out[:, 20] + out[:, 1500])).abs().max()) <= 2e-5 * scale
    assert float((out[:, 999] - out[:, 1000]).abs().max()) <= 1e-6 * scale
    cs = out.double().sum(dim=0); want = S_all.double().sum(dim=0)
    assert float(((cs - want).abs() / want.clamp(min=1.0)).max()) <= 2e-5


@pytest.mark.parametrize("V", [1023, 2049, 4100, 8200, 12500, 16384, 16385, 20000])
@pytest.mark.parametrize("variant", ["plain", "regularised", "constrained"])
def test_update_kernel_row_lengths(V, variant):
    """The fused row-dot + Adam kernel keeps a whole row of M, X and both moments in registers; its instantiations
    (256 or 512 threads x 1..8 float4 per array) and the two-kernel fallback for rows longer than 16 384 spots must all
    reproduce the fp64 oracle (loss trajectory, mapping, filter)."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    C, K = 24, 6
    rng = np.random.default_rng(V)
    S = rng.integers(0, 5, size=(C, K)).astype(np.float32) + 1.0
    G = rng.integers(0, 5, size=(V, K)).astype(np.float32) + 1.0
    d = (G.sum(1) / G.sum()).astype(np.float32)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    n = 3
    if variant == "constrained":
        F0 = rng.normal(size=(C,)).astype(np.float32)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_count=1.0, lambda_f_reg=1.0)
        e = HipMapperEngine(S, G, M0, d=d, F0=F0, mode="constrained", device=DEV, precision="fp32", lambdas=lam, target_count=10.0)
        o = orc.OracleMapperConstrained(S, G, d, M0=M0, F0=F0, dtype=np.float64, target_count=10.0, **lam)
    else:
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
        if variant == "regularised":
            lam.update(lambda_r=1e-3, lambda_l1=1e-4, lambda_l2=1e-5)
        e = HipMapperEngine(S, G, M0, d=d, device=DEV, precision="fp32", lambdas=lam)
        o = orc.OracleMapper(S, G, d=d, M0=M0, dtype=np.float64, **lam)
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    res = o.train(n, 0.1)
    ho = res[-1]
    h = hist.cpu().numpy()
    np.testing.assert_allclose(h[:, _capi.H_TOTAL], np.array(ho["total_loss"], dtype=np.float64), atol=2e-5, rtol=2e-6)
    if variant == "constrained":
        P, F = e.result(with_filter=True)
        np.testing.assert_allclose(P.cpu().numpy(), res[0], atol=2e-5)
        np.testing.assert_allclose(F.cpu().numpy(), res[1], atol=2e-5)
    else:
        np.testing.assert_allclose(e.result().cpu().numpy(), res[0], atol=2e-5)


def test_concurrent_mappings_bit_identical():
    """tangram_amd.train_many: independent mappings -- the five Mapper folds advance as ONE tg_batch (one launch per
    kernel, blockIdx.z = fold), the MapperConstrained ones on their own; then everything on separate HIP streams and host
    threads -- must give the very bits of the same mappings trained one after the other (handles share nothing)."""
    import tangram_amd as tg
    import tangram_amd.mapping_optimizer as mo
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(24, 40, 300, seed=4)
    ds = np.full(24, 1.0 / 24, np.float32)

    def cells(i):
        keep = [g for g in range(40) if g != i]
        return lambda: mo.Mapper(S=data["S"][:, keep], G=data["G"][:, keep], d=data["d"], d_source=ds, lambda_d=1, lambda_g2=0.5,
                                 device=DEV, random_state=i + 1)

    def constrained(i):
        return lambda: mo.MapperConstrained(S=data["S"], G=data["G"], d=data["d"], lambda_d=1, lambda_count=1, lambda_f_reg=1,
                                            target_count=100, device=DEV, random_state=i + 1)

    builders = [cells(i) for i in range(5)] + [constrained(i) for i in range(3)]
    seq = [b().train(num_epochs=120, learning_rate=0.1, print_each=None) for b in builders]
    for kw in (dict(batched="auto"), dict(batched=False, max_concurrent=4)):
        res, mappers = tg.train_many(builders, 120, 0.1, device=DEV, **kw)
        assert len(res) == len(seq) == len(mappers)
        for a, b in zip(seq, res):
            assert len(a) == len(b)
            np.testing.assert_array_equal(a[0], b[0])
            if len(a) == 3:
                np.testing.assert_array_equal(a[1], b[1])
            assert list(a[-1]["main_loss"]) == list(b[-1]["main_loss"])
        for m in mappers:
            m.release()


@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16"])
def test_batched_mappings(precision):
    """tg_batch on the GPU: 8 leave-one-gene-out folds of a clusters-mode shape in one launch per kernel, every fold against the
    fp64 oracle and bit-identical to the fold trained alone; also a shape that takes the 256^2 tiles and several forward splits."""
    from tests.test_batched import check_batched
    tol = pc.TOL[precision]
    # (plain bf16 operands: the per-epoch error of TOL compounds over 12 epochs of an 18-row softmax at lr 0.1 -> x4)
    check_batched(DEV, precision, C=18, K=60, V=1300, B=8, epochs=12, lam=dict(lambda_d=1, lambda_g1=1, lambda_g2=0.5),
                  tol_loss=tol["loss"] * (4 if precision == "bf16" else 1), tol_P=tol["P"])
    if precision == "bf16x3":
        check_batched(DEV, precision, C=4200, K=40, V=1100, B=3, epochs=4, lam=dict(lambda_d=1, lambda_g1=1, lambda_r=1e-3),
                      tol_loss=tol["loss"], tol_P=tol["P"])


def test_batched_tuning_seeds_with_val_each():
    """f-3, the reference's tuning caller (mapping_parameter_tuning.py:110-129): three seeds with val_each=1 in ONE tg_batch give the
    histories (all nine keys) and mappings of three solo runs bit for bit -- at the clusters-mode shape and on the GEMM kernels."""
    from tests.test_batched import check_batched_tuning_seeds
    check_batched_tuning_seeds(DEV, "bf16x3", C=18, K=60, V=1300, epochs=10, val_each=1)
    check_batched_tuning_seeds(DEV, "bf16x3", C=1500, K=40, V=900, epochs=6, val_each=2)


def test_batched_constrained_mappings():
    """tg_batch of MapperConstrained handles on the GPU (cross_val with mode='constrained', utils.py:576-600): 6 folds in one launch
    per kernel incl. the filter's Adam step and the re-folding of the new filters, bit-identical to the folds trained alone and
    against the fp64 oracle; and a shape on the 256-wide tiles."""
    from tests.test_batched import check_batched_constrained
    tol = pc.TOL["bf16x3"]
    check_batched_constrained(DEV, "bf16x3", C=40, K=50, V=1300, B=6, epochs=8, tol_loss=tol["loss"], tol_P=tol["P"])
    check_batched_constrained(DEV, "bf16x3", C=4200, K=30, V=900, B=2, epochs=3, tol_loss=tol["loss"], tol_P=tol["P"])


# (C, K, V, constrained, lambda_g2): the emulator's cases plus the tutorial's cross-validation shape and a four-chunk gene count
SMALL_C_CASES = [
    (5, 40, 70, False, 0.0), (18, 250, 330, False, 0.0), (18, 250, 130, True, 0.5), (32, 300, 129, False, 0.7),
    (20, 600, 64, False, 0.0), (3, 1, 1, False, 0.5), (12, 127, 200, True, 0.0), (29, 20, 63, False, 1.0),
    (18, 250, 9852, False, 0.0), (18, 249, 9852, True, 1.0), (24, 1000, 4097, False, 0.5), (32, 1000, 10000, False, 0.0),
    (1, 64, 1000, False, 0.5), (31, 255, 2000, True, 0.0),
]


@pytest.mark.parametrize("i", range(len(SMALL_C_CASES)))
def test_small_cluster_path_against_oracle_fp64(i):
    """Clusters mode (C <= 32) runs on tg_sc_softmax / tg_sc_forward / tg_sc_backward instead of the GEMM kernels."""
    C, K, V, constrained, g2 = SMALL_C_CASES[i]
    pc.small_cluster_case("cuda:0", C, K, V, constrained, g2, seed=300 + i, n=4)


def test_small_cluster_path_is_bit_reproducible():
    """No float atomics, fixed reduction orders: two runs of the clusters-mode kernels on the same inputs give the same bits,
    alone and as elements of a batch (8 folds of the tutorial's cross-validation shape, 30 epochs)."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V = 18, 250, 9852
    data = orc.make_synthetic(C, K, V, seed=12)
    M0 = orc.reference_init_M(C, V, 9)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3)
    runs = []
    for _ in range(2):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, lambdas=lam)
        h = e.new_history(30)
        e.step(30, 0.1, h)
        runs.append((e.result().cpu().numpy(), h.cpu().numpy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1], equal_nan=True)


def test_small_cluster_path_agrees_with_the_gemm_path():
    """The same clusters-mode problem on both paths (tile_size pins the GEMM kernels): 50 epochs, mappings within rounding."""
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    C, K, V = 18, 250, 3000
    data = orc.make_synthetic(C, K, V, seed=77)
    M0 = orc.reference_init_M(C, V, 5)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5)
    res = []
    for ts in (0, 128):
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision="fp32", lambdas=lam, tile_size=ts)
        h = e.new_history(50)
        e.step(50, 0.1, h)
        res.append((e.result().cpu().numpy(), h.cpu().numpy()))
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-5
    np.testing.assert_allclose(res[0][1][:, :4], res[1][1][:, :4], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("world,precision", [(2, "fp32"), (3, "bf16x3"), (4, "bf16x3")])
def test_spot_shards_on_one_gpu_match_single_engine(world, precision):
    """The multi-GPU driver (tangram_amd.sharded; the C library issues kernels + three exchanges per step) with `world` shards of one problem as
    threads on ONE GPU (tests/local_comm.py instead of RCCL): same history and mapping as the unsharded engine and the fp64
    oracle, with regularisers and a d_source prior, ragged shard widths."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from tangram_amd import _capi
    from tests.local_comm import run_ranks
    C, K, V = 400, 48, 1010
    data = orc.make_synthetic(C, K, V, seed=31)
    M0 = orc.reference_init_M(C, V, 7)
    rng = np.random.default_rng(3)
    ds = rng.random(C).astype(np.float32); ds /= ds.sum()
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3, lambda_l1=1e-4, lambda_l2=1e-5)
    n = 6

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], d_source=ds, device=DEV, precision=precision, lambdas=lam, comm=comm)
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)
        return hist.cpu().numpy(), sh.result_full().cpu().numpy(), sh.validate()    # (the rows `run` writes are the global history)

    res = run_ranks(world, rank_fn)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], d_source=ds, device=DEV, precision=precision, lambdas=lam)
    h1 = e.new_history(n)
    e.step(n, 0.1, h1)
    val1 = e.validate()
    h1, P1 = h1.cpu().numpy(), e.result().cpu().numpy()
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], d_source=ds, M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY, _capi.H_L1, _capi.H_L2]
    for hist, P, val in res:                            # every rank holds the same reduced history and the full mapping
        np.testing.assert_allclose(val, val1, rtol=5e-6, atol=2e-7)          # _val_loss_fn over all spots, all-reduced in the library
        assert val == res[0][2]
        np.testing.assert_allclose(hist[:, cols], h1[:, cols], atol=5e-6, rtol=2e-6)
        np.testing.assert_allclose(P, P1, atol=2e-6)
        np.testing.assert_allclose(hist[:, _capi.H_TOTAL], np.array(ho["total_loss"]), atol=2e-5, rtol=1e-5)
        assert np.abs(P - Po).max() < 2e-5


@pytest.mark.parametrize("world", [2, 3])
def test_spatial_terms_on_spot_shards_match_single_engine(world):
    """Neighbourhood, cell-type-island and autocorrelation terms on spot shards (the library gathers Ghat every iteration and every
    rank evaluates the terms on the whole spot graph, mapping_optimizer.py:234-263): shards as threads of one GPU (ceil partition,
    last block shorter) against the unsharded engine and the fp64 oracle."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from tangram_amd.synthetic import hex_grid_graph
    from tangram_amd import _capi
    from tests.local_comm import run_ranks
    C, K, V, T, n = 500, 40, 1001, 5, 5
    data = orc.make_synthetic(C, K, V, seed=19, n_types=T)
    M0 = orc.reference_init_M(C, V, 2)
    N, W = hex_grid_graph(V)
    Ws = orc.grid_graph(V, standardized=True, self_inclusion=False)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.3, lambda_neighborhood_g1=0.96, lambda_ct_islands=0.17,
               lambda_getis_ord=0.4, lambda_moran=0.3, lambda_geary=0.2)
    graphs = dict(voxel_weights=W, neighborhood_filter=N, ct_encode=data["ct_encode"], spatial_weights=Ws)

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam, comm=comm, **graphs)
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)
        out = hist.cpu().numpy(), sh.result_full().cpu().numpy()
        sh.release()
        return out

    res = run_ranks(world, rank_fn)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device=DEV, precision="bf16x3", lambdas=lam, **graphs)
    h1 = e.new_history(n)
    e.step(n, 0.1, h1)
    h1, P1 = h1.cpu().numpy(), e.result().cpu().numpy()
    e.release()
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, voxel_weights=W.toarray(),
                         neighborhood_filter=N.toarray(), ct_encode=data["ct_encode"], spatial_weights=Ws, **lam)
    Po, ho = o.train(n, 0.1)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_NB, _capi.H_CT, _capi.H_GETIS, _capi.H_MORAN, _capi.H_GEARY]
    for hist, P in res:
        np.testing.assert_array_equal(hist, res[0][0])                         # the same global history on every rank
        np.testing.assert_allclose(hist[:, cols], h1[:, cols], atol=5e-6, rtol=2e-6)
        np.testing.assert_allclose(P, P1, atol=2e-6)
        np.testing.assert_allclose(hist[:, _capi.H_TOTAL], np.array(ho["total_loss"]), atol=2e-5, rtol=1e-5)
        assert np.abs(P - Po).max() < 2e-4


@pytest.mark.parametrize("world", [2, 3])
def test_constrained_spot_shards_match_single_engine(world):
    """MapperConstrained on spot shards (the filter F is replicated; its gradient comes from the all-reduced row sums, the
    density prior's total from the set-up exchange): same history, mapping, filter and projection as the unsharded engine
    and the fp64 oracle."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.sharded import make_sharded
    from tangram_amd import _capi
    from tests.local_comm import run_ranks
    C, K, V = 350, 40, 777
    data = orc.make_synthetic(C, K, V, seed=17)
    M0, F0 = orc.reference_init_MF_constrained(C, V, 23)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.6, lambda_r=1e-3, lambda_count=0.7, lambda_f_reg=1.5)
    n, tc = 6, 120.0

    def rank_fn(comm):
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device=DEV, precision="bf16x3",
                          lambdas=lam, target_count=tc, comm=comm)
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)
        P, F = sh.result_full(with_filter=True)
        return hist.cpu().numpy(), P.cpu().numpy(), F.cpu().numpy(), sh.project_full().cpu().numpy()

    res = run_ranks(world, rank_fn)
    e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device=DEV, precision="bf16x3",
                        lambdas=lam, target_count=tc)
    h1 = e.new_history(n)
    e.step(n, 0.1, h1)
    P1, F1 = e.result(with_filter=True)
    h1, P1, F1, G1 = h1.cpu().numpy(), P1.cpu().numpy(), F1.cpu().numpy(), e.project().cpu().numpy()
    o = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
    Po, Fo, ho = o.train(n, 0.1)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY, _capi.H_COUNT, _capi.H_FREG]
    for hist, P, F, G in res:
        np.testing.assert_allclose(hist, res[0][0], rtol=3e-7, atol=0, equal_nan=True)    # every rank holds the same global history
        np.testing.assert_allclose(hist[:, cols], h1[:, cols], atol=5e-6, rtol=2e-6)
        np.testing.assert_allclose(P, P1, atol=2e-6)
        np.testing.assert_allclose(F, F1, atol=2e-6)
        np.testing.assert_allclose(G, G1, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(hist[:, _capi.H_TOTAL], np.array(ho["total_loss"], dtype=np.float64), atol=2e-5, rtol=1e-5)
        assert np.abs(P - Po).max() < 2e-5 and np.abs(F - Fo).max() < 2e-5


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_against_oracle(seed):
    """Seeded random problems (shape, priors, every combination of loss terms incl. the CSR spatial ones, both mapper classes)
    against the fp64 oracle, 4 iterations, fp32 and bf16x3 paths -- the combinations no fixture spells out."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    rng = np.random.default_rng(1000 + seed)
    C, K, V = int(rng.integers(1, 300)), int(rng.integers(1, 70)), int(rng.integers(2, 400))
    data = orc.make_synthetic(C, K, V, seed=seed, n_types=4)
    M0 = rng.normal(size=(C, V)).astype(np.float32)
    pick = lambda vals: float(rng.choice(vals))
    constrained = seed % 4 == 3
    n = 4
    if constrained:
        lam = dict(lambda_g1=1.0, lambda_d=pick([0.5, 1.0]), lambda_g2=pick([0.0, 0.5, 1.0]), lambda_r=pick([0.0, 1e-3]),
                   lambda_count=pick([0.5, 1.0]), lambda_f_reg=pick([0.5, 1.0]))
        F0 = rng.normal(size=(C,)).astype(np.float32)
        target = float(rng.integers(1, max(2, C)))
        mk_o = lambda: orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0, F0=F0, dtype=np.float64, target_count=target, **lam)
        mk_e = lambda p: HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device=DEV, precision=p,
                                         lambdas=lam, target_count=target)
    else:
        lam = dict(lambda_g1=1.0, lambda_d=pick([0.0, 0.5, 1.0, 2.0]), lambda_g2=pick([0.0, 0.4, 1.0]), lambda_r=pick([0.0, 1e-3, 1e-2]),
                   lambda_l1=pick([0.0, 1e-4]), lambda_l2=pick([0.0, 1e-5]), lambda_neighborhood_g1=pick([0.0, 0.96]),
                   lambda_ct_islands=pick([0.0, 0.17]), lambda_moran=pick([0.0, 0.0, 0.4]))
        kw_o, kw_e = {}, {}
        d = data["d"] if lam["lambda_d"] > 0 else None
        if d is not None and rng.random() < 0.5:
            ds = (rng.random(C) + 0.1).astype(np.float32)
            ds /= ds.sum()
            kw_o["d_source"] = kw_e["d_source"] = ds
        if lam["lambda_neighborhood_g1"] > 0:
            kw_o["voxel_weights"] = kw_e["voxel_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=True)
        if lam["lambda_ct_islands"] > 0:
            kw_o["neighborhood_filter"] = kw_e["neighborhood_filter"] = orc.grid_graph(V, standardized=False, self_inclusion=False)
            kw_o["ct_encode"] = kw_e["ct_encode"] = data["ct_encode"]
        if lam["lambda_moran"] > 0:
            kw_o["spatial_weights"] = kw_e["spatial_weights"] = orc.grid_graph(V, standardized=True, self_inclusion=False)
        mk_o = lambda: orc.OracleMapper(data["S"], data["G"], d=d, M0=M0, dtype=np.float64, **lam, **kw_o)
        mk_e = lambda p: HipMapperEngine(data["S"], data["G"], M0, d=d, device=DEV, precision=p, lambdas=lam, **kw_e)
    res = mk_o().train(n, 0.1)
    ho = np.array(res[-1]["total_loss"], dtype=np.float64)
    for prec in ("fp32", "bf16x3"):
        e = mk_e(prec)
        hist = e.new_history(n)
        e.step(n, 0.1, hist)
        h = hist[:, _capi.H_TOTAL].cpu().numpy()
        np.testing.assert_allclose(h, ho, atol=2e-5, rtol=2e-5, err_msg=f"{prec} {C}x{K}x{V} {lam}")
        out = e.result(with_filter=constrained)
        P = (out[0] if constrained else out).cpu().numpy()
        np.testing.assert_allclose(P, res[0], atol=5e-5, err_msg=f"{prec} {C}x{K}x{V} {lam}")
        if constrained:
            np.testing.assert_allclose(out[1].cpu().numpy(), res[1], atol=5e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_empty_spots_and_zero_density(precision):
    """Degenerate inputs the reference's formulas special-case: spots without any count (all-zero rows of G: the per-spot
    cosine clamps both norms at 1e-8, torch semantics) and a density prior with exact zeros (KLDivLoss: xlogy(0, 0) = 0)."""
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd import _capi
    C, K, V = 150, 30, 260
    data = orc.make_synthetic(C, K, V, seed=5)
    G = data["G"].copy()
    G[[0, 17, 255, 259]] = 0.0                                   # empty spots (first tile, tile edge, last rows)
    d = (G.sum(1) / G.sum()).astype(np.float32)                  # -> exact zeros for them
    assert (d == 0).sum() >= 4
    M0 = orc.reference_init_M(C, V, 11)
    lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=1.0, lambda_r=1e-3)
    e = HipMapperEngine(data["S"], G, M0, d=d, device=DEV, precision=precision, lambdas=lam)
    n = 6
    hist = e.new_history(n)
    e.step(n, 0.1, hist)
    o = orc.OracleMapper(data["S"], G, d=d, M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)
    h = hist.cpu().numpy()
    assert np.isfinite(h[:, [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY]]).all()
    for col, k in ((_capi.H_TOTAL, "total_loss"), (_capi.H_MAIN, "main_loss"), (_capi.H_VG, "vg_reg"), (_capi.H_KL, "kl_reg")):
        np.testing.assert_allclose(h[:, col], np.array(ho[k]), atol=1e-5, rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(e.result().cpu().numpy(), Po, atol=2e-5)


def test_project_genes_from_sparse_single_cell_matrix():
    """f-4: project_genes fed with the scipy CSR `adata_sc.X` (no toarray() on the host): bit-identical to the dense path, through
    the engine and through tangram_amd.project_genes."""
    import scipy.sparse as sp
    from tangram_amd.engine import HipMapperEngine
    from tangram_amd.synthetic import make_workload, init_logits
    C, K, V, K_all = 2000, 150, 700, 1234
    w = make_workload(C, K, V, DEV, seed=5)
    e = HipMapperEngine(w["S"], w["G"], init_logits(C, V, DEV, seed=2), d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_d=1.0))
    e.step(4, 0.1, e.new_history(4))
    rng = np.random.default_rng(0)
    dense = (rng.gamma(1.0, 2.0, size=(C, K_all)) * (rng.random((C, K_all)) < 0.1)).astype(np.float32)
    a = e.project_genes(sp.csr_matrix(dense)).cpu().numpy()
    b = e.project_genes(dense).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert a.shape == (V, K_all) and np.isfinite(a).all() and a.max() > 0


def test_device_preprocessing():
    """SURVEY 8 f-4 on the GPU: CSR gather of gene columns bit-exact, density prior and cluster sums within 1 ulp of the exactly
    rounded result (same checks as the emulated CPU test)."""
    from tests.test_preprocess import check_preprocessing
    check_preprocessing(DEV)


def test_map_cells_to_space_sparse_input_on_gpu():
    """map_cells_to_space fed with scipy-sparse AnnData matrices (training genes gathered on the device) equals the dense-input run."""
    import scipy.sparse as sp
    import tangram_amd as tg
    from tangram_amd.anndata_lite import AnnDataLite
    from tests.test_map_cells_to_space import _adatas
    for mode, kw in (("cells", {}), ("clusters", dict(cluster_label="subclass_label"))):
        ad_sc, ad_sp = _adatas(C=300, K=40, V=130, extra_genes=25)
        dense = tg.map_cells_to_space(ad_sc, ad_sp, mode=mode, device=DEV, num_epochs=8, random_state=42, verbose=False, **kw)
        ad_sc2, ad_sp2 = _adatas(C=300, K=40, V=130, extra_genes=25)
        ad_sc2 = AnnDataLite(sp.csr_matrix(ad_sc2.X), obs=ad_sc2.obs, var=ad_sc2.var, uns=ad_sc2.uns)
        ad_sp2 = AnnDataLite(sp.csr_matrix(ad_sp2.X), obs=ad_sp2.obs, var=ad_sp2.var, uns=ad_sp2.uns)
        sparse = tg.map_cells_to_space(ad_sc2, ad_sp2, mode=mode, device=DEV, num_epochs=8, random_state=42, verbose=False, **kw)
        np.testing.assert_allclose(sparse.X, dense.X, atol=2e-6, err_msg=mode)


def test_cross_val_batched_on_gpu():
    """`cross_val` (reference tangram/utils.py:503-668) in clusters mode, leave-one-out over 11 genes in batches of 4 folds per
    launch, against the same folds mapped one after the other with `map_cells_to_space(cv_train_genes=...)` and scored on the host."""
    import tangram_amd as tg
    from tests.test_map_cells_to_space import _adatas
    ad_sc, ad_sp = _adatas(C=300, K=11, V=700, seed=4)
    kw = dict(cluster_label="subclass_label", random_state=7, density_prior="rna_count_based")
    src = tg.adata_to_cluster_expression(ad_sc, "subclass_label", True, device=DEV)
    t_ref, tr_ref = [], []
    for train_genes, test_genes in tg.cv_data_gen(ad_sc, ad_sp, "loo"):
        ad_map = tg.map_cells_to_space(ad_sc, ad_sp, cv_train_genes=train_genes, mode="clusters", device=DEV, num_epochs=40, verbose=False, **kw)
        pred = ad_map.X.T.astype(np.float64) @ np.asarray(src[:, test_genes].X, dtype=np.float64)
        g = np.asarray(ad_sp[:, test_genes].X, dtype=np.float64)
        t_ref.append(float(((pred * g).sum(0) / (np.linalg.norm(pred, axis=0) * np.linalg.norm(g, axis=0))).mean()))
        tr_ref.append(float(list(ad_map.uns["training_history"]["main_loss"])[-1]))
    cv, ad_ge, df = tg.cross_val(ad_sc, ad_sp, mode="clusters", num_epochs=40, device=DEV, return_gene_pred=True, folds_per_launch=4, **kw)
    assert abs(cv["avg_train_score"] - np.mean(tr_ref)) < 1e-7            # the folds' trainings are the same bits
    np.testing.assert_allclose(ad_ge.var["test_score"].to_numpy(), t_ref, atol=5e-6)
    assert ad_ge.X.shape == (700, 11) and len(df) == 11


def test_device_initialiser_on_the_gpu():
    """tg_init_logits_normal on the hardware (`Mapper(init="device")`): any block of columns equals the same columns of the full
    plane bit for bit -- what makes a spot shard's logits independent of the number of ranks --, N(0, 1) moments over 4e6 draws,
    reproducible, and a Mapper started from it trains (the CPU suite checks the same on the emulator)."""
    from tangram_amd.device_init import device_normal
    from tangram_amd.mapping_optimizer import Mapper
    from oracle import tangram_oracle as orc
    C, V = 2000, 2003
    full = device_normal(C, V, DEV, seed=42)
    assert torch.equal(full, device_normal(C, V, DEV, seed=42))
    for lo, hi in [(0, 1000), (1000, 2003), (777, 778), (5, 1999)]:
        assert torch.equal(device_normal(C, hi - lo, DEV, seed=42, col0=lo, n_cols_total=V), full[:, lo:hi])
    x = full.double()
    assert abs(float(x.mean())) < 3e-3 and abs(float(x.std()) - 1.0) < 3e-3 and abs(float((x ** 3).mean())) < 1e-2 and abs(float((x ** 4).mean()) - 3.0) < 3e-2
    assert float(x.abs().max()) < 6.5 and bool(torch.isfinite(full).all())
    assert float((device_normal(C, V, DEV, seed=43) != full).float().mean()) > 0.99
    data = orc.make_synthetic(500, 60, 300, seed=1)
    kw = dict(d=data["d"], lambda_d=1, lambda_g1=1, device=DEV)
    P1, h1 = Mapper(data["S"], data["G"], random_state=9, init="device", **kw).train(8, print_each=None)
    P2, _ = Mapper(data["S"], data["G"], random_state=9, init="device", **kw).train(8, print_each=None)
    assert np.array_equal(P1, P2) and h1["main_loss"][-1] > h1["main_loss"][0]


def test_adam_square_root_and_divisions_against_ieee():
    """The update kernels evaluate Adam's `sqrt(v) / bias_correction2_sqrt + eps` and `m / denom` (torch `_single_tensor_adam`) with
    cheap sequences instead of hipcc's IEEE ones (tg_device.h, round 5): `tg_sqrt_cr` (v_rsq_f32 + a Newton step with fma residuals)
    and `tg_div_by` (multiplication by the correctly rounded reciprocal + one residual correction) are held to the correctly
    rounded IEEE results over the range Adam's second moment lives in: never more than 1 ulp away and equal in all but < 1e-4 of the
    cases; `tg_div_fr` (v_rcp_f32 + one residual correction) to the IEEE quotient within 1 ulp, equal in all but < 2 % of the cases
    (the measured fractions are written to gpurun_out/adam_math_vs_ieee.json).  Zero, denormal and infinite arguments of the
    square root come back as they are (sqrt of a denormal is <= 1.1e-19: it vanishes against eps in the denominator)."""
    import ctypes as ct
    from tangram_amd import _capi
    lib = _capi.lib()
    lib.tg_debug_adam_math.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_float, ct.c_void_p, ct.c_longlong, ct.c_void_p]
    lib.tg_debug_adam_math.restype = ct.c_int
    rng = np.random.default_rng(7)
    n = 1 << 21
    # second moments: squares of gradients over 30 decades, plus the special values; first moments / denominators likewise
    a = np.exp(rng.uniform(np.log(1e-36), np.log(1e30), n)).astype(np.float32)
    a[:8] = [0.0, 1.0, 4.0, 2.0, np.float32(1e-45), np.float32(1.1754944e-38), np.inf, np.float32(3.4e38)]
    b = (np.exp(rng.uniform(np.log(1e-8), np.log(1e6), n)) ).astype(np.float32)
    signed = a * np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
    bc = np.float32(np.sqrt(1.0 - 0.999 ** 37))
    out = torch.empty(3 * n, dtype=torch.float32, device=DEV)

    def run(x):
        xa, xb = torch.as_tensor(x, device=DEV), torch.as_tensor(b, device=DEV)
        assert lib.tg_debug_adam_math(xa.data_ptr(), xb.data_ptr(), ct.c_float(float(bc)), out.data_ptr(), n, None) == 0, lib.tg_last_error()
        torch.cuda.synchronize()
        return out.cpu().numpy().reshape(3, n)

    def ulps(got, ref):
        return np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))

    r = run(a)
    normal = (a >= np.float32(1.1754944e-38)) & np.isfinite(a)
    with np.errstate(all="ignore"):
        ref_sqrt = np.sqrt(a)                                   # IEEE, correctly rounded
    u_sqrt = ulps(r[0][normal], ref_sqrt[normal])
    assert u_sqrt.max() <= 1 and (u_sqrt != 0).mean() < 1e-4, (int(u_sqrt.max()), float((u_sqrt != 0).mean()))
    assert r[0][0] == 0.0 and r[0][4] == a[4] and np.isinf(r[0][6])      # zero / denormal / inf pass through
    ref_by = (a.astype(np.float32) / bc).astype(np.float32)
    fin = np.isfinite(ref_by) & normal
    u_by = ulps(r[2][fin], ref_by[fin])
    assert u_by.max() <= 1 and (u_by != 0).mean() < 1e-4, (int(u_by.max()), float((u_by != 0).mean()))
    r2 = run(signed)
    with np.errstate(all="ignore"):
        ref_div = (signed / b).astype(np.float32)
    ok = np.isfinite(ref_div) & (np.abs(ref_div) >= np.float32(1.1754944e-38)) & normal
    ulp = ulps(r2[1][ok], ref_div[ok])
    assert ulp.max() <= 1, int(ulp.max())
    assert (ulp != 0).mean() < 0.02, float((ulp != 0).mean())
    import json
    import os
    rec = dict(n=int(n), sqrt_fraction_1ulp=float((u_sqrt != 0).mean()), div_by_fraction_1ulp=float((u_by != 0).mean()),
               div_fr_fraction_1ulp=float((ulp != 0).mean()))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(rec, open(os.path.join(out_dir, "adam_math_vs_ieee.json"), "w"))
    print("Adam math vs IEEE:", rec)
