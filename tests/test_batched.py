"""Batched independent mappings (tg_batch): B mappings of one shape advance in one launch per kernel
(blockIdx.z = mapping).  CPU: through the emulated C ABI -- every batch element against the fp64 oracle and bit-identical to the
same mapping trained on its own.  The GPU version of the same check is tests/test_gpu_parity.py::test_batched_mappings."""
import numpy as np
import pytest

from tests.hipsim.build_sim import build_sim


@pytest.fixture(scope="module")
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


def check_batched(device, precision, C, K, V, B, epochs, lam, tol_loss, tol_P):
    """Leave-one-gene-out folds like cross_val (utils.py:576-600): fold i trains on all genes but gene i, own seed."""
    import tangram_amd as tg
    import tangram_amd.mapping_optimizer as mo
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(C, K, V, seed=9)
    ds = np.full(C, 1.0 / C, np.float32)

    def fold(i):
        keep = [g for g in range(K) if g != i]
        return dict(S=data["S"][:, keep], G=data["G"][:, keep], d=data["d"], d_source=ds, **lam), i + 1

    def builder(i):
        kw, seed = fold(i)
        return lambda: mo.Mapper(device=device, random_state=seed, gemm_precision=precision, **kw)

    res, mappers = tg.train_many([builder(i) for i in range(B)], epochs, 0.1, device=device)
    solo = [builder(i)().train(num_epochs=epochs, learning_rate=0.1, print_each=None) for i in range(B)]
    for i in range(B):
        P, hist = res[i]
        np.testing.assert_array_equal(P, solo[i][0], err_msg=f"fold {i}: batch != the mapping trained alone")
        for k in ("total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"):
            np.testing.assert_array_equal(np.array(hist[k], dtype=np.float64), np.array(solo[i][1][k], dtype=np.float64), err_msg=k)
        kw, seed = fold(i)
        o = orc.OracleMapper(M0=orc.reference_init_M(C, V, seed), dtype=np.float64, **kw)
        Po, ho = o.train(epochs, 0.1)
        for k in ("total_loss", "main_loss", "kl_reg"):
            ref = np.array(ho[k], dtype=np.float64)
            err = np.abs(np.array([float(x) for x in hist[k]]) - ref).max()
            assert err <= tol_loss * max(1.0, np.abs(ref).max()), (i, k, err)      # (lambda_r * entropy puts total_loss at ~30)
        assert np.abs(P - Po).max() <= tol_P, i
    # the handles remain usable on their own afterwards (same step count everywhere)
    assert len({m._engine.logits()[3] for m in mappers}) == 1


def check_batched_constrained(device, precision, C, K, V, B, epochs, tol_loss, tol_P):
    """B MapperConstrained folds in one tg_batch (cross_val passes any `mode`, utils.py:576-600): every batch element bit-identical
    to the same mapping trained alone, and against the fp64 oracle (mapping, filter, history incl. count / f_reg terms)."""
    import tangram_amd as tg
    import tangram_amd.mapping_optimizer as mo
    from tangram_amd.batched import _batch_key
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(C, K, V, seed=4)
    lam = dict(lambda_d=1, lambda_g1=1, lambda_g2=0.5, lambda_r=1e-3, lambda_count=0.8, lambda_f_reg=1.3)
    tc = 0.3 * V

    def fold(i):
        keep = [g for g in range(K) if g != i]
        return dict(S=data["S"][:, keep], G=data["G"][:, keep], d=data["d"], target_count=tc, **lam), i + 3

    def builder(i):
        kw, seed = fold(i)
        return lambda: mo.MapperConstrained(device=device, random_state=seed, gemm_precision=precision, **kw)

    res, mappers = tg.train_many([builder(i) for i in range(B)], epochs, 0.1, device=device)
    assert len({_batch_key(m) for m in mappers}) == 1 and _batch_key(mappers[0]) is not None      # they DID share one tg_batch
    solo = [builder(i)().train(num_epochs=epochs, learning_rate=0.1, print_each=None) for i in range(B)]
    keys = ("total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg", "count_reg", "lambda_f_reg")
    for i in range(B):
        P, F, hist = res[i]
        np.testing.assert_array_equal(P, solo[i][0], err_msg=f"fold {i}: batch != the mapping trained alone")
        np.testing.assert_array_equal(F, solo[i][1], err_msg=f"fold {i}: filter")
        for k in keys:
            assert hist[k] == solo[i][2][k], (i, k)
        kw, seed = fold(i)
        M0, F0 = orc.reference_init_MF_constrained(C, V, seed)
        o = orc.OracleMapperConstrained(kw["S"], kw["G"], kw["d"], M0=M0, F0=F0, target_count=tc, dtype=np.float64, **lam)
        Po, Fo, ho = o.train(epochs, 0.1)
        for k in keys:
            ref = np.array([float(x) for x in ho[k]], dtype=np.float64)
            err = np.abs(np.array([float(x) for x in hist[k]]) - ref).max()
            assert err <= tol_loss * max(1.0, np.abs(ref).max()), (i, k, err)
        assert np.abs(P - Po).max() <= tol_P and np.abs(F - Fo).max() <= tol_P, i
    assert len({m._engine.logits()[3] for m in mappers}) == 1


def test_batched_constrained_folds_emulated(sim):
    check_batched_constrained("cpu", "fp32", C=14, K=9, V=70, B=3, epochs=5, tol_loss=1e-5, tol_P=2e-5)


def check_batched_tuning_seeds(device, precision, C, K, V, epochs, val_each):
    """The reference's tuning caller (mapping_parameter_tuning.py:110-129): THREE seeds of one problem, train / validation gene
    split, `val_each=1` -- trained in ONE tg_batch that pauses at every validation epoch, against the same three mappings trained
    alone: all nine history keys and the mapping, bit for bit."""
    import tangram_amd as tg
    import tangram_amd.mapping_optimizer as mo
    from tangram_amd.batched import _batch_key
    from oracle import tangram_oracle as orc
    data = orc.make_synthetic(C, K, V, seed=21)
    tr = np.arange(0, K - 3)
    va = np.arange(K - 3, K)
    kw = dict(S=data["S"], G=data["G"], d=data["d"], train_genes_idx=tr, val_genes_idx=va, lambda_d=1, lambda_g1=1, lambda_g2=0.5)

    def builder(seed):
        return lambda: mo.Mapper(device=device, random_state=seed, gemm_precision=precision, **kw)

    seeds = (1, 2, 3)                      # (the reference's seeds are 0, 1, 2; seed 0 means "unseeded" there, :148 -- not comparable)
    res, mappers = tg.train_many([builder(s) for s in seeds], epochs, 0.1, device=device, val_each=val_each)
    assert len({_batch_key(m) for m in mappers}) == 1 and _batch_key(mappers[0]) is not None
    assert len({m._engine.logits()[3] for m in mappers}) == 1                # they were stepped together
    solo = [builder(s)().train(num_epochs=epochs, learning_rate=0.1, print_each=None, val_each=val_each) for s in seeds]
    n_val = sum(1 for t in range(1, epochs + 1) if (t - 1) % val_each == 0)
    keys = ("total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg", "val_total_loss", "val_gene_sim", "val_sp_sparsity_weighted_sim", "val_entropy")
    for i in range(3):
        P, hist = res[i]
        np.testing.assert_array_equal(P, solo[i][0], err_msg=f"seed {seeds[i]}: batch != the mapping trained alone")
        assert set(hist) == set(solo[i][1]) == set(keys)
        for k in keys:
            a, b = np.array(hist[k], dtype=np.float64), np.array(solo[i][1][k], dtype=np.float64)
            assert len(a) == len(b) == (n_val if k.startswith("val_") else epochs), k
            np.testing.assert_array_equal(a, b, err_msg=f"seed {seeds[i]}: {k}")
        assert np.isfinite(np.array(hist["val_gene_sim"], dtype=np.float64)).all()


def test_batched_tuning_seeds_with_val_each_emulated(sim):
    check_batched_tuning_seeds("cpu", "fp32", C=14, K=12, V=70, epochs=5, val_each=1)
    check_batched_tuning_seeds("cpu", "bf16x3", C=10, K=9, V=40, epochs=7, val_each=3)


def test_batched_folds_emulated(sim):
    check_batched("cpu", "fp32", C=14, K=9, V=70, B=3, epochs=5, lam=dict(lambda_d=1, lambda_g1=1, lambda_g2=0.5), tol_loss=1e-5, tol_P=2e-5)
    check_batched("cpu", "bf16x3", C=10, K=6, V=40, B=2, epochs=3, lam=dict(lambda_d=1, lambda_g1=1, lambda_r=1e-3, lambda_l2=1e-5),
                  tol_loss=1e-5, tol_P=2e-5)


def test_batched_folds_in_groups_emulated(sim):
    """From 8 mappings on tg_batch steps 2 - 4 groups of mappings (on streams of their own on the GPU): the per-group argument
    offsets, 9 Mapper folds as 4 + 5 and 13 MapperConstrained folds as 4 + 4 + 5."""
    check_batched("cpu", "fp32", C=9, K=12, V=66, B=9, epochs=4, lam=dict(lambda_d=1, lambda_g1=1, lambda_g2=0.5), tol_loss=1e-5, tol_P=2e-5)
    check_batched_constrained("cpu", "fp32", C=40, K=16, V=50, B=13, epochs=3, tol_loss=1e-5, tol_P=2e-5)


def test_batch_rejects_mixed_shapes(sim):
    import ctypes as ct
    import tangram_amd.mapping_optimizer as mo
    from tangram_amd.batched import MapperBatch
    from oracle import tangram_oracle as orc
    a = orc.make_synthetic(10, 6, 40, seed=1)
    b = orc.make_synthetic(10, 7, 40, seed=1)
    m1 = mo.Mapper(a["S"], a["G"], d=a["d"], lambda_d=1, device="cpu", random_state=1, gemm_precision="fp32")
    m2 = mo.Mapper(b["S"], b["G"], d=b["d"], lambda_d=1, device="cpu", random_state=1, gemm_precision="fp32")
    with pytest.raises(ValueError, match="ONE shape"):
        MapperBatch([m1, m2])
    with pytest.raises(ValueError, match="twice"):
        MapperBatch([m1, m1])


def test_mappings_that_fill_the_gpu_are_not_batched():
    """train_many's grouping key: above 2^25 cells x spots a batch measured slower than one mapping after the other."""
    from types import SimpleNamespace
    from tangram_amd import batched

    class Mapper:                                    # (the key looks at the class NAME and the engine's geometry only)
        def __init__(self, C, V):
            cfg = SimpleNamespace(lambda_neighborhood_g1=0, lambda_ct_islands=0, lambda_getis_ord=0, lambda_moran=0, lambda_geary=0,
                                  pipeline_bands=0, lambda_r=0, lambda_l1=0, lambda_l2=0, beta1=0.9, beta2=0.999, tile_size=0, fwd_splits=0)
            self._engine = SimpleNamespace(C=C, K=249, V=V, cfg=cfg, precision="bf16x3", device="cuda:0", _torch_stream=None)
            self._sharded = None

    assert batched._batch_key(Mapper(18, 9852)) is not None and batched._batch_key(Mapper(4200, 1100)) is not None
    assert batched._batch_key(Mapper(8000, 4000)) is not None                     # 32.0 M <= 2^25
    assert batched._batch_key(Mapper(26431, 9852)) is None and batched._batch_key(Mapper(30000, 10000)) is None
