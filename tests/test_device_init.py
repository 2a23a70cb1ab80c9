"""The opt-in device-side initialiser (`Mapper(..., init="device")`, tg_init_logits_normal) and the per-rank result path
(`gather_result=False`): what lets BASELINE config 4 -- 200 000 x 50 000 logits, 40 GB -- go through the drop-in seam without the
cells x spots plane ever existing on a host (the reference draws it with np.random.normal, mapping_optimizer.py:147-157).
CPU suite: emulated kernels; world-size-2 gloo for the sharded seam."""
import os
import socket
import tracemalloc

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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


def test_device_normal_is_a_function_of_seed_and_global_index(sim):
    from tangram_amd.device_init import device_normal
    C, V = 37, 211
    full = device_normal(C, V, "cpu", seed=42).numpy()
    assert full.shape == (C, V) and np.isfinite(full).all()
    np.testing.assert_array_equal(full, device_normal(C, V, "cpu", seed=42).numpy())            # reproducible
    for lo, hi in [(0, 100), (100, 211), (53, 54), (7, 200)]:                                      # any block of columns: the same logits
        blk = device_normal(C, hi - lo, "cpu", seed=42, col0=lo, n_cols_total=V).numpy()
        np.testing.assert_array_equal(blk, full[:, lo:hi])
    other = device_normal(C, V, "cpu", seed=43).numpy()
    assert (other != full).mean() > 0.99
    assert (device_normal(C, V, "cpu", seed=42, stream_id=1).numpy() != full).mean() > 0.99        # the filter's draw is another stream
    big = device_normal(400, 500, "cpu", seed=7).numpy().astype(np.float64)                       # 2e5 draws: N(0, 1) moments
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1.0) < 0.01 and abs((big ** 3).mean()) < 0.03 and abs((big ** 4).mean() - 3.0) < 0.08
    assert np.abs(big).max() < 6.0


def test_device_normal_rejects_bad_blocks(sim):
    from tangram_amd.device_init import device_normal
    with pytest.raises(ValueError):
        device_normal(4, 10, "cpu", seed=1, col0=5, n_cols_total=12)


def test_mapper_init_device_trains_and_is_seed_reproducible(sim):
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper, MapperConstrained
    C, K, V = 60, 24, 90
    data = orc.make_synthetic(C, K, V, seed=4)
    kw = dict(d=data["d"], lambda_d=1, lambda_g1=1, device="cpu", gemm_precision="fp32")
    P1, h1 = Mapper(data["S"], data["G"], random_state=11, init="device", **kw).train(5, print_each=None)
    P2, h2 = Mapper(data["S"], data["G"], random_state=11, init="device", **kw).train(5, print_each=None)
    P3, _ = Mapper(data["S"], data["G"], random_state=12, init="device", **kw).train(5, print_each=None)
    np.testing.assert_array_equal(P1, P2)
    assert np.abs(P1 - P3).max() > 1e-4
    assert h1["main_loss"][-1] > h1["main_loss"][0]
    np.testing.assert_allclose(P1.sum(axis=1), 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        Mapper(data["S"], data["G"], init="philox", **kw)
    Pc, Fc, hc = MapperConstrained(data["S"], data["G"], data["d"], device="cpu", gemm_precision="fp32", random_state=5, init="device",
                                   target_count=30).train(4, print_each=None)
    assert np.isfinite(Pc).all() and Fc.shape == (C,) and 0.0 < Fc.min() and Fc.max() < 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SHAPE = (700, 40, 900)          # the plane is 2.52 MB of fp32: small for the emulator, large against everything else a rank allocates


def _worker(rank, world, port, sim_path, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        from oracle import tangram_oracle as orc
        from tangram_amd.mapping_optimizer import Mapper
        C, K, V = SHAPE
        data = orc.make_synthetic(C, K, V, seed=9)
        # NumPy registers its buffers with tracemalloc; torch CPU tensors -- the emulator's stand-in for DEVICE memory -- are not
        # traced.  So the traced peak is what the seam allocates on the HOST.
        tracemalloc.start()
        m = Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device="cpu", gemm_precision="fp32", random_state=33,
                   distributed=True, init="device", gather_result=False)
        P_local, hist = m.train(4, print_each=None)
        _, peak = tracemalloc.get_traced_memory()
        tracemalloc.stop()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), P=P_local, lo=m.spot_range[0], hi=m.spot_range[1], peak=peak,
                 main=np.array(hist["main_loss"]))
    finally:
        dist.destroy_process_group()


def test_sharded_device_init_never_holds_the_plane_on_the_host(sim, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), sim, str(tmp_path)), nprocs=2, join=True)
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper
    C, K, V = SHAPE
    data = orc.make_synthetic(C, K, V, seed=9)
    P, hist = Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device="cpu", gemm_precision="fp32", random_state=33,
                     init="device").train(4, print_each=None)
    plane = C * V * 4
    covered = 0
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        assert z["P"].shape == (C, hi - lo)                               # only this rank's spots come back
        # the same logits as the unsharded run (device generator: any partition), the same trajectory up to summation order
        np.testing.assert_allclose(z["P"], P[:, lo:hi], atol=2e-6)
        np.testing.assert_allclose(z["main"], np.array(hist["main_loss"]), atol=2e-6)
        # host memory: this rank's block of the result (plane / 2) + inputs, never the plane (init="reference" would draw all of
        # it in float64 on every rank, and the gathered result would be another full plane)
        assert int(z["peak"]) < 0.8 * plane, (int(z["peak"]), plane)
        covered += hi - lo
    assert covered == V


def _worker_c(rank, world, port, sim_path, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        from oracle import tangram_oracle as orc
        from tangram_amd.mapping_optimizer import Mapper, MapperConstrained
        C, K, V = 160, 40, 210
        data = orc.make_synthetic(C, K, V, seed=12)
        # MapperConstrained on shards from the device generator (M block by block, the replicated filter F from its own stream),
        # local result; and the two-product path (count data: bf16-exact S) on shards
        mc = MapperConstrained(data["S"], data["G"], data["d"], device="cpu", gemm_precision="bf16x3", random_state=21, target_count=90,
                               distributed=True, init="device", gather_result=False, s_exact="auto")
        Pc, Fc, hc = mc.train(4, print_each=None)
        m = Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device="cpu", gemm_precision="bf16x3", random_state=21,
                   distributed=True, init="device", s_exact="auto")
        P, h = m.train(4, print_each=None)
        np.savez(os.path.join(outdir, f"c{rank}.npz"), Pc=Pc, Fc=Fc, lo=mc.spot_range[0], hi=mc.spot_range[1], P=P,
                 main=np.array(h["main_loss"]), eff=np.array([m._engine.effective_precision.startswith("bf16x3 (S exact"),
                                                              mc._engine.effective_precision.startswith("bf16x3 (S exact")]))
    finally:
        dist.destroy_process_group()


def test_sharded_constrained_device_init_and_two_product_path(sim, tmp_path):
    mp.spawn(_worker_c, args=(2, _free_port(), sim, str(tmp_path)), nprocs=2, join=True)
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper, MapperConstrained
    C, K, V = 160, 40, 210
    data = orc.make_synthetic(C, K, V, seed=12)
    Pc, Fc, _ = MapperConstrained(data["S"], data["G"], data["d"], device="cpu", gemm_precision="bf16x3", random_state=21, target_count=90,
                                  init="device").train(4, print_each=None)                      # one process, general path
    P, h = Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device="cpu", gemm_precision="bf16x3", random_state=21,
                  init="device").train(4, print_each=None)
    z0, z1 = np.load(tmp_path / "c0.npz"), np.load(tmp_path / "c1.npz")
    assert z0["eff"].all() and z1["eff"].all()                       # every rank found S exact and took the two-product path
    np.testing.assert_array_equal(z0["P"], z1["P"])                  # gathered result: identical on both ranks
    np.testing.assert_allclose(z0["P"], P, atol=2e-6)                # = the unsharded general-path run up to summation order
    np.testing.assert_allclose(z0["main"], np.array(h["main_loss"]), atol=2e-6)
    for z in (z0, z1):
        lo, hi = int(z["lo"]), int(z["hi"])
        np.testing.assert_allclose(z["Pc"], Pc[:, lo:hi], atol=2e-6)
        np.testing.assert_allclose(z["Fc"], Fc, atol=2e-6)
    assert int(z0["hi"]) == int(z1["lo"]) and int(z1["hi"]) == V
