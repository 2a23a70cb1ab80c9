"""world_size-2 test of the spot-sharded multi-GPU driver on CPU: gloo collectives + the emulated C ABI.
Checks that 2 shards reproduce the 1-shard run and the oracle (reduction-order tolerance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.hipsim.build_sim import build_sim


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sim_path, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        from tangram_amd.sharded import make_sharded
        from oracle import tangram_oracle as orc
        C, K, V = 90, 30, 150            # 150 spots -> 75 per rank, ragged against the 128 tile
        data = orc.make_synthetic(C, K, V, seed=21)
        M0 = orc.reference_init_M(C, V, 5)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3)
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=lam)
        n = 4
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)
        hist = sh.finalize_history(hist)
        P = sh.result_full()
        if rank == 0:
            np.savez(os.path.join(outdir, "sharded.npz"), P=P.numpy(), hist=hist.numpy())
    finally:
        dist.destroy_process_group()


def test_two_shards_match_single_and_oracle(tmp_path):
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, sim_path, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "sharded.npz")

    from tangram_amd import _capi
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    _capi._install_library_for_tests(sim_path)
    try:
        C, K, V = 90, 30, 150
        data = orc.make_synthetic(C, K, V, seed=21)
        M0 = orc.reference_init_M(C, V, 5)
        lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5, lambda_r=1e-3)
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=lam)
        n = 4
        h1 = e.new_history(n)
        e.step(n, 0.1, h1)
        P1 = e.result().numpy()
    finally:
        _capi._install_library_for_tests(None)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **lam)
    Po, ho = o.train(n, 0.1)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY]
    np.testing.assert_allclose(z["hist"][:, cols], h1.numpy()[:, cols], atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(z["P"], P1, atol=1e-6)
    for j, k in zip(cols, ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"]):
        np.testing.assert_allclose(z["hist"][:, j], np.array(ho[k]), atol=1e-5, rtol=1e-5, err_msg=k)
    assert np.abs(z["P"] - Po).max() < 1e-5
