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


LAM_C = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.7, lambda_r=1e-3, lambda_count=0.5, lambda_f_reg=2.0)


def _worker(rank, world, port, sim_path, outdir, transport="callbacks"):
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
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=lam, transport=transport)
        assert sh.transport == ("peer" if transport == "peer_checked" else transport)
        n = 4
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)               # ONE call of the C library: kernels + the three exchanges per step (gloo through callbacks)
        P = sh.result_full()
        # MapperConstrained on shards: the filter F is replicated, its gradient comes from the all-reduced row sums
        M0c, F0c = orc.reference_init_MF_constrained(C, V, 5)
        shc = make_sharded(data["S"], data["G"], M0c, d=data["d"], F0=F0c, mode="constrained", device="cpu", precision="fp32",
                           lambdas=LAM_C, target_count=40.0, transport=transport)
        hc = shc.eng.new_history(n)
        shc.run(n, 0.1, hc)
        Pc, Fc = shc.result_full(with_filter=True)
        Gc = shc.project_full()
        sh.peer_check(); shc.peer_check()      # (peer transport: no exchange ever gave up waiting)
        np.savez(os.path.join(outdir, f"sharded_{rank}.npz"), P=P.numpy(), hist=hist.numpy(), Pc=Pc.numpy(), Fc=Fc.numpy(),
                 hc=hc.numpy(), Gc=Gc.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["callbacks", "peer", "peer_checked"])
def test_two_shards_match_single_and_oracle(tmp_path, transport):
    """transport "callbacks": gloo collectives called back from the C step; "peer": the library's own one-hop exchange kernels over
    mailboxes the two processes map from each other (emulated build: POSIX shared memory stands in for hipIpc device memory);
    "peer_checked": the same after the set-up's self-test against the group's own collectives (what "auto" does on an nccl group)."""
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, sim_path, str(tmp_path), transport), nprocs=2, join=True)
    z = np.load(tmp_path / "sharded_0.npz")
    z1 = np.load(tmp_path / "sharded_1.npz")
    for k in z.files:                                    # every rank holds the same global history, mapping, filter, projection
        np.testing.assert_array_equal(z[k], z1[k], err_msg=k)

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
    # constrained
    M0c, F0c = orc.reference_init_MF_constrained(C, V, 5)
    oc = orc.OracleMapperConstrained(data["S"], data["G"], data["d"], M0=M0c, F0=F0c, target_count=40.0, dtype=np.float64, **LAM_C)
    Pco, Fco, hco = oc.train(n, 0.1)
    colsc = {_capi.H_TOTAL: "total_loss", _capi.H_MAIN: "main_loss", _capi.H_VG: "vg_reg", _capi.H_KL: "kl_reg",
             _capi.H_ENTROPY: "entropy_reg", _capi.H_COUNT: "count_reg", _capi.H_FREG: "lambda_f_reg"}
    for j, k in colsc.items():
        np.testing.assert_allclose(z["hc"][:, j], np.array(hco[k], dtype=np.float64), atol=2e-5, rtol=1e-5, err_msg="constrained " + k)
    assert np.abs(z["Pc"] - Pco).max() < 1e-5 and np.abs(z["Fc"] - Fco).max() < 1e-5
    f = Fco[:, None]
    ref = (Pco * f).T @ data["S"].astype(np.float64)
    assert np.linalg.norm(z["Gc"] - ref) / np.linalg.norm(ref) < 1e-5


def _val_each_run(device, distributed=False):
    """`Mapper(...).train(val_each=2)` (the tuning path, mapping_parameter_tuning.py:110-129); on spot shards the
    validation metrics are sums over ALL spots (tg_mapper_validate all-reduces them)."""
    import tangram_amd.mapping_optimizer as mo
    from oracle import tangram_oracle as orc
    C, K, V = 60, 16, 140
    data = orc.make_synthetic(C, K, V, seed=8)
    m = mo.Mapper(S=data["S"], G=data["G"], d=data["d"], lambda_d=1, lambda_g1=1, lambda_g2=0.5, device=device, random_state=7,
                  gemm_precision="fp32", distributed=distributed)
    P, hist = m.train(num_epochs=5, learning_rate=0.1, print_each=None, val_each=2)
    out = {"val_P": P}
    for k in hist:
        if k.startswith("val_"):
            out["valhist_" + k] = np.array([float(x) for x in hist[k]])
    return out


def _seam_worker(rank, world, port, sim_path, outdir):
    """The drop-in surface under a process group: the SAME call on every rank (what a user runs under torchrun)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        import tangram_amd as tg
        from tests.test_map_cells_to_space import _adatas
        out = {}
        for mode, kw in (("cells", {}), ("clusters", dict(cluster_label="subclass_label")), ("constrained", dict(target_count=9))):
            ad_sc, ad_sp = _adatas(C=70, K=14, V=150)
            ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode=mode, device="cpu", num_epochs=5, random_state=42, verbose=False,
                                           gemm_precision="fp32", lambda_g2=0.4, distributed=True, **kw)
            out[mode + "_X"] = ad_map.X
            out[mode + "_score"] = ad_map.uns["train_genes_df"]["train_score"].sort_index().to_numpy()
            out[mode + "_loss"] = np.array([float(x) for x in ad_map.uns["training_history"]["total_loss"]])
            if mode == "constrained":
                out["constrained_F"] = np.asarray(ad_map.obs["F_out"])
        out.update(_val_each_run("cpu", distributed=True))
        # sharding is opt-in: under the same process group, a mapper built WITHOUT distributed=True stays on its own device and
        # issues no collective (rank 1 trains a different problem than rank 0 here: a hang or a mismatch would show)
        import tangram_amd.mapping_optimizer as mo
        from oracle import tangram_oracle as orc
        own = orc.make_synthetic(40 + 7 * rank, 10, 90 + 5 * rank, seed=rank)
        m_own = mo.Mapper(S=own["S"], G=own["G"], d=own["d"], lambda_d=1, device="cpu", random_state=3 + rank, gemm_precision="fp32")
        assert m_own._sharded is None
        out["own_P"], _ = m_own.train(num_epochs=2, print_each=None)
        # an UNSEEDED sharded constrained run: every rank must slice the same draw of M and F (the seed comes from rank 0)
        ad_sc, ad_sp = _adatas(C=70, K=14, V=150)
        for rs in (None, 0):
            ad_u = tg.map_cells_to_space(ad_sc, ad_sp, mode="constrained", device="cpu", num_epochs=4, random_state=rs, verbose=False,
                                         gemm_precision="fp32", target_count=9, distributed=True)
            out[f"unseeded_{rs}_X"] = ad_u.X
            out[f"unseeded_{rs}_F"] = np.asarray(ad_u.obs["F_out"])
            out[f"unseeded_{rs}_loss"] = np.array([float(x) for x in ad_u.uns["training_history"]["total_loss"]])
        # ranks that disagree about the problem are told so (on every rank) instead of hanging in a collective
        bad = orc.make_synthetic(30, 8, 60 + rank, seed=1)
        try:
            mo.Mapper(S=bad["S"], G=bad["G"], d=bad["d"], lambda_d=1, device="cpu", random_state=1, gemm_precision="fp32", distributed=True)
            out["mismatch_raised"] = np.array(0)
        except ValueError as e:
            out["mismatch_raised"] = np.array(int("different problem" in str(e)))
        np.savez(os.path.join(outdir, f"seam_{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_map_cells_to_space_shards_over_the_process_group(tmp_path):
    """`map_cells_to_space(..., distributed=True)` / `Mapper` / `MapperConstrained` under an initialised process group of 2 ranks
    shard the spots (reference seam mapping_utils.py:355-389 carries only `device`): every rank gets the full mapping, equal to
    the single-process run of the same call.  Also: opt-in only, unseeded runs, and ranks that disagree about the problem."""
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    mp.spawn(_seam_worker, args=(2, _free_port(), sim_path, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "seam_0.npz"), np.load(tmp_path / "seam_1.npz")
    from tangram_amd import _capi
    import tangram_amd as tg
    from tests.test_map_cells_to_space import _adatas
    _capi._install_library_for_tests(sim_path)
    try:
        for mode, kw in (("cells", {}), ("clusters", dict(cluster_label="subclass_label")), ("constrained", dict(target_count=9))):
            ad_sc, ad_sp = _adatas(C=70, K=14, V=150)
            ad_map = tg.map_cells_to_space(ad_sc, ad_sp, mode=mode, device="cpu", num_epochs=5, random_state=42, verbose=False,
                                           gemm_precision="fp32", lambda_g2=0.4, **kw)
            for z in (z0, z1):
                np.testing.assert_allclose(z[mode + "_X"], ad_map.X, atol=2e-6, err_msg=mode)
                np.testing.assert_allclose(z[mode + "_loss"], [float(x) for x in ad_map.uns["training_history"]["total_loss"]],
                                           atol=1e-5, rtol=1e-6, err_msg=mode)
                np.testing.assert_allclose(z[mode + "_score"], ad_map.uns["train_genes_df"]["train_score"].sort_index().to_numpy(),
                                           atol=1e-5, err_msg=mode)
            np.testing.assert_array_equal(z0[mode + "_X"], z1[mode + "_X"])
            if mode == "constrained":
                np.testing.assert_allclose(z0["constrained_F"], np.asarray(ad_map.obs["F_out"]), atol=2e-6)
        single = _val_each_run("cpu")                 # val_each through the Mapper seam: sharded == single process
        keys = [k for k in single if k.startswith("valhist_")]
        assert len(keys) == 4 and all(len(single[k]) == 3 for k in keys)      # epochs 0, 2, 4
        for z in (z0, z1):
            np.testing.assert_allclose(z["val_P"], single["val_P"], atol=2e-6)
            for k in keys:
                np.testing.assert_allclose(z[k], single[k], rtol=5e-6, atol=1e-7, err_msg=k)
        for k in keys:
            np.testing.assert_array_equal(z0[k], z1[k])
        assert z0["own_P"].shape == (40, 90) and z1["own_P"].shape == (47, 95)        # opt-in: independent per-rank mappers
        assert int(z0["mismatch_raised"]) == 1 and int(z1["mismatch_raised"]) == 1
        for rs in ("None", "0"):                                                       # unseeded sharded run: one consistent mapping
            np.testing.assert_array_equal(z0[f"unseeded_{rs}_X"], z1[f"unseeded_{rs}_X"])
            np.testing.assert_array_equal(z0[f"unseeded_{rs}_F"], z1[f"unseeded_{rs}_F"])
            np.testing.assert_array_equal(z0[f"unseeded_{rs}_loss"], z1[f"unseeded_{rs}_loss"])
            np.testing.assert_allclose(z0[f"unseeded_{rs}_X"].sum(axis=1), 1.0, atol=1e-5)
            assert np.isfinite(z0[f"unseeded_{rs}_loss"]).all()
    finally:
        _capi._install_library_for_tests(None)


# ------------------------------------------------------------------------------------------------------------------------
# spatial terms on spot shards: every rank gathers Ghat and evaluates the terms on the whole spot graph
# ------------------------------------------------------------------------------------------------------------------------
SP_SHAPE = (60, 18, 131)           # 131 spots over 2 ranks: blocks of 66 and 65 (ceil partition)
SP_LAM = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.4, lambda_neighborhood_g1=0.8, lambda_ct_islands=0.3,
              lambda_getis_ord=0.5, lambda_moran=0.4, lambda_geary=0.3)


def _spatial_problem():
    from oracle import tangram_oracle as orc
    C, K, V = SP_SHAPE
    data = orc.make_synthetic(C, K, V, seed=12, n_types=4)
    M0 = orc.reference_init_M(C, V, 3)
    W = orc.grid_graph(V, standardized=True, self_inclusion=True)          # voxel_weights (mapping_utils.py:320)
    N = orc.grid_graph(V, standardized=False, self_inclusion=False)        # neighborhood_filter (:324)
    Ws = orc.grid_graph(V, standardized=True, self_inclusion=False)        # spatial_weights (:326-329)
    return data, M0, dict(voxel_weights=W, neighborhood_filter=N, ct_encode=data["ct_encode"], spatial_weights=Ws)


def _spatial_worker(rank, world, port, sim_path, outdir, transport="callbacks"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tangram_amd import _capi
        _capi._install_library_for_tests(sim_path)
        from tangram_amd.sharded import make_sharded
        import tangram_amd.mapping_optimizer as mo
        data, M0, graphs = _spatial_problem()
        sh = make_sharded(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=SP_LAM, transport=transport, **graphs)
        n = 4
        hist = sh.eng.new_history(n)
        sh.run(n, 0.1, hist)
        P = sh.result_full()
        sh.peer_check()
        # the same through the Mapper seam (distributed=True): history keys of the reference + the mapping
        m = mo.Mapper(S=data["S"], G=data["G"], d=data["d"], device="cpu", gemm_precision="fp32", M_init=M0, distributed=True,
                      **{k: v for k, v in SP_LAM.items()}, **graphs)
        Pm, hm = m.train(num_epochs=n, print_each=None)
        np.savez(os.path.join(outdir, f"spatial_{rank}.npz"), P=P.numpy(), hist=hist.numpy(), Pm=Pm,
                 total=np.array([float(x) for x in hm["total_loss"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["callbacks", "peer"])
def test_spatial_terms_on_two_shards_match_single_and_oracle(tmp_path, transport):
    """Neighbourhood, cell-type-island and the three autocorrelation terms on 2 spot shards (mapping_optimizer.py:234-263 on the
    whole spot graph: every rank gathers Ghat): same history and mapping as the unsharded engine and the fp64 oracle.
    "peer": the gathered Ghat blocks are longer than the mailbox, i.e. this is the case that moves a vector in PIECES."""
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    mp.spawn(_spatial_worker, args=(2, _free_port(), sim_path, str(tmp_path), transport), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "spatial_0.npz"), np.load(tmp_path / "spatial_1.npz")
    for k in z0.files:
        np.testing.assert_array_equal(z0[k], z1[k], err_msg=k)
    from tangram_amd import _capi
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    data, M0, graphs = _spatial_problem()
    n = 4
    _capi._install_library_for_tests(sim_path)
    try:
        e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cpu", precision="fp32", lambdas=SP_LAM, **graphs)
        h1 = e.new_history(n)
        e.step(n, 0.1, h1)
        P1 = e.result().numpy()
    finally:
        _capi._install_library_for_tests(None)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_NB, _capi.H_CT, _capi.H_GETIS, _capi.H_MORAN, _capi.H_GEARY]
    np.testing.assert_allclose(z0["hist"][:, cols], h1.numpy()[:, cols], atol=3e-6, rtol=2e-6)
    np.testing.assert_allclose(z0["P"], P1, atol=2e-6)
    np.testing.assert_allclose(z0["Pm"], P1, atol=2e-6)
    o = orc.OracleMapper(data["S"], data["G"], d=data["d"], M0=M0, dtype=np.float64, **SP_LAM, **graphs)
    Po, ho = o.train(n, 0.1)
    np.testing.assert_allclose(z0["hist"][:, _capi.H_TOTAL], np.array(ho["total_loss"]), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(z0["total"], np.array(ho["total_loss"]), atol=2e-5, rtol=1e-5)
    assert np.abs(z0["P"] - Po).max() < 2e-5


def test_peer_exchange_gives_up_on_a_missing_peer():
    """The polls of the peer transport are bounded: an exchange whose peer never shows up ends after the time-out with the status word
    raised (tg_comm_peer_status), and every later exchange returns at once instead of waiting again -- a lost rank costs one bounded
    wait, never a hang.  (Emulated build, two communicators of one process, raw-pointer mailboxes; rank 1 simply never calls.)"""
    import ctypes as ct
    import time
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    from tangram_amd import _capi
    lib = _capi._declare(ct.CDLL(sim_path))
    comms, handles = [], ct.create_string_buffer(128)
    for r in range(2):
        c, h = ct.c_void_p(), ct.create_string_buffer(64)
        assert lib.tg_comm_peer_create(2, r, 4096, 1, h, ct.byref(c)) == 0, lib.tg_last_error()
        handles[64 * r: 64 * (r + 1)] = h.raw
        comms.append(c)
    for c in comms:
        assert lib.tg_comm_peer_connect(c, handles) == 0, lib.tg_last_error()
    assert lib.tg_comm_peer_set_timeout_ms(comms[0], 150.0) == 0
    x = np.arange(3000, dtype=np.float32)
    flag = ct.c_int(-1)
    assert lib.tg_comm_peer_status(comms[0], ct.byref(flag)) == 0 and flag.value == 0
    t0 = time.perf_counter()
    assert lib.tg_comm_all_reduce_sum(comms[0], x.ctypes.data, x.size, None) == 0          # rank 1 never pushes
    t1 = time.perf_counter()
    assert 0.1 <= t1 - t0 < 5.0, t1 - t0
    assert lib.tg_comm_peer_status(comms[0], ct.byref(flag)) == 0 and flag.value == 1
    assert lib.tg_comm_all_reduce_sum(comms[0], x.ctypes.data, x.size, None) == 0          # ... and nobody waits a second time
    assert time.perf_counter() - t1 < 0.1
    for c in comms:
        lib.tg_comm_destroy(c)


@pytest.fixture
def sim():
    from tangram_amd import _capi
    path = build_sim()
    if path is None:
        pytest.skip("host clang not available to build the emulator")
    _capi._install_library_for_tests(path)
    yield path
    _capi._install_library_for_tests(None)


def test_a_sharded_step_whose_peer_is_gone_ends_bounded_and_says_so(sim):
    """The exchanges INSIDE the kernels (round 6: step area of the mailbox) are bounded like the exchange kernel: rank 0 of a 2-rank
    peer communicator steps alone (rank 1 created and mapped its mailbox, then never calls) -- attach and two steps return after ONE
    time-out (the raised status word ends every later wait at its first look), `tg_comm_peer_status` reports it, and the history rows
    of those steps are NaN (a caller that prints or consumes rows during a long run sees it at once)."""
    import ctypes as ct
    import time
    import torch
    from tangram_amd import _capi
    from tangram_amd.engine import HipMapperEngine
    from oracle import tangram_oracle as orc
    lib = _capi.lib()
    C, K, V = 40, 30, 64
    data = orc.make_synthetic(C, K, V, seed=5)
    M0 = orc.reference_init_M(C, V, 3)
    Vl = V // 2
    eng = HipMapperEngine(data["S"], data["G"][:Vl], M0[:, :Vl].copy(), d=data["d"][:Vl], device="cpu", precision="bf16x3",
                          lambdas=dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.5), n_spots_total=V, n_ranks=2)
    step_floats = int(eng.sizes.peer_step_floats)
    assert step_floats > 0
    comms, handles = [], ct.create_string_buffer(128)
    for r in range(2):
        c, h = ct.c_void_p(), ct.create_string_buffer(64)
        assert lib.tg_comm_peer_create_stepped(2, r, 6 * C + 64, step_floats, 1, 1, h, ct.byref(c)) == 0, lib.tg_last_error()
        handles[64 * r: 64 * (r + 1)] = h.raw
        comms.append(c)
    for c in comms:
        assert lib.tg_comm_peer_connect(c, handles) == 0, lib.tg_last_error()
    assert lib.tg_comm_peer_set_timeout_ms(comms[0], 150.0) == 0
    t0 = time.perf_counter()
    eng.attach_comm(comms[0])                              # its set-up exchanges wait for rank 1 once, then not again
    hist = torch.zeros((2, _capi.H_NTERMS), dtype=torch.float32)
    eng.step(2, 0.1, hist)
    assert 0.1 <= time.perf_counter() - t0 < 20.0
    flag = ct.c_int(0)
    assert lib.tg_comm_peer_status(comms[0], ct.byref(flag)) == 0 and flag.value == 1
    assert torch.isnan(hist).all(), hist
    eng.release()
    for c in comms:
        lib.tg_comm_destroy(c)


def test_three_shards_over_the_peer_transport(tmp_path):
    """world = 3 over the peer transport (three processes, shared-memory mailboxes): the rank-order sum of three vectors is not
    commutative-safe like a sum of two, so this is the case that shows every rank adds in the SAME order -- all three ranks must hold
    bit-identical histories, mappings, filters and projections -- and the run still is the single-engine run within tolerance."""
    sim_path = build_sim()
    if sim_path is None:
        pytest.skip("host clang not available to build the emulator")
    port = _free_port()
    mp.spawn(_worker, args=(3, port, sim_path, str(tmp_path), "peer"), nprocs=3, join=True)
    z = [np.load(tmp_path / f"sharded_{r}.npz") for r in range(3)]
    for k in z[0].files:
        np.testing.assert_array_equal(z[0][k], z[1][k], err_msg=k)
        np.testing.assert_array_equal(z[0][k], z[2][k], err_msg=k)
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
        h1 = e.new_history(4)
        e.step(4, 0.1, h1)
        P1 = e.result().numpy()
    finally:
        _capi._install_library_for_tests(None)
    cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY]
    np.testing.assert_allclose(z[0]["hist"][:, cols], h1.numpy()[:, cols], atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(z[0]["P"], P1, atol=1e-6)
