"""`tg_mapper_step` and `tg_batch_step` under HIP stream capture (include/tangram_hip.h: "nothing in it synchronises, times or
queries the device, so a call can be captured into a HIP graph").

Each case trains n steps eagerly from a saved state, restores the state, CAPTURES the same n steps on the handle's stream
(nothing executes while capturing), restores the state again, replays the graph and compares BITS with the eager run: history
rows, mapping, logits and both Adam moments.  The step indices (Adam bias corrections) and history rows are launch arguments,
so a replay repeats exactly the captured steps -- which is what is compared.  A second replay from the restored state must
reproduce the same bits again.
Reference loop being replaced: tangram/mapping_optimizer.py:382-396 (cells / clusters), utils.py:576-600 (the folds of a batch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _constrained(e):
    from tangram_amd import _capi
    return e.cfg.mode == _capi.TG_MODE_CONSTRAINED


def _save(e):
    M, m1, m2, step = e.logits()
    st = [M.clone(), m1.clone(), m2.clone()]
    if _constrained(e):
        st.append(e.filter_state().clone())
    return st


def _restore(e, st):
    M, m1, m2, _ = e.logits()
    M.copy_(st[0]); m1.copy_(st[1]); m2.copy_(st[2])
    if _constrained(e):
        e.filter_state().copy_(st[3])
    e.set_step(0)                                  # rebuilds the softmax statistics of the restored logits (tg_mapper_set_step)


def _bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("shape,mode", [((600, 80, 300), "cells"), ((18, 120, 2000), "clusters"), ((300, 60, 200), "constrained")],
                         ids=["cells", "clusters_sc_kernels", "constrained"])
def test_mapper_step_captured_into_a_graph_replays_bit_identically(shape, mode):
    import torch
    from oracle import tangram_oracle as orc
    from tangram_amd.engine import HipMapperEngine
    C, K, V = shape
    n, lr = 50, 0.1
    data = orc.make_synthetic(C, K, V, seed=21)
    M0 = orc.reference_init_M(C, V, 42)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                     # the handle binds to the stream that is current at construction
        if mode == "constrained":
            F0 = np.random.default_rng(3).normal(size=C).astype(np.float32)
            e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device="cuda:0", precision="bf16x3",
                                lambdas=dict(lambda_d=1.0, lambda_g1=1.0, lambda_g2=0.5, lambda_count=1.0, lambda_f_reg=1.0), target_count=float(V // 2))
        else:
            e = HipMapperEngine(data["S"], data["G"], M0, d=data["d"], device="cuda:0", precision="bf16x3", lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
        st = _save(e)
        h_eager = e.new_history(n)
        e.step(n, lr, h_eager)
        P_eager = e.result()
        M_e, m1_e, m2_e = (x.clone() for x in e.logits()[:3])
        s.synchronize()
        assert np.isfinite(h_eager.cpu().numpy()[:, 0]).all()

        _restore(e, st)
        h_graph = e.new_history(n)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            e.step(n, lr, h_graph)                 # captured, not executed
        assert np.isnan(h_graph.cpu().numpy()).all(), "capturing must not execute the steps"
        for _ in range(2):                         # two replays from the restored state: the same bits both times
            _restore(e, st)
            h_graph.fill_(float("nan"))
            g.replay()
            s.synchronize()
            assert np.array_equal(_bits(h_graph), _bits(h_eager))
            assert np.array_equal(_bits(e.result()), _bits(P_eager))
            M_g, m1_g, m2_g = e.logits()[:3]
            assert np.array_equal(_bits(M_g), _bits(M_e)) and np.array_equal(_bits(m1_g), _bits(m1_e)) and np.array_equal(_bits(m2_g), _bits(m2_e))
        e.release()


def test_batch_step_captured_into_a_graph_replays_bit_identically():
    import torch
    from oracle import tangram_oracle as orc
    from tangram_amd.mapping_optimizer import Mapper
    from tangram_amd.batched import MapperBatch
    C, K, V, B = 18, 100, 1500, 6
    n, lr = 40, 0.1
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        mappers = []
        for i in range(B):
            data = orc.make_synthetic(C, K, V, seed=100 + i)
            mappers.append(Mapper(data["S"], data["G"], d=data["d"], lambda_d=1, lambda_g1=1, device="cuda:0", M_init=orc.reference_init_M(C, V, 7 + i)))
        batch = MapperBatch(mappers)
        engines = batch.engines
        states = [_save(e) for e in engines]
        hists = batch.new_histories(n)
        batch.step(n, lr, hists, 0)                # eager; also uploads the batch's argument arrays for THESE history buffers
        s.synchronize()
        eager_h = [h.clone() for h in hists]
        eager_P = [e.result().clone() for e in engines]

        for e, st in zip(engines, states):
            _restore(e, st)
        for h in hists:
            h.fill_(float("nan"))
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            batch.step(n, lr, hists, 0)            # same history pointers: kernel launches + the groups' fork / join events only
        assert all(np.isnan(h.cpu().numpy()).all() for h in hists)
        for e, st in zip(engines, states):
            _restore(e, st)
        g.replay()
        s.synchronize()
        for i in range(B):
            assert np.array_equal(_bits(hists[i]), _bits(eager_h[i])), i
            assert np.array_equal(_bits(engines[i].result()), _bits(eager_P[i])), i

        # a capture that WOULD have to re-upload the argument arrays (other history buffers) is refused, not silently wrong
        other = batch.new_histories(n)
        for e, st in zip(engines, states):
            _restore(e, st)
        s.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="capture"):
            with torch.cuda.graph(g2, stream=s, capture_error_mode="thread_local"):
                batch.step(n, lr, other, 0)
        batch.close()
