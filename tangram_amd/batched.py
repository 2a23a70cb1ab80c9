"""
Independent mappings side by side on one GPU.

The reference runs independent mappings strictly one after the other: `cross_val` trains one mapping per held-out gene
(utils.py:576-600; 249 in the tutorial), the tuning driver three seeds per trial (mapping_parameter_tuning.py:109-131).
Clusters-mode problems are tiny (18 x 250 x 9852) and launch-bound on a 256-CU GPU: one iteration is ~6 dependent kernels of
~10 us that each occupy a fraction of the chip.

`MapperBatch` / `train_many` advance B mappings of one shape in ONE launch per kernel: the C library's `tg_batch` puts the
per-mapping kernel arguments into device arrays and adds a batch index to the grid of every kernel of the iteration
(blockIdx.z = mapping).  Results are the bits of training the same mappings one by one: nothing is shared between them.
`Mapper`s batch with `Mapper`s, `MapperConstrained`s with `MapperConstrained`s (utils.py:576-600 passes any `mode`).
Mappings that are not batched (a shape of their own, spatial terms, more than 16 384 spots, more than 2^25 cells x spots -- one
such mapping fills the GPU and a batch of them measured slower than one after the other --, or a group the C library
refuses) are trained by one host thread per mapping (the C ABI releases the GIL; different handles may be driven from different
threads); `batched=False` additionally gives every mapping a HIP stream of its own.
"""
from __future__ import annotations

import ctypes as ct
from concurrent.futures import ThreadPoolExecutor

import torch

from . import _capi


class MapperBatch:
    """B `Mapper`s -- or B `MapperConstrained`s -- of one shape / configuration stepped together (tg_batch)."""

    def __init__(self, mappers):
        self.mappers = list(mappers)
        if not self.mappers:
            raise ValueError("empty batch")
        engines = [m._engine for m in self.mappers]
        self.engines = engines
        e0 = engines[0]
        self._lib = e0._lib
        n = len(engines)
        self._scratch = torch.empty(int(self._lib.tg_batch_query_bytes(n)), dtype=torch.uint8, device=e0.device)
        arr = (ct.c_void_p * n)(*[e._h for e in engines])
        handle = ct.c_void_p()
        e0._call(self._lib.tg_batch_create, arr, n, self._scratch.data_ptr(), ct.byref(handle))
        self._h = handle

    def new_histories(self, n_rows):
        return [e.new_history(n_rows) for e in self.engines]

    def step(self, n_steps, lr, histories=None, first_row=0):
        n = len(self.engines)
        arr = (ct.c_void_p * n)(*[h.data_ptr() for h in histories]) if histories is not None else None
        self.engines[0]._call(self._lib.tg_batch_step, self._h, int(n_steps), float(lr), arr, int(first_row),
                              tensors=tuple(histories or ()))

    def close(self):
        if getattr(self, "_h", None):
            self.engines[0]._sync()
            self._lib.tg_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ROWPASS_MAX_SPOTS = 16384          # TG_ROWPASS_MAX_V (tg_capi.hip): rows the single-kernel update holds in registers
BATCH_MAX_ELEMENTS = 1 << 25       # cells x spots above which ONE mapping fills the GPU: batching then LOSES (measured, B = 2 - 4 vs one
                                   # after the other: 4 200 x 1 100: 1.4 - 1.75x, 8 000 x 3 000: 1.05 - 1.13x, 12 000 x 5 000: 0.96 - 0.98x,
                                   # 26 431 x 9 852: 0.83 - 0.85x; profiles/r03/run15_batch_sizes)
EMIT_MAX_GENE_COLS = 6128          # (2 Kp + 32) floats of dynamic LDS <= 48 KB in tg_dghat_emit<SELF>


def _batch_key(m):
    """Mappings with equal keys can share a tg_batch: the limits of tg_batch_create (tg_capi.hip) are applied here, so that a
    group the C library would refuse is never formed (the C library re-checks; a refusal falls back to the stream path)."""
    e = getattr(m, "_engine", None)
    if type(m).__name__ not in ("Mapper", "MapperConstrained") or e is None or getattr(m, "_sharded", None) is not None:
        return None
    c = e.cfg
    if c.lambda_neighborhood_g1 or c.lambda_ct_islands or c.lambda_getis_ord or c.lambda_moran or c.lambda_geary:
        return None
    if e.C * e.V > BATCH_MAX_ELEMENTS:
        return None
    if e.V > ROWPASS_MAX_SPOTS or e.K + 1 + 256 > EMIT_MAX_GENE_COLS or c.pipeline_bands > 1:
        return None
    stream = e._torch_stream.cuda_stream if e._torch_stream is not None else 0
    # (effective_precision: folds built with s_exact="auto" whose S differs in bf16-exactness run different GEMM kernels)
    return (type(m).__name__, e.C, e.K, e.V, e.precision, getattr(e, "effective_precision", e.precision), bool(c.lambda_r or c.lambda_l1 or c.lambda_l2), str(e.device), stream,
            c.beta1, c.beta2, c.tile_size, c.fwd_splits)


def _train_batched(mappers, num_epochs, learning_rate, val_each=None):
    """`val_each` (Mapper only; the tuning driver trains its three seeds with val_each = 1, mapping_parameter_tuning.py:110-129): the batch
    advances to the next validation epoch in one call, every handle's `_val_loss_fn` numbers are enqueued into a device table
    (tg_mapper_validate, no synchronisation) and the batch goes on -- the per-handle sequence of calls, hence the bits, of
    `Mapper.train(val_each=...)` on each mapping alone."""
    from .mapping_optimizer import _VAL_KEYS
    batch = MapperBatch(mappers)
    num_epochs = int(num_epochs)
    hists = batch.new_histories(max(num_epochs, 1))
    vals = None
    if val_each is not None:
        n_val = sum(1 for t in range(1, num_epochs + 1) if (t - 1) % val_each == 0)
        vals = [torch.empty((max(n_val, 1), 4), dtype=torch.float32, device=m._engine.device) for m in mappers]
    t, iv = 0, 0
    while t < num_epochs:
        if val_each is not None:
            stop = t if t % val_each == 0 else (t // val_each + 1) * val_each      # last epoch of this chunk (inclusive), like Mapper.train
            n = min(stop, num_epochs - 1) - t + 1
        else:
            n = num_epochs - t
        batch.step(n, learning_rate, hists, t)
        t += n
        if val_each is not None and (t - 1) % val_each == 0:   # reference :398-403: after optimizer.step() of epoch t - 1
            for m, v in zip(mappers, vals):
                m._engine.validate_into(v[iv])
            iv += 1
    out = []
    for j, (m, h) in enumerate(zip(mappers, hists)):
        if type(m).__name__ == "MapperConstrained":            # train() -> (mapping, filter, history) (mapping_utils.py:387-389)
            P, F = m._engine.result(with_filter=True)
            out.append((P.detach().cpu().numpy(), F.detach().cpu().numpy(), m._history_dict(h[:num_epochs])))
        else:
            history = m._history_dict(h[:num_epochs])
            if vals is not None:
                for row in vals[j][:iv].detach().cpu().numpy():
                    for k, x in zip(_VAL_KEYS, row):
                        history[k].append(float(x))
            out.append((m._engine.result().detach().cpu().numpy(), history))
    batch.close()
    return out


def train_many(builders, num_epochs, learning_rate=0.1, max_concurrent=4, device="cuda:0", batched="auto", **train_kwargs):
    """Train independent mappings together.

    builders: callables, each returning a `Mapper` / `MapperConstrained`.  They are called one after the other on the
              calling thread (the reference's initialisation draws from the global NumPy RNG, `np.random.seed(random_state)`,
              which must not be interleaved).
    batched:  "auto" (default): mappings that can share a `tg_batch` (one class, one shape, no spatial terms, <= 16 384 spots)
              advance in ONE launch per kernel -- `val_each=k` (Mapper) included: the batch pauses at every validation epoch; the others -- and any group the C library refuses -- get a host
              thread each, their kernels sharing the common creation stream.  False: every mapper is created on a HIP stream
              of its own and trained by its own host thread (`max_concurrent` at a time): kernels of different mappings overlap.
    Returns the list of `mapper.train(...)` results (in the order of `builders`) and the mappers themselves."""
    device = torch.device(device)
    builders = list(builders)
    n = len(builders)
    results, mappers = [None] * n, [None] * n
    val_each = train_kwargs.get("val_each")
    if batched and set(train_kwargs) <= {"val_each"}:
        with (torch.cuda.device(device) if device.type == "cuda" else _null()):
            for i in range(n):
                mappers[i] = builders[i]()
            groups = {}
            for i, m in enumerate(mappers):
                key = _batch_key(m)
                if val_each is not None and type(m).__name__ != "Mapper":          # (MapperConstrained.train has no val_each: let it say so)
                    key = None
                groups.setdefault(key or ("single", i), []).append(i)
            rest = []
            for key, idx in groups.items():
                if key[0] != "single" and len(idx) > 1:
                    try:
                        for i, r in zip(idx, _train_batched([mappers[i] for i in idx], num_epochs, learning_rate, val_each)):
                            results[i] = r
                        continue
                    except (RuntimeError, ValueError) as e:             # refused by tg_batch_create: nothing has been stepped yet
                        if any(mappers[i]._engine.logits()[3] != 0 for i in idx):
                            raise
                        import logging
                        logging.info("tangram_amd: %d mappings not batchable (%s); training them on streams", len(idx), e)
                rest += idx
        _train_on_streams(sorted(rest), mappers, results, device, num_epochs, learning_rate, max_concurrent, train_kwargs)
        return results, mappers
    streams = {}
    with (torch.cuda.device(device) if device.type == "cuda" else _null()):
        for i in range(n):
            if device.type == "cuda":                           # the library binds a handle to the stream it is created on
                streams[i] = torch.cuda.Stream(device=device)
                with torch.cuda.stream(streams[i]):
                    mappers[i] = builders[i]()
                streams[i].synchronize()
            else:
                mappers[i] = builders[i]()
    _train_on_streams(list(range(n)), mappers, results, device, num_epochs, learning_rate, max_concurrent, train_kwargs, streams)
    return results, mappers


def _train_on_streams(idx, mappers, results, device, num_epochs, learning_rate, max_concurrent, train_kwargs, streams=None):
    """One host thread per mapping, `max_concurrent` at a time, each on the HIP stream its mapper was created on (`streams`;
    kernels of different mappings then overlap on the GPU).  Without `streams` (mappers built on one common stream, the
    left-overs of the batched path) the threads only overlap the host side: a handle's kernels run on its creation stream."""
    if not idx:
        return
    if device.type != "cuda":                                   # (emulated build of the CPU test-suite: no streams)
        for i in idx:
            results[i] = mappers[i].train(num_epochs=num_epochs, learning_rate=learning_rate, print_each=None, **train_kwargs)
        return
    streams = streams or {i: torch.cuda.Stream(device=device) for i in idx}

    def work(i):
        with torch.cuda.device(device), torch.cuda.stream(streams[i]):
            results[i] = mappers[i].train(num_epochs=num_epochs, learning_rate=learning_rate, print_each=None, **train_kwargs)
        streams[i].synchronize()

    with ThreadPoolExecutor(max_workers=max(1, int(max_concurrent))) as pool:
        for f in [pool.submit(work, i) for i in idx]:
            f.result()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
