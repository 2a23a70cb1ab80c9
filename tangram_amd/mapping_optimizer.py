"""
Host-side mirror of `tangram.mapping_optimizer` (reference: tangram/mapping_optimizer.py) on top of
libtangram_hip.so.  Same class names, constructor arguments, `train()` signatures, return values and
history keys as the reference, so that `tangram.mapping_utils.map_cells_to_space` (and
`mapping_parameter_tuning.train_multiple_Mapper`) can use these classes unchanged:

    Mapper(S, G, d=..., device=..., random_state=..., **hyperparameters).train(learning_rate=, num_epochs=, print_each=)
        -> (mapping_matrix [C, V] float32 ndarray, training_history dict)       (mapping_utils.py:355-363)
    MapperConstrained(...).train(...) -> (mapping_matrix, F_out, training_history)   (mapping_utils.py:383-389)

All arithmetic of the training loop runs in hand-written HIP kernels (tangram_amd/csrc); this file
only prepares inputs, owns the handle and formats the history.  There is no CPU path.

Extra keywords (not in the reference): `gemm_precision` in {"bf16x3" (default, fp32-parity split-bf16
matrix-core products), "fp32" (exact fp32 MFMA), "bf16"}; `distributed` / `group` -- multi-GPU:

    torchrun --nproc-per-node 8 script.py        # one process per GPU, torch.distributed.init_process_group("nccl")
    ad_map = tg.map_cells_to_space(adata_sc, adata_sp, device=f"cuda:{LOCAL_RANK}", ...)    # same call on every rank

    ad_map = tg.map_cells_to_space(adata_sc, adata_sp, device=f"cuda:{LOCAL_RANK}", distributed=True, ...)

Sharding is OPT-IN (`distributed=True`, optionally `group=`): a script may use its process group for independent per-rank work
(folds, seeds, tuning trials), so an initialised group alone never makes a mapper collective.  With `distributed=True` the
SPOTS are sharded over the ranks of the group (tangram_amd/sharded.py: each rank trains its block of columns of M, three small
RCCL exchanges per iteration); the ranks first check that they were handed the same problem (shape, mode, terms), and every rank
returns the full mapping matrix and the same training history.  The initial logits are drawn exactly like the reference does on
every rank (same seed, same full C x V draw) and then sliced, so a sharded run follows the single-GPU trajectory up to summation
order; an UNSEEDED sharded run (`random_state` None or 0) takes its seed from rank 0, so that all ranks still draw the same
logits and filter.
"""
from __future__ import annotations

import logging

import numpy as np
import torch

from .host_rng import legacy_normal_f32
from . import _capi
from .engine import HipMapperEngine

_KEYS = ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg"]                       # reference :378
_VAL_KEYS = ["val_total_loss", "val_gene_sim", "val_sp_sparsity_weighted_sim", "val_entropy"]  # reference :379
_PRINT_NAMES = [  # reference :286-298, with the history column each comes from
    ("Gene-voxel score", _capi.H_MAIN), ("Voxel-gene score", _capi.H_VG), ("Cell densities reg", _capi.H_KL),
    ("Entropy reg", _capi.H_ENTROPY), ("L1 reg", _capi.H_L1), ("L2 reg", _capi.H_L2),
    ("Spatial weighted score", _capi.H_NB), ("Cell type islands penalty", _capi.H_CT),
    ("Getis-Ord score", _capi.H_GETIS), ("Moran score", _capi.H_MORAN), ("Geary score", _capi.H_GEARY),
]


def _to_numpy_f32(x):
    """float32 array for the engine: a torch tensor stays a tensor (a DEVICE tensor is consumed where it is: no host round
    trip for matrices that were built on the GPU, tangram_amd.preprocess), everything else becomes a contiguous ndarray."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.detach().to(torch.float32)
    if hasattr(x, "to_numpy"):            # pandas Series (density priors come from adata.obs)
        x = x.to_numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _shard_context(distributed, group):
    """(sharded?, world, rank).  Sharding is opt-in: `distributed=True` shards the spots over the process group (which must
    be initialised); None / False keep the mapper on its own GPU even when a process group exists (the caller may be using
    it for independent per-rank work)."""
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    if not distributed:
        return False, world, rank
    if not live:
        raise RuntimeError("distributed=True needs an initialised torch.distributed process group (one rank per GPU)")
    if world <= 1:
        return False, world, rank
    return True, world, rank


def _check_same_problem(group, device, signature):
    """Every rank of a sharded mapper must have been handed the same problem: compare a small integer signature (shape, mode,
    active terms) across the group before any state is built; a mismatch raises on EVERY rank instead of hanging later."""
    import torch.distributed as dist
    backend = dist.get_backend(group)
    dev = device if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(x) for x in signature], dtype=torch.int64, device=dev)
    world = dist.get_world_size(group)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    for r, other in enumerate(every):
        if not torch.equal(other.cpu(), mine.cpu()):
            raise ValueError(f"distributed=True: rank {r} was given a different problem than rank {dist.get_rank(group)} "
                             f"({other.tolist()} vs {mine.tolist()}: cells, genes, spots, mode, terms); every rank must make the "
                             "same call")


def _shared_seed(group, device):
    """An unseeded sharded run: rank 0 draws a seed from its global NumPy RNG (advancing it like any other draw), all ranks
    use it -- otherwise every rank would slice a DIFFERENT logits / filter draw."""
    import torch.distributed as dist
    backend = dist.get_backend(group)
    dev = device if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([int(np.random.randint(1, 2**31 - 1))], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def _print_terms(names_vals):
    msg = ["{}: {:.3f}".format(k, v) for k, v in names_vals if not np.isnan(v)]
    print(str(msg).replace("[", "").replace("]", "").replace("'", ""))        # reference :307


class Mapper:
    """MI355X-native drop-in for `tangram.mapping_optimizer.Mapper` (reference :14-408)."""

    def __init__(
        self,
        S,
        G,
        train_genes_idx=None,
        val_genes_idx=None,
        d=None,
        d_source=None,
        lambda_g1=1.0,
        lambda_d=0,
        lambda_g2=0,
        lambda_r=0,
        lambda_l1=0,
        lambda_l2=0,
        lambda_neighborhood_g1=0,
        voxel_weights=None,
        lambda_getis_ord=0,
        lambda_geary=0,
        lambda_moran=0,
        neighborhood_filter=None,
        ct_encode=None,
        lambda_ct_islands=0,
        spatial_weights=None,
        device="cuda:0",
        adata_map=None,
        random_state=None,
        *,
        gemm_precision="bf16x3",
        M_init=None,
        distributed=False,
        group=None,
        init="reference",
        gather_result=True,
        s_exact="auto",
    ):
        """Extra keywords (the defaults give the reference's results):
        s_exact="auto" (default; only meaningful with gemm_precision="bf16x3"): the library checks S once at construction; if
            every element is exactly representable in bf16 (raw counts below 256 are; normalised / log-transformed expression
            is not) both GEMMs run with two matrix-core products per element instead of three -- the SAME values (the skipped
            product adds exact zeros), ~25 % less GEMM time; otherwise the general three-product path runs.  `mapper.
            _engine.effective_precision` says which one was taken.  s_exact=False: always the general path.
        init="device": the initial logits come from the library's counter-based device generator (device_init.py) instead of
            NumPy's global stream -- seed-reproducible (`random_state`), identical for every partition of the spots, and never
            materialised on the host: what a problem of BASELINE config 4's size needs (40 GB of logits; a sharded run with
            init="reference" draws the FULL plane on every rank like the reference would).
        gather_result=False (with distributed=True): `train` returns this rank's block `softmax(M)[:, lo:hi]` only
            (`self.spot_range = (lo, hi)`) instead of assembling the full mapping on every rank."""
        if adata_map is not None:
            raise NotImplementedError("resuming from adata_map is not implemented (neither is it in the reference, :151-153)")
        if init not in ("reference", "device"):
            raise ValueError("init must be 'reference' (NumPy's stream, like the reference) or 'device'")
        self._gather_result = bool(gather_result)
        self.spot_range = None
        lambda_getis_ord = lambda_getis_ord if (lambda_getis_ord and lambda_getis_ord > 0) else 0.0     # reference :170,:174,:179
        lambda_moran = lambda_moran if (lambda_moran and lambda_moran > 0) else 0.0
        lambda_geary = lambda_geary if (lambda_geary and lambda_geary > 0) else 0.0
        if not (lambda_getis_ord or lambda_moran or lambda_geary):
            spatial_weights = None
        if not (lambda_neighborhood_g1 and lambda_neighborhood_g1 > 0):      # reference :234: only evaluated when > 0
            lambda_neighborhood_g1, voxel_weights = 0.0, None
        if not (lambda_ct_islands and lambda_ct_islands > 0):                # reference :242
            lambda_ct_islands, neighborhood_filter, ct_encode = 0.0, None, None
        self.device = torch.device(device)
        self.random_state = random_state
        S = _to_numpy_f32(S)
        G = _to_numpy_f32(G)
        if train_genes_idx is not None:                      # reference :87-92
            S_train, G_train = S[:, train_genes_idx], G[:, train_genes_idx]
        else:
            S_train, G_train = S, G
        self.val_genes_idx = val_genes_idx
        self.lambda_d, self.lambda_g1, self.lambda_g2 = lambda_d, lambda_g1, lambda_g2
        self.lambda_r, self.lambda_l1, self.lambda_l2 = lambda_r, lambda_l1, lambda_l2
        self.target_density_enabled = d is not None          # reference :114
        self.source_density_enabled = d_source is not None   # reference :118
        d = _to_numpy_f32(d)
        d_source = _to_numpy_f32(d_source)
        # the reference ignores lambda_d when d is None (:212-221) and uses d_source only together with d (:214)
        lambdas = dict(lambda_g1=lambda_g1, lambda_d=lambda_d if d is not None else 0.0, lambda_g2=lambda_g2,
                       lambda_r=lambda_r, lambda_l1=lambda_l1, lambda_l2=lambda_l2,
                       lambda_neighborhood_g1=lambda_neighborhood_g1, lambda_ct_islands=lambda_ct_islands,
                       lambda_getis_ord=lambda_getis_ord, lambda_moran=lambda_moran, lambda_geary=lambda_geary)
        sharded, self._world, self._rank = _shard_context(distributed, group)
        if sharded:
            _check_same_problem(group, self.device, [S_train.shape[0], S_train.shape[1], G_train.shape[0], 0,
                                                     d is not None, d_source is not None] + [bool(v) for v in lambdas.values()])
        dev_seed = None
        if M_init is None:
            seed = self.random_state
            if sharded and not seed:
                seed = _shared_seed(group, self.device)
            if init == "device":                             # generated where it is used, block by block (device_init.py)
                dev_seed = int(seed) if seed else int(np.random.randint(1, 2**31 - 1))
                if not sharded:
                    from .device_init import device_normal
                    M_init = device_normal(S.shape[0], G.shape[0], self.device, dev_seed)
            else:
                if seed:                                     # reference :148-150 (seed 0 / None => unseeded)
                    np.random.seed(seed=seed)
                M_init = legacy_normal_f32((S.shape[0], G.shape[0]))     # np.random.normal(0, 1, ...) bit for bit (host_rng.py)
        self._sharded = None
        if sharded:
            from .sharded import make_sharded
            self._sharded = make_sharded(S_train, G_train, M_init, d=d, d_source=d_source if d is not None else None,
                                         device=self.device, precision=gemm_precision, lambdas=lambdas, group=group,
                                         voxel_weights=voxel_weights, neighborhood_filter=neighborhood_filter,
                                         ct_encode=_to_numpy_f32(ct_encode), spatial_weights=spatial_weights,
                                         device_init_seed=dev_seed if M_init is None else None, s_exact=s_exact)
            self._engine = self._sharded.eng
        else:
            self._engine = HipMapperEngine(S_train, G_train, M_init, d=d, d_source=d_source if d is not None else None,
                                           mode="mapper", device=self.device, precision=gemm_precision, lambdas=lambdas,
                                           voxel_weights=voxel_weights, neighborhood_filter=neighborhood_filter,
                                           ct_encode=_to_numpy_f32(ct_encode), spatial_weights=spatial_weights, s_exact=s_exact)

    # ------------------------------------------------------------------------------------------------
    def _history_dict(self, hist):
        h = hist.detach().cpu().numpy()
        out = {k: [] for k in _KEYS + _VAL_KEYS}
        nan = float("nan")
        for row in h:
            out["total_loss"].append(np.array(row[_capi.H_TOTAL], dtype=np.float32))     # 0-d ndarray like :390
            out["main_loss"].append(float(row[_capi.H_MAIN]))
            out["vg_reg"].append(float(row[_capi.H_VG]) if self.lambda_g2 else nan)
            # reference :219-220: kl_reg = (lambda_d * KL) / lambda_d -> nan when d is given but lambda_d == 0
            out["kl_reg"].append(float(row[_capi.H_KL]) if (self.target_density_enabled and self.lambda_d) else nan)
            out["entropy_reg"].append(float(row[_capi.H_ENTROPY]) if self.lambda_r else nan)
        return out

    def train(self, num_epochs, learning_rate=0.1, print_each=100, val_each=None):
        """Run the optimizer; returns (mapping matrix ndarray [C, V], training_history) like the reference (:358-408)."""
        if self.random_state:
            torch.manual_seed(seed=self.random_state)        # reference :371-372 (no RNG is consumed afterwards)
        if print_each:
            logging.info(f"Printing scores every {print_each} epochs.")
        eng = self._engine
        run = self._sharded.run if self._sharded is not None else eng.step      # sharded: kernels + exchanges in one C call
        quiet = self._rank != 0                              # a sharded run prints its scores once, not once per rank
        hist = eng.new_history(max(int(num_epochs), 1))
        val_rows = []
        t = 0
        while t < num_epochs:
            stops = [num_epochs - 1]                         # last epoch of the next chunk (inclusive)
            if print_each:
                stops.append(t if t % print_each == 0 else (t // print_each + 1) * print_each)
            if val_each is not None:
                stops.append(t if t % val_each == 0 else (t // val_each + 1) * val_each)
            n = min(stops) - t + 1
            run(n, learning_rate, hist, t)
            t += n
            if print_each and (t - 1) % print_each == 0 and not (self._sharded is not None and quiet):
                if self._sharded is not None:
                    self._sharded.checked()                  # (peer transport: never print a row an exchange gave up on)
                row = hist[t - 1].detach().cpu().numpy()
                _print_terms([(name, float(row[col])) for name, col in _PRINT_NAMES])
            if val_each is not None and (t - 1) % val_each == 0:
                val_rows.append((self._sharded or eng).validate())   # reference :398-403: after optimizer.step() of epoch t-1
        if self._sharded is not None and not self._gather_result:
            P_local, self.spot_range = self._sharded.result_local()      # this rank's spots only (gather_result=False)
            output = P_local.detach().cpu().numpy()
        elif self._sharded is not None:
            output = self._sharded.result_full(host=True)    # block by block: no rank holds two full C x V copies on its GPU
        else:
            output = eng.result().detach().cpu().numpy()     # reference :406-408
        history = self._history_dict(hist[:num_epochs])
        for row in val_rows:
            for k, x in zip(_VAL_KEYS, row):
                history[k].append(x)
        return output, history

    # extras -----------------------------------------------------------------------------------------
    def release(self):
        """Free the device memory of this mapper (logits, Adam moments, X, workspace); the object is unusable afterwards."""
        (self._sharded or self._engine).release()

    def project_genes_device(self, S_all=None):
        """softmax(M)^T S on the device (what mapping_utils.py:402 and utils.py:368 compute in NumPy on the host);
        `S_all` [n_cells, n_genes_any]: project another gene set (project_genes), default: the training genes.
        On a sharded run every rank projects onto its spots and the row blocks are gathered: [V_total, K] everywhere."""
        if self._sharded is not None:
            return self._sharded.project_full(S_all)
        return self._engine.project() if S_all is None else self._engine.project_genes(S_all)


_KEYS_CONSTRAINED = ["total_loss", "main_loss", "vg_reg", "kl_reg", "entropy_reg", "count_reg", "lambda_f_reg"]  # :609-617
_PRINT_NAMES_CONSTRAINED = [("Score", _capi.H_MAIN), ("VG reg", _capi.H_VG), ("KL reg", _capi.H_KL),
                            ("Entropy reg", _capi.H_ENTROPY), ("Count reg", _capi.H_COUNT),
                            ("Lambda f reg", _capi.H_FREG)]                                              # :555-562


class MapperConstrained:
    """MI355X-native drop-in for `tangram.mapping_optimizer.MapperConstrained` (reference :411-639)."""

    def __init__(
        self,
        S,
        G,
        d,
        lambda_d=1,
        lambda_g1=1,
        lambda_g2=1,
        lambda_r=0,
        lambda_count=1,
        lambda_f_reg=1,
        target_count=None,
        device="cuda:0",
        adata_map=None,
        random_state=None,
        *,
        gemm_precision="bf16x3",
        M_init=None,
        F_init=None,
        distributed=False,
        group=None,
        init="reference",
        gather_result=True,
        s_exact="auto",
    ):
        """`init` / `gather_result` / `s_exact`: as for `Mapper` (init="device" also draws the filter logits F from the device generator)."""
        if adata_map is not None:
            raise NotImplementedError("resuming from adata_map is not implemented (neither is it in the reference, :476-478)")
        if init not in ("reference", "device"):
            raise ValueError("init must be 'reference' (NumPy's stream, like the reference) or 'device'")
        self._gather_result = bool(gather_result)
        self.spot_range = None
        self.device = torch.device(device)
        self.random_state = random_state
        S = _to_numpy_f32(S)
        G = _to_numpy_f32(G)
        self.target_density_enabled = d is not None
        d = _to_numpy_f32(d)
        self.lambda_d, self.lambda_g1, self.lambda_g2, self.lambda_r = lambda_d, lambda_g1, lambda_g2, lambda_r
        self.lambda_count, self.lambda_f_reg = lambda_count, lambda_f_reg
        self.target_count = G.shape[0] if target_count is None else target_count          # :480-483
        lambdas = dict(lambda_g1=lambda_g1, lambda_d=lambda_d if d is not None else 0.0, lambda_g2=lambda_g2,
                       lambda_r=lambda_r, lambda_count=lambda_count, lambda_f_reg=lambda_f_reg)
        sharded, self._world, self._rank = _shard_context(distributed, group)
        if sharded:
            _check_same_problem(group, self.device, [S.shape[0], S.shape[1], G.shape[0], 1, d is not None, 0] +
                                [bool(v) for v in lambdas.values()])
        dev_seed = None
        if M_init is None or F_init is None:
            seed = self.random_state
            if sharded and not seed:                                                       # (every rank must draw the same M and F)
                seed = _shared_seed(group, self.device)
            if init == "device":
                from .device_init import device_normal
                dev_seed = int(seed) if seed else int(np.random.randint(1, 2**31 - 1))
                M_init = None if sharded else device_normal(S.shape[0], G.shape[0], self.device, dev_seed)
                F_init = device_normal(1, S.shape[0], self.device, dev_seed, stream_id=1).reshape(-1)
            else:
                if seed:                                                                   # :473-474
                    np.random.seed(seed=seed)
                legacy_normal_f32((S.shape[0], G.shape[0]), discard=True)                  # :475 (first draw is discarded by :485)
                M_init = legacy_normal_f32((S.shape[0], G.shape[0]))                       # :485
                F_init = np.random.normal(0, 1, S.shape[0]).astype(np.float32)             # :490
        self._sharded = None
        if sharded:
            from .sharded import make_sharded
            self._sharded = make_sharded(S, G, M_init, d=d, F0=F_init, mode="constrained", device=self.device,
                                         precision=gemm_precision, lambdas=lambdas, target_count=float(self.target_count), group=group,
                                         device_init_seed=dev_seed if M_init is None else None, s_exact=s_exact)
            self._engine = self._sharded.eng
        else:
            self._engine = HipMapperEngine(S, G, M_init, d=d, F0=F_init, mode="constrained", device=self.device,
                                           precision=gemm_precision, lambdas=lambdas, target_count=float(self.target_count),
                                           s_exact=s_exact)

    def train(self, num_epochs, learning_rate=0.1, print_each=100):
        """Returns (mapping matrix [C, V], filter [C], training_history) like the reference (:589-639).
        History values are strings like the reference's (`str(x)`, :630); `total_loss` is `str(float)` rather than
        the reference's tensor repr."""
        if self.random_state:
            torch.manual_seed(seed=self.random_state)
        eng = self._engine
        run = self._sharded.run if self._sharded is not None else eng.step
        hist = eng.new_history(max(int(num_epochs), 1))
        t = 0
        while t < num_epochs:
            if print_each:
                nxt = t if t % print_each == 0 else (t // print_each + 1) * print_each
                n = min(nxt, num_epochs - 1) - t + 1
            else:
                n = num_epochs - t
            run(n, learning_rate, hist, t)
            t += n
            if print_each and (t - 1) % print_each == 0 and not (self._sharded is not None and self._rank != 0):
                if self._sharded is not None:
                    self._sharded.checked()
                row = hist[t - 1].detach().cpu().numpy()
                _print_terms([(name, float(row[col])) for name, col in _PRINT_NAMES_CONSTRAINED])
        if self._sharded is not None and not self._gather_result:
            P, self.spot_range, F = self._sharded.result_local(with_filter=True)
        elif self._sharded is not None:
            P, F = (torch.as_tensor(x) for x in self._sharded.result_full(with_filter=True, host=True))
        else:
            P, F = eng.result(with_filter=True)
        return P.detach().cpu().numpy(), F.detach().cpu().numpy(), self._history_dict(hist[:num_epochs])

    def _history_dict(self, hist):
        """History rows -> the reference's dict of stringified values (:609-617, :630)."""
        h = hist.detach().cpu().numpy()
        cols = [_capi.H_TOTAL, _capi.H_MAIN, _capi.H_VG, _capi.H_KL, _capi.H_ENTROPY, _capi.H_COUNT, _capi.H_FREG]
        active = [True, True, bool(self.lambda_g2), bool(self.target_density_enabled and self.lambda_d), bool(self.lambda_r), True, True]
        history = {k: [] for k in _KEYS_CONSTRAINED}
        for row in h:
            for k, c, on in zip(_KEYS_CONSTRAINED, cols, active):
                history[k].append(str(float(row[c]) if on else float("nan")))
        return history

    def release(self):
        """Free the device memory of this mapper; the object is unusable afterwards."""
        (self._sharded or self._engine).release()

    def project_genes_device(self, S_all=None, unfiltered=True):
        """Like Mapper.project_genes_device; `unfiltered`: softmax(M) alone, as `adata_map.X` holds it (:637), else times
        the learned filter.  Without `S_all`: the (filtered) training-time projection."""
        if self._sharded is not None:
            return self._sharded.project_full(S_all, unfiltered=unfiltered)
        return self._engine.project() if S_all is None else self._engine.project_genes(S_all, unfiltered=unfiltered)
