"""Spot-sharded multi-GPU driver: one process per GPU (torch.distributed launches the ranks; backend "nccl" = RCCL over
xGMI), each rank owns a contiguous block of spots: M[:, V_g], its Adam moments, G[V_g, :], d[V_g].  S (and the
constrained-mode filter F) are replicated.  Per iteration exactly three small vectors cross GPUs:

    E2  per-gene cosine statistics [2][Kp]            all-reduce(sum)     after the forward GEMM
    E3  per-cell softmax-backward row dots [np][C]    all-reduce(sum)     after the backward GEMM
    E1  per-cell (max, sum exp) of the new logits     all-gather + merge  after the Adam update (+ 2 history scalars per rank)

The gradient of M itself is column-sharded exactly like M and never leaves its GPU.  The whole step -- kernels AND the three
exchanges -- is issued by the C library inside `tg_mapper_step` (include/tangram_hip.h: tg_comm, tg_mapper_attach_comm):

  * transport "rccl": the library binds librccl.so itself and calls ncclAllReduce / ncclAllGather on the handle's stream, between
    its own kernels: no Python, no second stream and no event between a kernel and the collective that consumes its output.
    The communicator is bootstrapped with one 128-byte broadcast over torch.distributed.
  * transport "peer" (round 5): no collective library at all -- every rank owns a mailbox in its HBM that its peers map (hipIpc), an
    exchange is ONE kernel on the handle's stream and ONE xGMI hop (8-byte {value, sequence number} granules stored into every
    mailbox, polled in the own one, summed in rank order): the latency of a kernel launch, not of a 2 (N - 1)-hop ring.
    Round 6: with a STEP AREA in the mailbox (tg_comm_peer_create_stepped) a step launches no exchange kernel at all -- the per-gene statistics
    are pushed and polled inside tg_gene_reduce, the row sums inside the update kernel (the row stays in registers across the exchange),
    the row pairs are pushed from the update kernel's tail and polled at the head of the merge.  OPT-IN (`transport="peer_checked"`,
    or TG_SHARD_TRANSPORT=peer_checked in the environment): set-up + a self-test against the group's own all-reduce / all-gather on
    this very topology, all ranks agreeing on the verdict; any failure (mailbox allocation, hipIpc mapping, a wrong or late result)
    falls back to "rccl", logged.  `transport="auto"` (the default) is RCCL on an nccl group: the peer transport has never crossed
    xGMI (no box of any round had two GPUs), and a default must not rest on an unmeasured path (round-5 advisor).
  * transport "callbacks": the library calls back into this module, which runs the collective through any object with
    `all_reduce(t)` / `all_gather_into_tensor(out, t)` (torch.distributed with gloo in the CPU tests, an in-process communicator
    for several shards of one GPU in the GPU tests).

The reference has no distributed code at all; this module is new capability, not a translation.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .engine import HipMapperEngine


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DistComm:
    """The collectives of the sharded step on a torch.distributed process group."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce(self, t):
        dist.all_reduce(t, group=self.group)

    def all_gather_into_tensor(self, out, t):
        dist.all_gather_into_tensor(out, t, group=self.group)

    def all_gather(self, outs, t):
        dist.all_gather(outs, t, group=self.group)

    def broadcast(self, t, src=0):
        dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)

    def barrier(self):
        dist.barrier(group=self.group)


def _single_node(group, device):
    """Do all ranks of the group run on one host?  (hipIpc mailboxes need it.)"""
    import socket
    import zlib
    h = float(zlib.crc32(socket.gethostname().encode()) % (1 << 23))
    t = torch.tensor([h, -h], dtype=torch.float32, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t[0].item() == h and -t[1].item() == h)


def shard_bounds(n, world, rank, blocks=False):
    """Contiguous partition of range(n) into `world` blocks.  Default: balanced (the first n % world blocks one longer).
    `blocks=True`: every block ceil(n / world) long, only the last shorter -- the partition of runs with spatial terms, whose
    all-gathered blocks must BE the global spot-by-gene matrix (padding only at its end)."""
    if blocks:
        size = -(-n // world)
        return min(rank * size, n), min((rank + 1) * size, n)
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rccl_library_path():
    """The librccl.so PyTorch-ROCm ships (the one torch.distributed's "nccl" backend uses), else the ROCm installation's."""
    for p in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"):
        if os.path.exists(p):
            return p
    return None


class ShardedMapperEngine:
    def __init__(self, S, G_local, M0_local, d_local=None, d_source=None, F0=None, *, n_spots_total, device, mode="mapper",
                 precision="bf16x3", lambdas=None, target_count=0.0, group=None, fwd_splits=0, tile_size=0, bwd_tile=0, comm=None,
                 transport="auto", spot_offset=0, voxel_weights=None, neighborhood_filter=None, ct_encode=None, spatial_weights=None,
                 s_exact=False):
        # `comm`: anything with world, rank, all_reduce, all_gather_into_tensor, all_gather (tests drive several shards of one
        # GPU through an in-process communicator); default: the torch.distributed group
        self.pycomm = comm if comm is not None else DistComm(group)
        self.group = group
        self.world = self.pycomm.world
        self.rank = self.pycomm.rank
        self.lam = dict(lambda_g1=1.0, lambda_d=0.0, lambda_g2=0.0, lambda_r=0.0, lambda_l1=0.0, lambda_l2=0.0)
        self.lam.update(lambdas or {})
        # spatial terms (spot graphs over ALL spots, replicated on every rank): the library gathers Ghat every iteration and evaluates
        # them on the whole graph, identically on every rank; the shards must then be the blocks of shard_bounds(..., blocks=True)
        self.spatial = any(self.lam.get(k, 0) > 0 for k in ("lambda_neighborhood_g1", "lambda_ct_islands", "lambda_getis_ord",
                                                             "lambda_moran", "lambda_geary"))
        self.mode = mode
        if F0 is not None and self.world > 1:
            # the filter logits are REPLICATED: every rank must start from rank 0's values (a caller that draws F0 per rank from
            # an unseeded RNG would otherwise gate the all-reduced statistics differently on every rank and silently diverge)
            F0 = self._from_rank0(F0, torch.device(device))
        self.eng = HipMapperEngine(S, G_local, M0_local, d=d_local, d_source=d_source, F0=F0, mode=mode, device=device,
                                   precision=precision, lambdas=self.lam, n_spots_total=n_spots_total, n_ranks=self.world,
                                   fwd_splits=fwd_splits, tile_size=tile_size, bwd_tile=bwd_tile, target_count=target_count,
                                   spot_offset=spot_offset, voxel_weights=voxel_weights, neighborhood_filter=neighborhood_filter,
                                   ct_encode=ct_encode, spatial_weights=spatial_weights, s_exact=s_exact)
        self.has_density = d_local is not None
        self.n_spots_total = int(n_spots_total)
        lib = self.eng._lib
        if transport == "auto":
            transport = os.environ.get("TG_SHARD_TRANSPORT", "auto")      # (experiments: "peer", "peer_checked", "rccl", "callbacks")
        fallback = None
        if transport in ("auto", "peer_checked"):
            is_nccl = comm is None and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
            base = "rccl" if (is_nccl and not _capi.is_emulated() and self.eng.device.type == "cuda") else "callbacks"
            # Default: the process group's own transport (RCCL on an nccl group).  The peer transport is opt-in ("peer_checked": only after
            # its self-test on this very topology, every rank deciding the same; "peer": unconditionally) until it has been run and
            # timed on a node with more than one GPU.
            if transport == "auto":
                transport = base
            elif not (self.world <= 16 and (comm is not None or _single_node(group, self.eng.device))):
                transport = base                         # (mailboxes are mapped through hipIpc: one node, at most 16 ranks)
            fallback = base
        self._error = None
        self._init_transport(lib, group, transport, comm, fallback)

    def _init_transport(self, lib, group, transport, comm, fallback):
        self.transport = transport
        handle = ct.c_void_p()
        if transport == "rccl":
            uid = torch.zeros(128, dtype=torch.uint8)
            path = rccl_library_path()
            cpath = path.encode() if path else None
            if self.rank == 0:
                buf = ct.create_string_buffer(128)
                _capi.check(lib.tg_comm_rccl_unique_id(cpath, buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            uid = uid.to(self.eng.device)                # the nccl backend broadcasts device tensors
            self.pycomm.broadcast(uid, 0) if hasattr(self.pycomm, "broadcast") else dist.broadcast(uid, 0, group=group)
            raw = bytes(uid.cpu().numpy().tobytes())
            with torch.cuda.device(self.eng.device):
                _capi.check(lib.tg_comm_create_rccl(cpath, raw, self.world, self.rank, ct.byref(handle)))
        elif transport == "callbacks":
            self._cb_ar = _capi.ALL_REDUCE_FN(self._cb_all_reduce)       # (kept alive: the C side stores the function pointers)
            self._cb_ag = _capi.ALL_GATHER_FN(self._cb_all_gather)
            _capi.check(lib.tg_comm_create_callbacks(self.world, self.rank, self._cb_ar, self._cb_ag, None, ct.byref(handle)))
        elif transport in ("peer", "peer_checked"):
            # one-hop exchange kernels over mailboxes the ranks map from each other (include/tangram_hip.h: tg_comm_peer_create).
            # "peer_checked": only after a self-test against the process group's own collectives; else (collectively) fall back.
            handle = self._make_peer_comm(lib, group, selftest=(transport == "peer_checked"))
            if handle is None:
                if transport == "peer":
                    raise RuntimeError("tangram_amd: the peer-memory transport could not be set up on this node (see the log)")
                return self._init_transport(lib, group, fallback, comm, None)
            self.transport = "peer"
        else:
            raise ValueError("transport must be 'auto', 'rccl', 'peer', 'peer_checked' or 'callbacks'")
        self._comm = handle
        self._attach()

    # -- peer transport set-up (collective; every step's verdict is agreed on by all ranks before anybody goes on) ---------------
    def _all_agree(self, ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32)
        if isinstance(self.pycomm, DistComm) and dist.get_backend(self.group) == "nccl":
            t = t.to(self.eng.device)
        if isinstance(self.pycomm, DistComm):
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return bool(t.item() > 0.5)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.pycomm.all_gather(outs, t)
        return all(bool(o.item() > 0.5) for o in outs)

    def _make_peer_comm(self, lib, group, selftest):
        import logging
        log = logging.getLogger("tangram_amd")
        dev = self.eng.device
        same = bool(getattr(self.pycomm, "same_process", False))
        cap = max(6 * self.eng.C + 64, 2 * (self.eng.K + 1024))               # the longest per-step vector; longer ones travel in pieces
        # Messages of megabytes (the per-step all-gather of Ghat in runs with spatial terms) go in `cap`-sized pieces through the same
        # mailbox; the step area behind it (sizes.peer_step_floats) is what lets a step run its three exchanges inside its kernels.
        step = int(self.eng.sizes.peer_step_floats)
        colocated = self._colocated_ranks(dev)
        self.mailbox_bytes = 256 + 2 * self.world * (cap + step + 1024) * 8      # (reported: it is the one allocation outside tg_query_sizes)
        handle, ok, why = ct.c_void_p(), True, ""
        ctx = torch.cuda.device(dev) if dev.type == "cuda" else _Null()
        with ctx:
            buf = ct.create_string_buffer(64)
            try:
                _capi.check(lib.tg_comm_peer_create_stepped(self.world, self.rank, cap, step, colocated, int(same), buf, ct.byref(handle)))
            except Exception as e:        # noqa: BLE001
                ok, why = False, f"create: {e}"
            if not self._all_agree(ok):
                log.info("tangram_amd: peer transport not available (%s)", why or "another rank failed to create its mailbox")
                if handle:
                    lib.tg_comm_destroy(handle)
                return None
            mine = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            on_dev = isinstance(self.pycomm, DistComm) and dist.get_backend(group) == "nccl"
            if on_dev:
                mine = mine.to(dev)
            outs = [torch.empty_like(mine) for _ in range(self.world)]
            self.pycomm.all_gather(outs, mine)
            allh = b"".join(bytes(o.cpu().numpy().tobytes()) for o in outs)
            try:
                _capi.check(lib.tg_comm_peer_connect(handle, allh))
            except Exception as e:        # noqa: BLE001
                ok, why = False, f"connect: {e}"
            if not self._all_agree(ok):
                log.info("tangram_amd: peer transport not available (%s)", why or "another rank could not map the mailboxes")
                lib.tg_comm_destroy(handle)
                return None
            if hasattr(self.pycomm, "barrier"):
                self.pycomm.barrier()                                        # nobody pushes before everybody has mapped everybody
            if selftest:
                try:
                    ok = self._peer_selftest(lib, handle, cap)
                    why = "" if ok else "self-test: results differ from the process group's collectives"
                except Exception as e:    # noqa: BLE001
                    ok, why = False, f"self-test: {e}"
                if not self._all_agree(ok):
                    log.warning("tangram_amd: peer transport failed its self-test on this node (%s); using %s", why or "on another rank", "the process group's transport")
                    self.eng._sync()
                    lib.tg_comm_destroy(handle)
                    return None
                log.info("tangram_amd: peer-memory transport verified on this node (%d ranks)", self.world)
        return handle

    def _colocated_ranks(self, dev):
        """How many ranks of the communicator run on THIS rank's device (1 in deployment; the one-GPU tests: all of them)."""
        if self.world == 1:
            return 1
        import socket
        import zlib
        ident = socket.gethostname()
        if dev.type == "cuda":
            pr = torch.cuda.get_device_properties(dev)
            ident += "|" + str(getattr(pr, "uuid", "")) + "|" + "/".join(str(getattr(pr, k, "")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        else:
            ident += "|cpu"                              # (emulated build: every process is its own "device")
            return 1
        mine = torch.tensor([float(zlib.crc32(ident.encode()) % (1 << 23))], dtype=torch.float32)
        if isinstance(self.pycomm, DistComm) and dist.get_backend(self.group) == "nccl":
            mine = mine.to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self.pycomm.all_gather(outs, mine)
        return max(1, sum(1 for o in outs if float(o.item()) == float(mine.item())))

    def _peer_selftest(self, lib, handle, cap):
        """A few exchanges of the sizes a step moves (and one longer than the mailbox: pieces), against torch.distributed's own
        all-reduce / all-gather on the same vectors; polls bounded to 3 s while testing."""
        dev = self.eng.device
        _capi.check(lib.tg_comm_peer_set_timeout_ms(handle, 3000.0))
        stream = self.eng._hip_stream
        good = True
        for i, n in enumerate((257, 2 * self.eng.K + 64, self.eng.C, 2 * self.eng.C + 64, cap + 4099)):
            g = torch.Generator(device="cpu").manual_seed(1000 * i + self.rank)
            x = torch.randn(n, generator=g, dtype=torch.float32).to(dev)
            ref = x.clone()
            self.pycomm.all_reduce(ref)
            got = x.clone()
            outs = torch.empty(self.world * n, dtype=torch.float32, device=dev)
            # a failure of THIS rank's library calls must not desynchronise the process group: every rank keeps issuing the same
            # sequence of group collectives and the verdicts are only compared at the end (round-5 advisor)
            if good:
                try:
                    _capi.check(lib.tg_comm_all_reduce_sum(handle, got.data_ptr(), n, stream))
                    _capi.check(lib.tg_comm_all_gather(handle, x.data_ptr(), outs.data_ptr(), n, stream))
                except Exception:        # noqa: BLE001
                    good = False
            ref_g = torch.empty(self.world * n, dtype=torch.float32, device=dev)
            self.pycomm.all_gather_into_tensor(ref_g, x)
            self.eng._sync()
            good = good and bool(torch.allclose(got, ref, rtol=1e-5, atol=1e-5)) and bool(torch.equal(outs, ref_g))
        flag = ct.c_int(0)
        _capi.check(lib.tg_comm_peer_status(handle, ct.byref(flag)))
        _capi.check(lib.tg_comm_peer_set_timeout_ms(handle, float(os.environ.get("TG_PEER_TIMEOUT_MS", "20000"))))
        return good and not flag.value

    def peer_check(self):
        """Peer transport: raise if an exchange ever gave up waiting for a peer (bounded polls, TG_PEER_TIMEOUT_MS); synchronises."""
        if self.transport != "peer" or not getattr(self, "_comm", None):
            return
        flag = ct.c_int(0)
        with (torch.cuda.device(self.eng.device) if self.eng.device.type == "cuda" else _Null()):
            _capi.check(self.eng._lib.tg_comm_peer_status(self._comm, ct.byref(flag)))
        if flag.value:
            raise RuntimeError("tangram_amd: a peer-memory exchange timed out waiting for another rank (results are invalid)")

    def _from_rank0(self, x, device):
        """`x` as held by rank 0, on every rank (float32 device tensor)."""
        t = x.detach().to(device=device, dtype=torch.float32).contiguous().clone() if isinstance(x, torch.Tensor) else \
            torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device=device)
        if hasattr(self.pycomm, "broadcast"):
            backend_cpu = isinstance(self.pycomm, DistComm) and dist.get_backend(self.group) != "nccl" and t.is_cuda
            buf = t.cpu() if backend_cpu else t
            self.pycomm.broadcast(buf, 0)
            return buf.to(device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.pycomm.all_gather(outs, t)
        return outs[0]

    # -- callback transport -----------------------------------------------------------------------------------------------
    def _cb_all_reduce(self, ctx, buf, n, stream):
        try:
            self.pycomm.all_reduce(self.eng.workspace_view(buf, n))
            return 0
        except BaseException as e:       # noqa: BLE001 -- reported by the caller of tg_mapper_step, never raised through C
            self._error = e
            return 1

    def _cb_all_gather(self, ctx, send, recv, n, stream):
        try:
            self.pycomm.all_gather_into_tensor(self.eng.workspace_view(recv, n * self.world), self.eng.workspace_view(send, n))
            return 0
        except BaseException as e:       # noqa: BLE001
            self._error = e
            return 1

    def _guard(self, fn, *args, **kw):
        try:
            return fn(*args, **kw)
        except Exception:
            if self._error is not None:
                err, self._error = self._error, None
                raise err
            raise

    def _attach(self):
        self._guard(self.eng.attach_comm, self._comm)

    # -- the hot path -----------------------------------------------------------------------------------------------------
    def run(self, n_steps, lr, history=None, first_row=0):
        """`n_steps` sharded iterations in ONE call of the C library; the history rows are global (identical on every rank)."""
        self._guard(self.eng.step, n_steps, lr, history, first_row)
        self._steps_since_check = getattr(self, "_steps_since_check", 0) + n_steps

    def checked(self):
        """Peer transport: synchronise and raise if any exchange since the last check gave up waiting (callers that consume history rows
        or validation numbers DURING a run -- print_each / val_each -- call this before they trust them)."""
        if getattr(self, "_steps_since_check", 0):
            self._steps_since_check = 0
            self.peer_check()

    def step(self, lr, history_row=None):
        hist = history_row.view(1, -1) if history_row is not None else None
        self.run(1, lr, hist, 0)

    def validate(self):
        """`_val_loss_fn` of the current mapping over ALL spots (collective: every rank calls it, every rank gets the same four
        numbers): per-gene sums, spot-cosine sum, entropy sum and non-zero fractions are all-reduced inside tg_mapper_validate."""
        out = self._guard(self.eng.validate)
        self.checked()
        return out

    def finalize_history(self, history):
        """Kept for callers of the earlier API: the rows written by `run` are already the global history."""
        return history

    def result_local(self, with_filter=False):
        """This rank's block of the mapping only: (softmax(M)[:, lo:hi] as a device tensor, (lo, hi)[, filter]) -- nothing is
        gathered (config 4: the full mapping is 40 GB; 8 ranks each returning it would be 320 GB of host memory)."""
        lo, hi = shard_bounds(self.n_spots_total, self.world, self.rank, self.spatial)
        self.peer_check()
        if with_filter:
            P_local, F = self.eng.result(with_filter=True)
            return P_local, (lo, hi), F
        return self.eng.result(), (lo, hi)

    def result_full(self, with_filter=False, host=False):
        """The column blocks of softmax(M) of every rank -> [C, V_total] on every rank (+ the replicated filter).
        `host=True`: a NumPy array assembled block by block (one rank's block is broadcast at a time), so that no GPU ever holds
        more than its own block plus one peer's -- for problems whose full mapping does not fit beside the training state."""
        self.peer_check()
        if with_filter:
            P_local, F = self.eng.result(with_filter=True)
        else:
            P_local, F = self.eng.result(), None
        P = self._gather_columns_host(P_local) if host else self._gather_columns(P_local)
        if host and F is not None:
            F = F.detach().cpu().numpy()
        return (P, F) if with_filter else P

    def _gather_columns_host(self, X_local):
        bounds = [shard_bounds(self.n_spots_total, self.world, r, self.spatial) for r in range(self.world)]
        if not hasattr(self.pycomm, "broadcast"):
            return self._gather_columns(X_local).detach().cpu().numpy()
        out = np.empty((X_local.shape[0], self.n_spots_total), dtype=np.float32)
        cpu_backend = isinstance(self.pycomm, DistComm) and dist.get_backend(self.group) != "nccl"
        for r, (lo, hi) in enumerate(bounds):
            if r == self.rank:
                blk = X_local.contiguous()
            else:
                blk = torch.empty((X_local.shape[0], hi - lo), dtype=torch.float32, device=X_local.device)
            if cpu_backend and blk.is_cuda:
                blk = blk.cpu()
            self.pycomm.broadcast(blk, r)
            out[:, lo:hi] = blk.detach().cpu().numpy()
            del blk
        return out

    def project_full(self, S_all=None, unfiltered=True):
        """softmax(M)^T S for every spot: each rank projects onto its own spots, the row blocks are gathered -> [V_total, K]."""
        self.checked()
        Gh = self.eng.project() if S_all is None else self.eng.project_genes(S_all, unfiltered=unfiltered)
        return self._gather_columns(Gh.t().contiguous()).t().contiguous()

    def _gather_columns(self, X_local):
        widths = [b - a for a, b in (shard_bounds(self.n_spots_total, self.world, r, self.spatial) for r in range(self.world))]
        wmax = max(widths)
        pad = torch.zeros((X_local.shape[0], wmax), dtype=torch.float32, device=X_local.device)
        pad[:, :X_local.shape[1]] = X_local
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.pycomm.all_gather(out, pad)
        return torch.cat([o[:, :w] for o, w in zip(out, widths)], dim=1)

    def release(self):
        if getattr(self, "eng", None) is not None:
            self.eng.release()
        if getattr(self, "_comm", None):
            self.eng._lib.tg_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def make_sharded(S, G, M0, d=None, d_source=None, F0=None, *, device, mode="mapper", precision="bf16x3", lambdas=None,
                 target_count=0.0, group=None, fwd_splits=0, tile_size=0, bwd_tile=0, comm=None, transport="auto",
                 voxel_weights=None, neighborhood_filter=None, ct_encode=None, spatial_weights=None, device_init_seed=None,
                 s_exact=False):
    """Slice full problem arrays (identical on every rank) into this rank's spot block.  The spot graphs of the spatial terms
    (`voxel_weights`, `neighborhood_filter`, `spatial_weights`: V x V over ALL spots) and `ct_encode` are passed whole.
    `device_init_seed` (with M0 = None): this rank's block of the initial logits is generated on ITS device
    (device_init.device_normal: the same logits whatever the number of ranks) -- the cells x spots plane never exists on a host."""
    pc = comm if comm is not None else DistComm(group)
    world, rank = pc.world, pc.rank
    V = G.shape[0]
    lam = lambdas or {}
    spatial = any(lam.get(k, 0) > 0 for k in ("lambda_neighborhood_g1", "lambda_ct_islands", "lambda_getis_ord", "lambda_moran", "lambda_geary"))
    # Emptiness is decided for EVERY rank's block, identically on every rank, before anything collective happens: with the
    # block partition of the spatial terms (ceil(V / world) spots per rank) trailing ranks can end up empty (V = 17 on 8 ranks
    # leaves ranks 6 and 7 without spots); raising only there would leave the others hanging in the next collective.
    empty = [r for r in range(world) if shard_bounds(V, world, r, spatial)[1] - shard_bounds(V, world, r, spatial)[0] < 1]
    if empty:
        raise ValueError(f"rank(s) {empty} would own no spots (V={V}, world={world}"
                         + (", blocks of ceil(V / world) spots for the spatial terms" if spatial else "") + ")")
    lo, hi = shard_bounds(V, world, rank, spatial)
    G_l = G[lo:hi]
    if M0 is None:
        if device_init_seed is None:
            raise ValueError("make_sharded needs M0 or device_init_seed")
        from .device_init import device_normal
        M_l = device_normal(S.shape[0], hi - lo, device, device_init_seed, col0=lo, n_cols_total=V)
    else:
        M_l = M0[:, lo:hi]
        M_l = np.ascontiguousarray(M_l) if isinstance(M_l, np.ndarray) else M_l.contiguous()
    d_l = None if d is None else d[lo:hi]
    return ShardedMapperEngine(S, G_l, M_l, d_l, d_source, F0, n_spots_total=V, device=device, mode=mode, precision=precision,
                               lambdas=lambdas, target_count=target_count, group=group, fwd_splits=fwd_splits, tile_size=tile_size,
                               bwd_tile=bwd_tile, comm=comm, transport=transport, spot_offset=lo, s_exact=s_exact, voxel_weights=voxel_weights,
                               neighborhood_filter=neighborhood_filter, ct_encode=ct_encode, spatial_weights=spatial_weights)
