"""ctypes binding of libtangram_hip.so (C ABI declared in include/tangram_hip.h).

The product path loads the in-tree HIP library built by `__graft_entry__.build()` /
`tangram_amd._build.build()` and fails loudly when it is missing: there is no CPU fallback.
(The CPU test-suite installs an *emulated* build of the same sources through
`_install_library_for_tests`; nothing else calls that hook.)
"""
from __future__ import annotations

import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtangram_hip.so")      # the in-tree build, nothing else (experiments: scripts/with_lib.py)

TG_ABI_VERSION = 6
TG_MODE_MAPPER, TG_MODE_CONSTRAINED = 0, 1
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2}
H_NTERMS = 16
H_TOTAL, H_MAIN, H_VG, H_KL, H_ENTROPY, H_L1, H_L2, H_NB, H_CT, H_COUNT, H_FREG, H_GETIS, H_MORAN, H_GEARY = range(14)


class TgConfig(ct.Structure):
    _fields_ = [(n, ct.c_int32) for n in
                ("abi_version", "mode", "precision", "n_cells", "n_genes", "n_spots", "n_spots_total",
                 "has_density", "has_d_source", "fwd_splits", "tile_size", "pipeline_bands")] + \
               [(n, ct.c_float) for n in
                ("lambda_g1", "lambda_d", "lambda_g2", "lambda_r", "lambda_l1", "lambda_l2",
                 "lambda_count", "lambda_f_reg", "target_count", "lambda_neighborhood_g1", "lambda_ct_islands")] + \
               [(n, ct.c_int32) for n in ("n_cell_types", "nnz_w", "nnz_n")] + \
               [(n, ct.c_float) for n in ("lambda_getis_ord", "lambda_moran", "lambda_geary")] + [("nnz_s", ct.c_int32)] + \
               [(n, ct.c_float) for n in ("beta1", "beta2", "eps")] + \
               [(n, ct.c_int32) for n in ("n_ranks", "bwd_tile", "spot_offset", "s_exact_mode")]


ALL_REDUCE_FN = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_void_p)                 # tg_all_reduce_sum_fn
ALL_GATHER_FN = ct.CFUNCTYPE(ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_void_p)    # tg_all_gather_fn


class TgSizes(ct.Structure):
    _fields_ = [("state_bytes", ct.c_size_t), ("workspace_bytes", ct.c_size_t),
                ("m_pitch", ct.c_int32), ("history_terms", ct.c_int32), ("peer_step_floats", ct.c_size_t)]


class TgInputs(ct.Structure):
    _fields_ = [(n, ct.c_void_p) for n in ("S_dev", "G_dev", "d_dev", "d_source_dev", "M0_dev", "F0_dev", "ct_encode_dev",
                                           "w_indptr", "w_indices", "w_data", "wt_indptr", "wt_indices", "wt_data",
                                           "n_indptr", "n_indices", "n_data", "nt_indptr", "nt_indices", "nt_data",
                                           "s_indptr", "s_indices", "s_data", "st_indptr", "st_indices", "st_data")]


_lib = None
_is_sim = False


def _declare(lib):
    vp, i32, f32 = ct.c_void_p, ct.c_int, ct.c_float
    lib.tg_abi_version.restype = i32
    lib.tg_last_error.restype = ct.c_char_p
    lib.tg_query_sizes.argtypes = [ct.POINTER(TgConfig), ct.POINTER(TgSizes)]
    lib.tg_mapper_create.argtypes = [ct.POINTER(TgConfig), ct.POINTER(TgInputs), vp, vp, vp, ct.POINTER(vp)]
    lib.tg_mapper_destroy.argtypes = [vp]
    lib.tg_mapper_destroy.restype = None
    lib.tg_mapper_step.argtypes = [vp, i32, f32, vp, i32]
    lib.tg_comm_create_callbacks.argtypes = [i32, i32, ALL_REDUCE_FN, ALL_GATHER_FN, vp, ct.POINTER(vp)]
    lib.tg_comm_rccl_unique_id.argtypes = [ct.c_char_p, vp]
    lib.tg_comm_create_rccl.argtypes = [ct.c_char_p, vp, i32, i32, ct.POINTER(vp)]
    lib.tg_comm_peer_create.argtypes = [i32, i32, ct.c_size_t, i32, vp, ct.POINTER(vp)]
    lib.tg_comm_peer_create_stepped.argtypes = [i32, i32, ct.c_size_t, ct.c_size_t, i32, i32, vp, ct.POINTER(vp)]
    lib.tg_comm_peer_connect.argtypes = [vp, vp]
    lib.tg_comm_peer_status.argtypes = [vp, ct.POINTER(i32)]
    lib.tg_comm_peer_set_timeout_ms.argtypes = [vp, ct.c_double]
    lib.tg_comm_all_reduce_sum.argtypes = [vp, vp, ct.c_size_t, vp]
    lib.tg_comm_all_gather.argtypes = [vp, vp, vp, ct.c_size_t, vp]
    for name in ("tg_comm_peer_create", "tg_comm_peer_create_stepped", "tg_comm_peer_connect", "tg_comm_peer_status", "tg_comm_peer_set_timeout_ms", "tg_comm_all_reduce_sum",
                 "tg_comm_all_gather"):
        getattr(lib, name).restype = i32
    lib.tg_comm_destroy.argtypes = [vp]
    lib.tg_comm_destroy.restype = None
    lib.tg_mapper_attach_comm.argtypes = [vp, vp]
    lib.tg_mapper_result.argtypes = [vp, vp, vp]
    lib.tg_mapper_project.argtypes = [vp, vp]
    lib.tg_mapper_project_genes.argtypes = [vp, vp, ct.c_int64, i32, vp, ct.c_int64, i32]
    lib.tg_csr_columns_to_dense.argtypes = [vp, vp, vp, ct.c_int64, i32, i32, vp, ct.c_int64, vp]
    lib.tg_csr_columns_to_dense.restype = i32
    lib.tg_batch_query_bytes.argtypes = [i32]
    lib.tg_batch_query_bytes.restype = ct.c_size_t
    lib.tg_batch_create.argtypes = [ct.POINTER(vp), i32, vp, ct.POINTER(vp)]
    lib.tg_batch_create.restype = i32
    lib.tg_batch_step.argtypes = [vp, i32, f32, ct.POINTER(vp), i32]
    lib.tg_batch_step.restype = i32
    lib.tg_batch_destroy.argtypes = [vp]
    lib.tg_batch_destroy.restype = None
    lib.tg_csr_gather_columns.argtypes = [vp, vp, vp, ct.c_int64, vp, i32, vp, ct.c_int64, vp]
    lib.tg_row_sums.argtypes = [vp, ct.c_int64, i32, vp, vp, ct.c_int64, vp, i32, vp]
    lib.tg_cluster_aggregate.argtypes = [vp, ct.c_int64, i32, vp, vp, i32, i32, vp, ct.c_int64, vp]
    lib.tg_init_logits_normal.argtypes = [vp, ct.c_int64, ct.c_int64, ct.c_int64, ct.c_uint64, ct.c_int64, ct.c_int64, vp]
    for name in ("tg_csr_gather_columns", "tg_row_sums", "tg_cluster_aggregate", "tg_init_logits_normal"):
        getattr(lib, name).restype = i32
    lib.tg_mapper_state.argtypes = [vp, ct.POINTER(vp), ct.POINTER(vp), ct.POINTER(vp), ct.POINTER(ct.c_int32),
                                    ct.POINTER(ct.c_int64)]
    lib.tg_mapper_set_step.argtypes = [vp, ct.c_int64]
    lib.tg_mapper_effective_precision.argtypes = [vp]
    lib.tg_mapper_effective_precision.restype = i32
    lib.tg_mapper_filter_state.argtypes = [vp, ct.POINTER(vp), ct.POINTER(ct.c_int32)]
    lib.tg_mapper_filter_state.restype = i32
    lib.tg_mapper_validate.argtypes = [vp, vp]
    lib.tg_mapper_profile.argtypes = [vp, i32]
    lib.tg_mapper_profile_read.argtypes = [vp, ct.c_char_p, ct.c_size_t, ct.POINTER(ct.c_float), ct.POINTER(i32), i32,
                                           ct.POINTER(i32)]
    for name in ("tg_query_sizes", "tg_mapper_create", "tg_mapper_step", "tg_comm_create_callbacks", "tg_comm_rccl_unique_id",
                 "tg_comm_create_rccl", "tg_mapper_attach_comm", "tg_mapper_result", "tg_mapper_project", "tg_mapper_project_genes",
                 "tg_mapper_state", "tg_mapper_set_step", "tg_mapper_profile", "tg_mapper_profile_read", "tg_mapper_validate"):
        getattr(lib, name).restype = i32
    return lib


EXPORTS = ["tg_abi_version", "tg_last_error", "tg_query_sizes", "tg_mapper_create", "tg_mapper_destroy",
           "tg_mapper_step", "tg_comm_create_callbacks", "tg_comm_rccl_unique_id", "tg_comm_create_rccl", "tg_comm_peer_create", "tg_comm_peer_create_stepped", "tg_comm_peer_connect", "tg_comm_peer_status", "tg_comm_peer_set_timeout_ms",
           "tg_comm_all_reduce_sum", "tg_comm_all_gather", "tg_comm_destroy",
           "tg_mapper_attach_comm", "tg_mapper_result",
           "tg_mapper_project", "tg_mapper_project_genes", "tg_csr_columns_to_dense", "tg_csr_gather_columns", "tg_row_sums",
           "tg_cluster_aggregate", "tg_batch_query_bytes", "tg_batch_create", "tg_batch_step", "tg_batch_destroy", "tg_mapper_state", "tg_mapper_set_step",
           "tg_mapper_filter_state", "tg_mapper_profile",
           "tg_mapper_profile_read", "tg_mapper_validate", "tg_init_logits_normal", "tg_mapper_effective_precision"]


def lib():
    """The loaded HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). tangram_amd has no CPU fallback.")
        _lib = _declare(ct.CDLL(LIB_PATH))
        if _lib.tg_abi_version() != TG_ABI_VERSION:
            raise RuntimeError("libtangram_hip.so ABI version mismatch (rebuild: python -c 'import __graft_entry__ as g; g.build()')")
    return _lib


def is_emulated():
    return _is_sim


def _install_library_for_tests(path):
    """TEST HOOK: use an emulated (TG_SIM) build of the same C ABI; `None` restores the HIP library."""
    global _lib, _is_sim
    if path is None:
        _lib, _is_sim = None, False
    else:
        _lib, _is_sim = _declare(ct.CDLL(path)), True


class TangramHipError(RuntimeError):
    pass


def check(rc):
    if rc == 0:
        return
    msg = lib().tg_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise TangramHipError(f"libtangram_hip error {rc}: {msg}")
