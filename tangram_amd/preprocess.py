"""Host pre-processing on the device: the NumPy / pandas work `map_cells_to_space` and `pp_adatas` do on the
host before the first iteration, as kernels of libtangram_hip.so fed with the AnnData matrices AS THEY ARE STORED (scipy CSR):

  gather_training_genes   `adata[:, training_genes].X.toarray()` (mapping_utils.py:259-275) -> the dense [n_obs, K] float32 matrix
                          of the training genes, built on the device from the uploaded CSR arrays (bit-identical values);
                          no cells x genes dense matrix ever exists on the host
  rna_count_density       `X.sum(axis=1) / X.sum()` (pp_adatas, mapping_utils.py:88-89), sums in double, rounded once
  cluster_expression      the per-cluster sum / mean loop of adata_to_cluster_expression (mapping_utils.py:126-132), double accumulate

Everything goes through the C ABI (tg_csr_gather_columns, tg_row_sums, tg_cluster_aggregate); torch only owns the memory.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np
import torch

from . import _capi


def _stream(device):
    return ct.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else None


def _check_device(device):
    device = torch.device(device)
    if device.type != "cuda" and not _capi.is_emulated():
        raise RuntimeError(f"tangram_amd runs on a HIP device only (got device={device!r}); there is no CPU path")
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _call(device, fn, *args):
    if device.type == "cuda":
        with torch.cuda.device(device):
            return _capi.check(fn(*args))
    return _capi.check(fn(*args))


class DeviceCSR:
    """A scipy.sparse matrix uploaded once: int64 indptr, int32 indices, float32 data (canonical: sorted, no duplicates)."""

    def __init__(self, X, device):
        import scipy.sparse as sp
        self.device = _check_device(device)
        csr = X.tocsr() if sp.issparse(X) else sp.csr_matrix(np.asarray(X))
        if not csr.has_canonical_format:
            csr = csr.copy()                       # never edit the caller's matrix
            csr.sum_duplicates()
        self.shape = csr.shape
        self.indptr = torch.as_tensor(np.asarray(csr.indptr, dtype=np.int64), device=self.device)
        self.indices = torch.as_tensor(np.asarray(csr.indices, dtype=np.int32), device=self.device)
        self.data = torch.as_tensor(np.asarray(csr.data, dtype=np.float32), device=self.device)


def gather_training_genes(X, col_index, device):
    """Dense float32 device tensor [n_obs, len(col_index)] = X[:, col_index] for a scipy.sparse (or DeviceCSR) X.
    Reference: `np.array(adata[:, training_genes].X.toarray(), dtype="float32")` (mapping_utils.py:259-275)."""
    csr = X if isinstance(X, DeviceCSR) else DeviceCSR(X, device)
    col_index = np.asarray(col_index, dtype=np.int64)
    if len(col_index) < 1 or col_index.min() < 0 or col_index.max() >= csr.shape[1]:
        raise ValueError("gene column index out of range")
    if len(np.unique(col_index)) != len(col_index):
        raise ValueError("duplicate gene columns")
    colmap = np.full(csr.shape[1], -1, dtype=np.int32)
    colmap[col_index] = np.arange(len(col_index), dtype=np.int32)
    colmap_d = torch.as_tensor(colmap, device=csr.device)
    out = torch.empty((csr.shape[0], len(col_index)), dtype=torch.float32, device=csr.device)
    lib = _capi.lib()
    _call(csr.device, lib.tg_csr_gather_columns, csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.data.data_ptr(), int(csr.shape[0]),
          colmap_d.data_ptr(), int(len(col_index)), out.data_ptr(), int(out.stride(0)), _stream(csr.device))
    return out


def row_sums(X, device, normalize=False):
    """float32 device vector of the row sums of X (dense array / tensor, scipy.sparse or DeviceCSR), accumulated in double;
    `normalize`: divided by their total -> `rna_count_per_spot / np.sum(rna_count_per_spot)` (mapping_utils.py:88-89)."""
    lib = _capi.lib()
    if isinstance(X, DeviceCSR) or hasattr(X, "tocsr"):
        csr = X if isinstance(X, DeviceCSR) else DeviceCSR(X, device)
        out = torch.empty(csr.shape[0], dtype=torch.float32, device=csr.device)
        _call(csr.device, lib.tg_row_sums, None, 0, 0, csr.indptr.data_ptr(), csr.data.data_ptr(), int(csr.shape[0]), out.data_ptr(),
              int(bool(normalize)), _stream(csr.device))
        return out
    device = _check_device(device)
    Xd = torch.as_tensor(np.asarray(X) if not isinstance(X, torch.Tensor) else X).to(device=device, dtype=torch.float32)
    if Xd.dim() != 2:
        raise ValueError("X must be a matrix")
    if Xd.stride(1) != 1:
        Xd = Xd.contiguous()
    out = torch.empty(Xd.shape[0], dtype=torch.float32, device=device)
    _call(device, lib.tg_row_sums, Xd.data_ptr(), int(Xd.stride(0)), int(Xd.shape[1]), None, None, int(Xd.shape[0]), out.data_ptr(),
          int(bool(normalize)), _stream(device))
    return out


def rna_count_density(X, device):
    """`rna_count_based_density` of pp_adatas (mapping_utils.py:88-89) as a float32 device vector."""
    return row_sums(X, device, normalize=True)


def cluster_expression(X_dev, labels, unique_labels, scale=True):
    """[n_clusters, K] float32 device tensor: for every label in `unique_labels` (in that order) the sum (scale=True) or mean of the
    rows of the dense device matrix `X_dev` carrying it (mapping_utils.py:126-132)."""
    if not isinstance(X_dev, torch.Tensor) or X_dev.dtype != torch.float32 or X_dev.dim() != 2:
        raise ValueError("X_dev must be a float32 device matrix")
    if X_dev.stride(1) != 1:
        X_dev = X_dev.contiguous()
    labels = np.asarray(labels)
    if len(labels) != X_dev.shape[0]:
        raise ValueError("one label per row expected")
    order, indptr = [], [0]
    for l in unique_labels:
        rows = np.nonzero(labels == l)[0]
        order.append(rows)
        indptr.append(indptr[-1] + len(rows))
    dev = X_dev.device
    rows_d = torch.as_tensor(np.concatenate(order).astype(np.int32) if indptr[-1] else np.zeros(1, np.int32), device=dev)
    indptr_d = torch.as_tensor(np.asarray(indptr, dtype=np.int32), device=dev)
    out = torch.empty((len(unique_labels), X_dev.shape[1]), dtype=torch.float32, device=dev)
    _call(dev, _capi.lib().tg_cluster_aggregate, X_dev.data_ptr(), int(X_dev.stride(0)), int(X_dev.shape[1]), indptr_d.data_ptr(),
          rows_d.data_ptr(), int(len(unique_labels)), 0 if scale else 1, out.data_ptr(), int(out.stride(0)), _stream(dev))
    return out
