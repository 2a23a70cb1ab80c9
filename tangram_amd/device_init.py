"""Initial logits generated on the device (opt-in: `Mapper(..., init="device")`).

The reference draws `np.random.normal(0, 1, (n_cells, n_spots))` on the host in float64 (mapping_optimizer.py:147-157); the default
path here reproduces that stream bit for bit (host_rng.py).  At BASELINE config 4 that plane is 200 000 x 50 000 = 40 GB of fp32 (80 GB of
float64 draws) PER RANK of a spot-sharded run -- it cannot exist on the host.  `tg_init_logits_normal` (include/tangram_hip.h) fills a
block of the plane on the device from a counter-based generator: element (cell, global spot) depends on (seed, cell * n_spots_total +
spot) only, so a rank generates exactly the columns it owns and every partition of the spots starts from the same logits."""
from __future__ import annotations

import torch

from . import _capi


def device_normal(n_rows, n_cols, device, seed, col0=0, n_cols_total=None, stream_id=0):
    """[n_rows, n_cols] float32 standard-normal block on `device`: columns col0 .. col0 + n_cols of an [n_rows, n_cols_total] plane.
    `stream_id` separates independent draws of one seed (0: the logits M, 1: the filter logits F of MapperConstrained)."""
    device = torch.device(device)
    n_cols_total = int(n_cols if n_cols_total is None else n_cols_total)
    if device.type != "cuda" and not _capi.is_emulated():
        raise RuntimeError(f"tangram_amd runs on a HIP device only (got device={device!r}); there is no CPU path")
    out = torch.empty((int(n_rows), int(n_cols)), dtype=torch.float32, device=device)
    seed = (int(seed) * 0x9E3779B1 + int(stream_id) * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
    lib = _capi.lib()
    if device.type == "cuda":
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _capi.check(lib.tg_init_logits_normal(out.data_ptr(), int(n_rows), int(n_cols), int(n_cols), seed, int(col0), n_cols_total, stream))
    else:
        _capi.check(lib.tg_init_logits_normal(out.data_ptr(), int(n_rows), int(n_cols), int(n_cols), seed, int(col0), n_cols_total, None))
    return out
