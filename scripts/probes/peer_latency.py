"""Latency of ONE exchange of the peer transport on one rank (push into and poll from the rank's own mailbox: the kernel's fixed
cost without a link), as a DISTRIBUTION: `reps` batches of `batch` back-to-back exchanges between two HIP events each, for the three
vector lengths a step of cfg2 moves.  Round-5 records disagreed (8 - 9 us in run2/run3, 17.9 / 19.4 us in final/): this is the
measurement the review asked for.  Prints one JSON object."""
import ctypes as ct
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tangram_amd import _capi  # noqa: E402


def main():
    lib = _capi.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    C, Kp = 30000, 1024
    cap = 6 * C + 64
    h = ct.create_string_buffer(64)
    comm = ct.c_void_p()
    _capi.check(lib.tg_comm_peer_create_stepped(1, 0, cap, 0, 1, 1, h, ct.byref(comm)))
    _capi.check(lib.tg_comm_peer_connect(comm, h.raw))
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {"device": torch.cuda.get_device_name(0)}
    reps, batch = int(os.environ.get("REPS", 50)), int(os.environ.get("BATCH", 20))
    for name, n, gather in (("gene_stats_all_reduce_2Kp", 2 * Kp, 0), ("row_dots_all_reduce_C", C, 0), ("row_pairs_all_gather_2C", 2 * C + 64, 1)):
        x = torch.randn(n, device=dev)
        y = torch.empty(n, device=dev)
        def go():
            if gather:
                _capi.check(lib.tg_comm_all_gather(comm, x.data_ptr(), y.data_ptr(), n, stream))
            else:
                _capi.check(lib.tg_comm_all_reduce_sum(comm, x.data_ptr(), n, stream))
        for _ in range(50):
            go()
        torch.cuda.synchronize()
        us = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(batch):
                go()
            e1.record()
            torch.cuda.synchronize()
            us.append(1e3 * e0.elapsed_time(e1) / batch)
        us = np.array(us)
        out[name] = dict(n=n, batches=reps, per_batch=batch, us_min=float(us.min()), us_p10=float(np.percentile(us, 10)), us_median=float(np.median(us)),
                         us_p90=float(np.percentile(us, 90)), us_max=float(us.max()))
    flag = ct.c_int(0)
    _capi.check(lib.tg_comm_peer_status(comm, ct.byref(flag)))
    out["timed_out"] = flag.value
    # reference points on the same box: an empty kernel launch and a plain copy kernel of the same size, back to back
    a = torch.empty(2 * C + 64, device=dev)
    b = torch.empty_like(a)
    for nm, fn in (("torch_copy_2C_back_to_back", lambda: b.copy_(a)),):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1000):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[nm + "_us"] = 1e3 * e0.elapsed_time(e1) / 1000
    lib.tg_comm_destroy(comm)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
