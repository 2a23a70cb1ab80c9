#!/usr/bin/env python3
"""PROBE (not part of the product): the kernels of one cfg2 iteration on a CU-MASKED stream.

The round-4 review gates the backward || update overlap on this number: "the update alone on an ~80-CU mask must reach >= 5.5 TB/s".
A `tg_mapper` binds to the stream it is created on; this script creates one handle per mask on a stream from
hipExtStreamCreateWithCUMask (bit i = CU i/8 of XCD i%8, scripts/probes/cumask_probe.hip: the first n bits spread n/8 CUs over each
XCD), steps it with the library's per-kernel HIP-event profile and prints, per mask: ms of tg_fwd_kernel / tg_bwd_kernel /
tg_adam_rowpass and the update's traffic rate (28 B per element actually moved: reads X, M, m, v, writes M, m, v).

    python scripts/probes/cumask_step.py [--shape 30000,1000,10000] [--cus 256,192,160,128,96,80,64,48,32] [--steps 6]
"""
import argparse
import ctypes as ct
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tangram_amd.engine import HipMapperEngine  # noqa: E402
from tangram_amd.synthetic import init_logits, make_workload  # noqa: E402


def masked_stream(hip, n_cus):
    mask = (ct.c_uint32 * 8)()
    for i in range(n_cus):
        mask[i // 32] |= 1 << (i % 32)
    s = ct.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ct.byref(s), 8, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="30000,1000,10000")
    ap.add_argument("--cus", default="256,192,160,128,96,80,64,48,32")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    C, K, V = (int(x) for x in args.shape.split(","))
    dev = torch.device("cuda:0")
    hip = ct.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ct.POINTER(ct.c_void_p), ct.c_uint32, ct.POINTER(ct.c_uint32)]
    w = make_workload(C, K, V, dev, seed=0)
    M0 = init_logits(C, V, dev, seed=42)
    torch.cuda.synchronize()
    rows = []
    for n in [int(x) for x in args.cus.split(",")]:
        if n >= 256:
            stream = torch.cuda.Stream(device=dev)
        else:
            stream = torch.cuda.ExternalStream(masked_stream(hip, n).value, device=dev)
        with torch.cuda.stream(stream):
            e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=dev, precision=args.precision,
                                lambdas=dict(lambda_g1=1.0, lambda_d=1.0))
            e.step(2, 0.1)
            e.profile(True)
            e.step(args.steps, 0.1)
            prof = {k: ms / max(cnt, 1) for k, ms, cnt in e.profile_read()}
            e.release()
        stream.synchronize()
        upd = prof.get("tg_adam_rowpass", float("nan"))
        row = dict(cus=n, fwd_ms=round(prof.get("tg_fwd_kernel", float("nan")), 4), bwd_ms=round(prof.get("tg_bwd_kernel", float("nan")), 4),
                   update_ms=round(upd, 4), update_TBps=round(28.0 * C * V / (upd * 1e-3) / 1e12, 3),
                   step_ms=round(sum(prof.values()), 4))
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(dict(probe="cumask_step", shape=[C, K, V], precision=args.precision, rows=rows)))


if __name__ == "__main__":
    main()
