import sys, os, time, ctypes as ct, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tangram_amd.engine import HipMapperEngine
from tangram_amd.synthetic import make_workload, init_logits
DEV = "cuda:0"
C, K, V = 30000, 1000, 10000
w = make_workload(C, K, V, DEV, seed=0)
M0 = init_logits(C, V, DEV, seed=42)
e = HipMapperEngine(w["S"], w["G"], M0, d=w["d"], device=DEV, precision="bf16x3", lambdas=dict(lambda_g1=1.0, lambda_d=1.0), schedule=2)
del M0
e._lib.tg_debug_fused_sync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int]
e.step(3, 0.1); torch.cuda.synchronize()
t0 = time.perf_counter(); e.step(1, 0.1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
buf = (ct.c_uint * 256)()
n = e._lib.tg_debug_fused_sync(e._h, buf, 256)
a = np.array(buf[:n]).astype(np.float64)
cyc = a[12:16] * 64 / 256          # average cycles per workgroup
print("step %.3f ms; per-WG average cycles: tile %.3e wait %.3e rows %.3e draw %.3e (2.2 GHz: tile %.2f wait %.2f rows %.2f draw %.2f ms)" %
      ((1e3 * dt,) + tuple(cyc) + tuple(cyc / 2.2e6)), "err", a[8], flush=True)
