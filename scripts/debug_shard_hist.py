import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tangram_oracle as orc
from tangram_amd.sharded import make_sharded
from tests.local_comm import run_ranks
C, K, V = 350, 40, 777
data = orc.make_synthetic(C, K, V, seed=17)
M0, F0 = orc.reference_init_MF_constrained(C, V, 23)
lam = dict(lambda_g1=1.0, lambda_d=1.0, lambda_g2=0.6, lambda_r=1e-3, lambda_count=0.7, lambda_f_reg=1.5)
n = 6
def f(comm):
    sh = make_sharded(data["S"], data["G"], M0, d=data["d"], F0=F0, mode="constrained", device="cuda:0", precision="bf16x3", lambdas=lam, target_count=120.0, comm=comm)
    hs = sh.eng.new_history(n); sh.run(n, 0.1, hs)
    P, F = sh.result_full(with_filter=True)
    return hs.cpu().numpy(), F.cpu().numpy()
for trial in range(2):
    res = run_ranks(2, f)
    d = res[0][0].astype(np.float64) - res[1][0].astype(np.float64)
    print("trial", trial, "hist diff at", np.argwhere(np.nan_to_num(d) != 0).tolist(), "F diff", np.abs(res[0][1] - res[1][1]).max())
    print(res[0][0][:, [0, 1, 2, 3, 4, 9, 10]])
    print(res[1][0][:, [0, 1, 2, 3, 4, 9, 10]])
