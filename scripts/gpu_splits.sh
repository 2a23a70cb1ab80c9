#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/splits; rm -rf gpurun_out/*; mkdir -p $O
for P in bf16x3 bf16; do for S in 0 3 6 8 11 16; do
  timeout 200 python bench.py --splits $S --precision $P --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/${P}_$S.json 2> $O/${P}_$S.err || echo FAIL $P $S
  python - $O/${P}_$S.json $P $S <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k={x["name"]:round(x["avg_ms"],4) for x in d["kernels"]}
    print(sys.argv[2],"splits",sys.argv[3],"ms/step %.4f"%d["ms_per_step"],"fwd",k.get("tg_fwd_kernel"),"ghat_reduce",k.get("tg_ghat_reduce"),"sum",round(k.get("tg_fwd_kernel")+k.get("tg_ghat_reduce"),4))
except Exception as e: print("parse fail",sys.argv[1:],e)
PY
done; done
