#!/bin/bash
# bench.py with several argument sets in one call: gpu_args.sh tag "args1" "args2" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
i=0
for A in "$@"; do
  i=$((i+1))
  timeout 600 python bench.py --no-cpu-baseline --no-alt $A > $O/b$i.json 2> $O/b$i.err || echo "FAIL: $A"
  python - $O/b$i.json "$A" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k={x["name"]:x["avg_ms"] for x in d["kernels"]}
    print("%-60s step %.3f | fwd %.3f bwd %.3f adam %.3f small %.3f" % (sys.argv[2], d["ms_per_step"], k.get("tg_fwd_kernel",0), k.get("tg_bwd_kernel",0), (k.get("tg_adam_update",0)+k.get("tg_adam_rowpass",0)), sum(v for n,v in k.items() if n not in ("tg_fwd_kernel","tg_bwd_kernel","tg_adam_update","tg_adam_rowpass"))))
except Exception as e: print("parse fail", sys.argv[2], e)
PY
done
