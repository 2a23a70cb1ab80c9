#!/bin/bash
# round 6, call D: the whole GPU suite + the shard proxy at the settled design
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06d
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
echo "== shard proxy"; timeout 900 python scripts/bench_shard_proxy.py 8_bf16x3 > $O/shard.json 2> $O/shard.err; echo "rc=$?"
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/shard.json").read().strip().splitlines()[-1])
for k,v in d.items(): print(k, "%.4f ms"%v["ms_per_step"], "loss %.9f"%v["main_loss"], v["kernels_us"])
PY
