#!/bin/bash
# round 6, call H: BASELINE config 4 as one of its eight ranks sees it (per-GPU proxy of the 8-GPU run that no box of this pool can hold)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06h
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
timeout 900 python scripts/bench_shard_proxy.py cfg4 > $O/shard_cfg4.json 2> $O/shard_cfg4.err; echo "rc=$?"; tail -3 $O/shard_cfg4.err
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/shard_cfg4.json").read().strip().splitlines()[-1])
for k,v in d.items(): print(k, "%.3f ms"%v["ms_per_step"], "loss %.6f"%v["main_loss"], v["kernels_us"])
PY
