#!/bin/bash
# GPU tests, shard proxy (sharded driver on one GPU), optional extra bench arguments
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/*; mkdir -p gpurun_out/multi
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/multi/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/multi/pytest_gpu.log
timeout 600 python scripts/bench_shard_proxy.py > gpurun_out/multi/shard.json 2> gpurun_out/multi/shard.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/multi/shard.json") if l.startswith("{")][-1])
for k,v in d.items(): print(k, round(v["ms_per_step"],3))
PY
for A in "$@"; do timeout 600 python bench.py --no-cpu-baseline --no-alt $A > gpurun_out/multi/b.json 2> gpurun_out/multi/b.err; python - "$A" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/multi/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), {x["name"]:round(x["avg_ms"],3) for x in d["kernels"]})
PY
done
