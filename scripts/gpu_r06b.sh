#!/bin/bash
# round 6, call B: the fused exchanges -- shard proxy (all transports), the exchange latency distribution, the sharded / peer GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06b
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== shard proxy"; timeout 900 python scripts/bench_shard_proxy.py > $O/shard.json 2> $O/shard.err; echo "rc=$?"; tail -3 $O/shard.err
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/shard.json").read().strip().splitlines()[-1])
for k,v in d.items(): print(k, "%.4f ms"%v["ms_per_step"], "enq %.3f"%v["host_enqueue_ms_per_step"], "loss %.9f"%v["main_loss"], v["kernels_us"])
PY
echo "== peer latency"; timeout 300 python scripts/probes/peer_latency.py > $O/peer_latency.json 2> $O/peer_latency.err; echo "rc=$?"; cat $O/peer_latency.json
echo "== pytest sharded / peer"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x -k "shard or peer or rccl" > $O/pytest_shard.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_shard.log
