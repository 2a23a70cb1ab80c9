#!/bin/bash
# round 5, call B: the peer-memory transport on the GPU (in-process shards on streams of their own; two processes over hipIpc) and the
# shard proxy with both transports.   usage: gpu_r05b.sh tag
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== peer transport tests"
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q --timeout 600 -k "peer_transport" > $O/pytest_peer.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_peer.log
echo "== shard proxy"
timeout 600 python scripts/bench_shard_proxy.py > $O/shard.json 2> $O/shard.err; echo "rc=$?"; tail -3 $O/shard.err
python - $O <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]+"/shard.json") if l.startswith("{")][-1])
    for k,v in d.items(): print(k, "%.4f ms/step"%v["ms_per_step"], "enqueue %.3f"%v["host_enqueue_ms_per_step"], v["kernels_us"])
except Exception as e: print("no shard json", e)
PY
du -sh $R/gpurun_out
