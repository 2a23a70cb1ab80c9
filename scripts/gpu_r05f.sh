#!/bin/bash
# round 5, call F: peer_checked (set-up self-test) across processes on one GPU; 2-rank bench smoke over gloo with both transports
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== peer"; timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q --timeout 600 -k "peer_transport" > $O/pytest_peer.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_peer.log
for T in callbacks peer_checked; do
  echo "== 2-rank gloo smoke, transport $T"
  TG_BENCH_BACKEND=gloo TG_SHARD_TRANSPORT=$T timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2rank_$T.json 2> $O/bench_2rank_$T.err; echo "rc=$?"
  python - $O/bench_2rank_$T.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%.1f it/s %.3f ms"%(d["value"],d["ms_per_step"]), d["config"]["parallelism"], {k["name"]:round(k["avg_ms"],4) for k in d["kernels"] if "exchange" in k["name"] or k["avg_ms"]>0.1}, "loss %.6f"%d["last_main_loss"])
except Exception as e: print("unparsable", e)
PY
done
du -sh $R/gpurun_out
