#!/bin/bash
# multi-rank plumbing smoke (gloo, 2 ranks sharing the one GPU), cfg4 per-GPU shard proxy, cfg5a/5b runs
TAG=${1:-misc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
echo "== 2-rank gloo smoke"; TG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --shape 6000,500,4000 --no-cpu-baseline --no-alt > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "rc=$?"; tail -c 600 $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
echo "== same problem 1 rank"; timeout 600 python bench.py --steps 5 --warmup 2 --shape 6000,500,4000 --no-cpu-baseline --no-alt > $O/bench_1rank_small.json 2> $O/bench_1rank_small.err; echo "rc=$?"; tail -c 300 $O/bench_1rank_small.json
echo "== cfg4 shard proxy (200k x 2k x 6250 = 1/8 of 50k spots), bf16"; timeout 900 python bench.py --steps 6 --warmup 2 --shape 200000,2000,6250 --precision bf16 --no-cpu-baseline --no-alt > $O/bench_cfg4_shard_bf16.json 2> $O/bench_cfg4_shard_bf16.err; echo "rc=$?"; tail -3 $O/bench_cfg4_shard_bf16.err
echo "== cfg3 shard proxy (30k x 1k x 1250), bf16x3"; timeout 900 python bench.py --steps 20 --warmup 4 --shape 30000,1000,1250 --no-cpu-baseline --no-alt > $O/bench_cfg3_shard.json 2> $O/bench_cfg3_shard.err; echo "rc=$?"
echo "== cfg5 constrained / spatial"; timeout 900 python scripts/bench_cfg5.py > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "rc=$?"; tail -3 $O/bench_cfg5.err; cat $O/bench_cfg5.json
for f in $O/bench_cfg4_shard_bf16.json $O/bench_cfg3_shard.json $O/bench_1rank_small.json $O/bench_2rank_gloo.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","n_gpus","last_main_loss")})
    for k in d["kernels"]: print("    %-26s %8.4f ms x%d" % (k["name"],k["avg_ms"],k["launches"]))
except Exception as e: print("parse fail",e)
PY
done
du -sh $R/gpurun_out
