#!/bin/bash
# round 6, call A: baseline of the round on today's box -- shard proxy, default bench (no CPU leg), cfg4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06a
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== shard proxy"; timeout 600 python scripts/bench_shard_proxy.py > $O/shard.json 2> $O/shard.err; echo "rc=$?"; tail -c 3000 $O/shard.json
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?"; tail -c 1500 $O/bench_cfg2.json
echo "== bench cfg4"; timeout 900 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "rc=$?"; tail -c 1500 $O/bench_cfg4.json
