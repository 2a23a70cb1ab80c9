#!/bin/bash
# What the driver runs at round end -- smoke, pytest -m gpu, the default bench -- plus the other workloads, a self-launched 2-rank
# gloo smoke and rocprofv3 kernel stats of the default command.   usage: gpu_final.sh tag [keep]
TAG=${1:-final}; KEEP=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
[ -z "$KEEP" ] && rm -rf $R/gpurun_out/*
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name"; echo "cgroup memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"; free -g | head -2) > $O/env.log 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
echo "== bench default"; SECONDS=0; timeout 1200 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$? wall=${SECONDS}s"; tail -3 $O/bench_cfg2.err
for W in cfg5a cfg5b cfg4; do
  echo "== bench $W"; timeout 900 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_$W.json 2> $O/bench_$W.err; echo "rc=$?"
done
for T in callbacks peer_checked; do   # two ranks sharing the one GPU (gloo bootstraps): the callback transport, and the peer transport behind its self-test
  echo "== 2-rank gloo smoke, transport $T"; TG_BENCH_BACKEND=gloo TG_SHARD_TRANSPORT=$T timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_2rank_$T.json 2> $O/bench_2rank_$T.err; echo "rc=$?"; tail -c 200 $O/bench_2rank_$T.json
done
echo "== shard proxy"; timeout 600 python scripts/bench_shard_proxy.py > $O/shard.json 2> $O/shard.err; echo "rc=$?"
echo "== batched folds"; timeout 600 python scripts/bench_batched.py --no-e2e --batches 8,16 > $O/batched.json 2> $O/batched.err; echo "rc=$?"; tail -c 400 $O/batched.json
echo "== rocprof of the default command (shortened)"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
for W in cfg5a cfg5b cfg4; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o r -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/rocprof_$W.log 2>&1; echo "rocprof $W rc=$?"
done
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_cfg*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"unparsable"); continue
    print(os.path.basename(f), "%.1f it/s  %.3f ms"%(d["value"],d["ms_per_step"]), {k["name"]:round(k["avg_ms"],3) for k in d["kernels"] if k["avg_ms"]>0.05})
PY
head -12 $O/prof/*kernel_stats.csv | cut -c1-150
du -sh $R/gpurun_out
