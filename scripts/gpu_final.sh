#!/bin/bash
# what the driver runs at round end: smoke, pytest -m gpu, default bench; plus rocprofv3 kernel stats of the same command
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $O/env.log 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
echo "== bench default"; SECONDS=0; timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? wall=${SECONDS}s"; tail -3 $O/bench_default.err
echo "== 2-rank gloo smoke"; TG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "rc=$?"; tail -c 400 $O/bench_2rank_gloo.json
echo "== rocprof of the same command"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","scaling","vs_baseline","dtype","data","config","cell_spot_gene_per_s","roofline","iteration_roofline","cpu_baseline","alt_precisions","kernels_pass"):
    print(k,":",d.get(k))
for k in d["kernels"]: print("    %-26s %8.4f ms x%d" % (k["name"],k["avg_ms"],k["launches"]))
PY
head -12 $O/prof/*kernel_stats.csv | cut -c1-150
du -sh $R/gpurun_out
