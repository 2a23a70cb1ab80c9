#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/*; mkdir -p gpurun_out/batch
S=$SECONDS; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/batch/pytest_gpu.log 2>&1; echo "pytest rc=$? in $((SECONDS-S)) s"; tail -5 gpurun_out/batch/pytest_gpu.log
timeout 900 python scripts/bench_batched.py > gpurun_out/batch/bench_batched.json 2> gpurun_out/batch/bench_batched.err; echo "rc=$?"; tail -2 gpurun_out/batch/bench_batched.err; cat gpurun_out/batch/bench_batched.json
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/batch/bench_cfg2.json 2> gpurun_out/batch/bench_cfg2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/batch/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], {k["name"]:round(k["avg_ms"],4) for k in d["kernels"]}, {k:round(v["value"],1) for k,v in d["alt_precisions"].items()})
PY
