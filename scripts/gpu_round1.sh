#!/bin/bash
# One gpurun call: tests -> smoke -> bench (3 precisions) -> rocprofv3 kernel trace.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
echo "== rocminfo" > $O/env.log; (rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; nproc; free -g | head -2; lscpu | grep "Model name") >> $O/env.log 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
echo "== bench bf16x3"; timeout 900 python bench.py --steps 30 --warmup 5 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; echo "rc=$?"; tail -c 1500 $O/bench_bf16x3.json
echo "== bench bf16"; timeout 600 python bench.py --steps 30 --warmup 5 --precision bf16 --no-cpu-baseline --no-alt > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "rc=$?"
echo "== bench fp32"; timeout 600 python bench.py --steps 10 --warmup 2 --precision fp32 --no-cpu-baseline --no-alt > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "rc=$?"
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16x3 -o r1 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $O/rocprof_bf16x3.log 2>&1; echo "rocprof rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o r1 -- python $R/bench.py --steps 10 --warmup 2 --precision bf16 --no-cpu-baseline --no-alt > $O/rocprof_bf16.log 2>&1; echo "rocprof rc=$?"
cd $R
# keep only the small summaries
find $O -name "*.csv" -size +2M -delete
ls -la $O $O/prof_bf16x3 2>/dev/null | head -40
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","roofline","last_main_loss")})
    for k in d["kernels"]: print("   ",k)
    print(d.get("cpu_baseline"))
except Exception as e: print("parse fail",e)
PY
done
