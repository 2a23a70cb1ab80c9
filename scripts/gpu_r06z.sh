#!/bin/bash
# round 6, last call: the driver's three commands at HEAD -- smoke, pytest -m gpu, the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06z
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_at_head.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu_at_head.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_driver_args.json').read().strip().splitlines()[-1]); print(d['value'], d['value_long']['value'], d['roofline']['frac'], d['roofline']['update_TBps_actual'], d['roofline']['traffic'])"
