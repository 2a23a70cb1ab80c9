#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -x -k "long_horizon or eight_processes or tuning_seeds or bench_prints" > $O/pytest_new.log 2>&1; echo "pytest rc=$? ${SECONDS}s"; tail -8 $O/pytest_new.log | cut -c1-400
cat gpurun_out/full_size_long_horizon_cfg2.json gpurun_out/cfg3_eight_processes_vs_reference.json 2>/dev/null | cut -c1-600
