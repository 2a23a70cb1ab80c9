#!/bin/bash
# GPU parity tests, then the same-box A/B of scripts/gpu_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/gpu_ab.sh "$@"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
