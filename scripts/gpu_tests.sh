#!/bin/bash
# pytest -m gpu (optionally -k EXPR), log under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/*; mkdir -p gpurun_out
S=$SECONDS
timeout 1200 python -m pytest tests -m gpu -q ${1:+-k "$1"} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? in $((SECONDS-S)) s"; tail -15 gpurun_out/pytest_gpu.log
