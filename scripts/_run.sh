R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*; mkdir -p gpurun_out/r03d; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 > gpurun_out/r03d/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03d/pytest.log
bash scripts/gpu_ab.sh r03d_ab "bf16x3 bf16" 2 "" keep
