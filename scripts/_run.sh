R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*; mkdir -p gpurun_out/r03c; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 > gpurun_out/r03c/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03c/pytest.log
export TG_BENCH_TIMING_ABLATION=1
bash scripts/gpu_ab.sh r03c_abl "bf16x3" 2 "" keep
