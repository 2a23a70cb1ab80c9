R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*; export PYTHONUNBUFFERED=1
bash scripts/gpu_ab.sh r03e_pitch "bf16x3 bf16" 3 "" keep
