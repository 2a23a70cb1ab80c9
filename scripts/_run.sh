R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*; export PYTHONUNBUFFERED=1
bash scripts/gpu_ab.sh r03i_rot "bf16x3" 3 "" keep
