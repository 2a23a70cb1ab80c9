R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*; export PYTHONUNBUFFERED=1
bash scripts/gpu_ab.sh r03g_w4 "bf16x3" 3 "" keep
python scripts/with_lib.py build/ab_w4.so -m pytest 2>/dev/null | tail -1
