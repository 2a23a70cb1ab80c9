R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/*
bash scripts/gpu_final.sh final keep
bash scripts/gpu_pmc.sh pmc_final bf16x3 traffic keep 2>&1 | tail -3
bash scripts/gpu_pmc.sh pmc_final_bf16 bf16 traffic keep 2>&1 | tail -3
