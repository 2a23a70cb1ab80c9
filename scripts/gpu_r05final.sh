#!/bin/bash
# round 5: the final validation (scripts/gpu_final.sh) followed by the PMC passes the traffic table is built from (same box, same sources)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/scripts/gpu_final.sh final
bash $R/scripts/gpu_pmc.sh pmc_final bf16x3 traffic keep
bash $R/scripts/gpu_pmc.sh pmc_final_bf16 bf16 traffic keep
du -sh $R/gpurun_out
