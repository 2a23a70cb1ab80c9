#!/bin/bash
# round 6 final evidence: what the driver runs (smoke, pytest -m gpu, default bench) + every workload + rocprofv3 stats + the PMC traffic passes
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/scripts/gpu_final.sh r06final
echo "== peer latency"; cd $R; timeout 300 python scripts/probes/peer_latency.py > $R/gpurun_out/r06final/peer_latency.json 2>/dev/null; echo "rc=$?"
echo "== batched (with the tuning caller)"; timeout 900 python scripts/bench_batched.py --batches 16 --epochs 500 > $R/gpurun_out/r06final/batched_e2e.json 2> $R/gpurun_out/r06final/batched_e2e.err; echo "rc=$?"; tail -c 600 $R/gpurun_out/r06final/batched_e2e.json
bash $R/scripts/gpu_pmc.sh pmc_final bf16x3 traffic keep > $R/gpurun_out/r06final/pmc_bf16x3.out 2>&1; echo "pmc rc=$?"
bash $R/scripts/gpu_pmc.sh pmc_final_bf16 bf16 traffic keep > $R/gpurun_out/r06final/pmc_bf16.out 2>&1; echo "pmc rc=$?"
du -sh $R/gpurun_out
