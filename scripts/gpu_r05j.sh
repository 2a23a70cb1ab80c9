#!/bin/bash
# round 5, call J: tg_spmm with XCD-banded rows -- cfg5b bench + rocprof stats, the spatial GPU tests, then the PMC passes on the final sources
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05j
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
timeout 600 python bench.py --workload cfg5b --steps 100 --warmup 10 --no-cpu-baseline --no-alt > $O/bench_cfg5b.json 2> $O/bench_cfg5b.err; echo "rc=$?"
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_cfg5b.json").read().strip().splitlines()[-1])
print("cfg5b %.1f it/s %.3f ms"%(d["value"],d["ms_per_step"]), {k["name"]:round(k["avg_ms"],4) for k in d["kernels"] if k["avg_ms"]>0.004})
PY
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "spatial or cfg5b or csr or autocorr or golden or random_configurations" > $O/pytest_spatial.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_spatial.log | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5b -o r -- python $R/bench.py --workload cfg5b --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/rocprof_cfg5b.log 2>&1; echo "rocprof rc=$?"
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
grep -E "tg_spmm|tg_colstats|tg_loss_finalize" $O/prof_cfg5b/r_kernel_stats.csv | cut -d, -f1-4
bash $R/scripts/gpu_pmc.sh pmc_final bf16x3 traffic keep > $O/pmc_bf16x3.out 2>&1; echo "pmc rc=$?"
bash $R/scripts/gpu_pmc.sh pmc_final_bf16 bf16 traffic keep > $O/pmc_bf16.out 2>&1; echo "pmc rc=$?"
