#!/bin/bash
# round 5, call I: after the filter-kernel restructure -- cfg5a bench + profile, the GPU suite, and the PMC passes on the final sources
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== bench cfg5a"; timeout 600 python bench.py --workload cfg5a --steps 100 --warmup 10 --no-cpu-baseline --no-alt > $O/bench_cfg5a.json 2> $O/bench_cfg5a.err; echo "rc=$?"
echo "== bench cfg2"; timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$?"
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_cfg*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "%.1f it/s  %.3f ms"%(d["value"],d["ms_per_step"]), {k["name"]:round(k["avg_ms"],4) for k in d["kernels"] if k["avg_ms"]>0.005}, "traffic", d["roofline"].get("traffic"))
PY
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5a -o r -- python $R/bench.py --workload cfg5a --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $O/rocprof_cfg5a.log 2>&1; echo "rocprof rc=$?"
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
bash $R/scripts/gpu_pmc.sh pmc_final bf16x3 traffic keep > $O/pmc_bf16x3.out 2>&1; echo "pmc rc=$?"
bash $R/scripts/gpu_pmc.sh pmc_final_bf16 bf16 traffic keep > $O/pmc_bf16.out 2>&1; echo "pmc rc=$?"
du -sh $R/gpurun_out
