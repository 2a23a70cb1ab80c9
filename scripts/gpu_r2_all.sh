#!/bin/bash
# round 2: smoke, pytest -m gpu, every BASELINE workload through bench.py, the --gpus 2 refusal on a 1-GPU box, rocprofv3 stats
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name"; free -g | head -2) > $O/env.log 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest gpu"; S=$SECONDS; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in $((SECONDS-S)) s"; tail -12 $O/pytest_gpu.log
echo "== bench default"; S=$SECONDS; timeout 1200 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "rc=$? wall=$((SECONDS-S))s"; tail -3 $O/bench_cfg2.err
for W in cfg5a cfg5b; do
  echo "== bench $W"; S=$SECONDS; timeout 1200 python bench.py --workload $W --no-alt > $O/bench_$W.json 2> $O/bench_$W.err; echo "rc=$? wall=$((SECONDS-S))s"; tail -3 $O/bench_$W.err
done
echo "== bench cfg4"; S=$SECONDS; timeout 1200 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "rc=$? wall=$((SECONDS-S))s"; tail -3 $O/bench_cfg4.err
echo "== bench --gpus 2 on this box"; python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "rc=$?"; cat $O/bench_gpus2.err | tail -3
echo "== 2-rank gloo smoke (self-launched)"; TG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-alt > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "rc=$?"; tail -c 300 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank_gloo.err
echo "== rocprof of the default command (shorter)"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_cfg*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "PARSE FAIL", e); continue
    print(os.path.basename(f), "value %.2f it/s  %.3f ms/step  loss %.5f" % (d["value"], d["ms_per_step"], d["last_main_loss"]))
    r=d["roofline"]; print("   roofline:", {k:r[k] for k in ("bound","achieved","peak","unit","frac","traffic","hbm_frac","mfma_frac")})
    for k in r["kernels"]: print("      %-22s %8.4f ms  %s %.3f" % (k["name"],k["avg_ms"],k["bound"],k["frac"]))
    print("   cpu_baseline:", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","cores","kind","cell_spot_gene_per_s","wall_s","error","thread_sweep_s_per_iter")})
    print("   alt:", {k:round(v["value"],1) for k,v in d.get("alt_precisions",{}).items()})
PY
head -12 $O/prof/*kernel_stats.csv 2>/dev/null | cut -c1-150
du -sh $R/gpurun_out
