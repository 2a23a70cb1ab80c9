#!/bin/bash
# One gpurun call: tests -> bench -> rocprofv3 kernel trace (CSV stats only).  Small outputs land in gpurun_out/.
# usage: gpu_round.sh [tag] [skip_tests]
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > $O/env.log 2>&1
if [ -z "$2" ]; then
  echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
fi
for P in bf16x3 bf16 fp32; do
  echo "== bench $P"; timeout 900 python bench.py --steps 20 --warmup 4 --precision $P --no-cpu-baseline --no-alt > $O/bench_$P.json 2> $O/bench_$P.err; echo "rc=$?"
done
if [ -f $R/build/libtangram_hip_alt.so ]; then
for P in bf16x3 bf16; do
  echo "== bench $P ALT lib"; TANGRAM_AMD_LIB=$R/build/libtangram_hip_alt.so timeout 900 python bench.py --steps 20 --warmup 4 --precision $P --no-cpu-baseline --no-alt > $O/bench_${P}_alt.json 2> $O/bench_${P}_alt.err; echo "rc=$?"
done
fi
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
for P in bf16x3 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$P -o r -- python $R/bench.py --steps 10 --warmup 2 --precision $P --no-cpu-baseline --no-alt > $O/rocprof_$P.log 2>&1; echo "rocprof $P rc=$?"
done
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","roofline","last_main_loss")}, d.get("kernels_pass"))
    for k in d["kernels"]: print("    %-26s %8.4f ms x%d" % (k["name"],k["avg_ms"],k["launches"]))
except Exception as e: print("parse fail",e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
done
find $O -name "*stats*.csv" | head; for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -12 $f; done
du -sh $R/gpurun_out
