#!/bin/bash
# round 5, call H: memory-pattern variants of tg_adam_rowpass (out-of-tree builds under build/: ab_late0 = second moment requested with
# the other arrays, not behind pass 1; ab_ntst0 = ordinary instead of non-temporal stores of M, m, v), same-box A/B at the default step count
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for r in 1 2 3; do
 for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  n=$(basename $lib .so)
  timeout 300 python scripts/with_lib.py $lib bench.py --steps 200 --warmup 20 --precision bf16x3 --no-cpu-baseline --no-alt > $O/${n}_r$r.json 2> $O/${n}_r$r.err || echo "FAIL $n"
 done
done
python - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*_r?.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print("parse fail",f); continue
    print(os.path.basename(f), "%.1f it/s %.3f ms"%(d["value"],d["ms_per_step"]), {x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if x["avg_ms"]>0.1})
PY
