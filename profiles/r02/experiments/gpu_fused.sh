#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/fused
rm -rf gpurun_out/*; mkdir -p $O; export PYTHONUNBUFFERED=1
S=$SECONDS; timeout 600 python -m pytest tests/test_gpu_production_tiles.py -q -x -k fused > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$? in $((SECONDS-S)) s"; tail -6 $O/pytest_fused.log
for r in 1 2; do for sch in 1 2; do for P in bf16x3 bf16; do
  timeout 300 python bench.py --steps 40 --warmup 5 --precision $P --schedule $sch --no-cpu-baseline --no-alt > $O/s${sch}_${P}_r$r.json 2> $O/s${sch}_${P}_r$r.err || echo FAIL $sch $P
done; done; done
python - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/s*_r?.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print("parse fail",f); continue
    print(os.path.basename(f), "step %.3f ms  %.1f it/s" % (d["ms_per_step"], d["value"]), {k["name"]:round(k["avg_ms"],3) for k in d["kernels"] if k["avg_ms"]>0.05}, "loss %.6f" % d["last_main_loss"])
PY
